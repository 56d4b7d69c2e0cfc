#!/usr/bin/env python3
"""Every file the design notes cite must exist: scans DESIGN.md, docs/design/*.md, INTEGRATION.md, README.md and profiles/README*.md
for back-ticked repository paths (profiles/..., scripts/..., tests/..., galah_amd/..., oracle/..., include/..., docs/..., bench.py)
-- with an optional `:line`, `:from-to` or `::name` behind them -- and fails on a path that does not exist, a line number past
the end of the file, or a `::name` the file does not contain.  Reference citations (`src/...rs:N`, `tests/test_cmdline.rs:N`)
are checked the same way against /root/reference where that exists (the build container; not the GPU box).  A double-quoted
string that directly follows a cited profiles/ file ("profiles/x.json: \"...\"") must occur in that file.
Run by tests/test_docs.py in the CPU suite.  usage: check_citations.py [-v]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md"] + sorted(glob.glob(os.path.join(ROOT, "docs", "design", "*.md"))) + sorted(glob.glob(os.path.join(ROOT, "profiles", "README*.md")))
REPO_PREFIX = ("profiles/", "scripts/", "tests/", "galah_amd/", "oracle/", "include/", "docs/", "bench.py", "__graft_entry__.py")
REF_PREFIX = ("src/", "tests/test_cmdline.rs", "tests/data/", "Cargo.toml", "pixi.", "docs/preludes/")
TOKEN = re.compile(r"`([^`\s]+)`")
# files that exist only after a run / a build, or are named as patterns of such
GENERATED = ("gpurun_out/", "oracle/_ref", "galah_amd/libgalah_hip.so", "galah_amd/csrc/build/", "tests/emu/build/", "tests/emu/libgalah_hip_emu.so",
             "tests/emu/fake_rccl/librccl.so.1", "oracle/libgalah_oracle.so", "scripts/ubench/hash_variants", "scripts/ubench/int_ops", "scripts/ubench/host_costs")


def line_count(path, cache={}):
    if path not in cache:
        with open(path, errors="replace") as f:
            cache[path] = sum(1 for _ in f)
    return cache[path]


def check_token(tok, where, problems, checked):
    tok = tok.rstrip(".,;)")
    name, line, member = tok, None, None
    m = re.match(r"^(.*?)::([\w\[\]-]+)$", tok)
    if m:
        name, member = m.groups()
    m = re.match(r"^(.*?):(\d+)(?:[-–](\d+))?(?:,\d+(?:[-–]\d+)?)*$", name)
    if m:
        name, line = m.group(1), int(m.group(3) or m.group(2))
    is_repo = name.startswith(REPO_PREFIX)
    is_ref = name.startswith(REF_PREFIX) and not (is_repo and os.path.exists(os.path.join(ROOT, name)))
    if is_ref and not (name.startswith("tests/") and os.path.exists(os.path.join(ROOT, name))):
        if not os.path.isdir(REF):
            return
        base = REF
    elif is_repo:
        base = ROOT
    else:
        return
    if any(name.startswith(g) for g in GENERATED) or "<" in name or "{" in name or "…" in name or "..." in name:
        return
    path = os.path.join(base, name)
    checked.append(tok)
    if "*" in name or "?" in name:
        if not glob.glob(path):
            problems.append(f"{where}: `{tok}` matches nothing")
        return
    if not os.path.exists(path):
        problems.append(f"{where}: `{tok}` does not exist" + (" in the reference" if base == REF else ""))
        return
    if line is not None and os.path.isfile(path) and line > line_count(path):
        problems.append(f"{where}: `{tok}` -- the file has {line_count(path)} lines")
    if member and os.path.isfile(path):
        with open(path, errors="replace") as f:
            if member.split("[")[0] not in f.read():
                problems.append(f"{where}: `{tok}` -- no `{member}` in the file")


def main():
    problems, checked = [], []
    for doc in DOCS:
        path = doc if os.path.isabs(doc) else os.path.join(ROOT, doc)
        if not os.path.exists(path):
            continue
        rel = os.path.relpath(path, ROOT)
        with open(path) as f:
            text = f.read()
        fence = False
        for no, ln in enumerate(text.splitlines(), 1):
            if ln.strip().startswith("```"):
                fence = not fence
            if fence:
                continue
            for m in TOKEN.finditer(ln):
                check_token(m.group(1), f"{rel}:{no}", problems, checked)
            for m in re.finditer(r"`(profiles/[^`\s:]+)`:?\s+\"([^\"]{3,80})\"", ln):
                p = os.path.join(ROOT, m.group(1))
                if os.path.isfile(p):
                    with open(p, errors="replace") as f:
                        if m.group(2) not in f.read():
                            problems.append(f"{rel}:{no}: \"{m.group(2)}\" is not in `{m.group(1)}`")
    if "-v" in sys.argv:
        print(f"{len(checked)} citations checked in {len(DOCS)} documents")
    for p in problems:
        print(p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
