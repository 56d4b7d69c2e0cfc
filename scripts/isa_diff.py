#!/usr/bin/env python3
"""Which gfx950 kernels differ, instruction for instruction, between two commits (default: the last commit that ran on hardware,
cfef442 = round 4's call "r04a", and the work tree)?  Both trees' .hip files are compiled to device assembly (hipcc -S
--offload-device-only), every kernel's body is cut out and compared after dropping labels' numbering and comments.  A kernel
whose ISA is identical carries its measurements over; one that differs is new code as far as the GPU is concerned.
usage: isa_diff.py [old-commit] [--write profiles/r06_isa_vs_last_hardware_run.txt]"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["sketch.hip", "pairs.hip", "pairs_probe.hip", "pairs_join.hip", "ani.hip", "gz_inflate.hip"]


def kernels_of(tree, f):
    src = os.path.join(tree, "galah_amd", "csrc", f)
    if not os.path.exists(src):
        return {}
    with tempfile.NamedTemporaryFile(suffix=".s") as out:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--offload-device-only", src, "-o", out.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        text = open(out.name).read()
    res = {}
    # a kernel: from "name:" after a .type name,@function to its .Lfunc_end
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.groups()
        if ".amdhsa_kernel " + name not in text:
            continue   # a device function, not a kernel
        lines = []
        for ln in body.splitlines():
            ln = ln.split(";")[0].rstrip()
            if not ln.strip() or ln.lstrip().startswith("."):
                continue   # directives and block labels
            lines.append(re.sub(r"\.LBB\d+_\d+", ".LBB", ln))   # branch targets by name only (their order is kept by the instruction stream)
        res[name] = (hashlib.sha256("\n".join(lines).encode()).hexdigest()[:12], len(lines))
    return res


def pretty(d):
    d = d.replace("(anonymous namespace)::", "")
    d = re.sub(r"^void ", "", d)
    depth = 0
    for i, ch in enumerate(d):   # cut the argument list: the first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return d[:i]
    return d


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
            if len(out) == len(names):
                return [pretty(o) for o in out]
        except OSError:
            pass
    return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    write = sys.argv[sys.argv.index("--write") + 1] if "--write" in sys.argv else None
    old = args[0] if args and args[0] != write else "cfef442"
    with tempfile.TemporaryDirectory() as d:
        tar = subprocess.run(["git", "-C", ROOT, "archive", old, "galah_amd/csrc", "include"], capture_output=True, check=True).stdout
        subprocess.run(["tar", "-x", "-C", d], input=tar, check=True)
        with ThreadPoolExecutor(5) as ex:
            olds = list(ex.map(lambda f: kernels_of(d, f), FILES))
            news = list(ex.map(lambda f: kernels_of(ROOT, f), FILES))
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "galah_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
    out = [f"# device ISA of every kernel: commit {old} (old) against {head}{' + uncommitted changes' if dirty else ''} (new); hipcc -O3 --offload-arch=gfx950 -S",
           f"{'file':16} {'kernel':64} {'old instr':>9} {'new instr':>9}  verdict"]
    same = differ = 0
    for f, o, n in zip(FILES, olds, news):
        names = sorted(set(o) | set(n))
        for nm, pretty in zip(names, demangle(names)):
            a, b = o.get(nm), n.get(nm)
            if a and b and a[0] == b[0]:
                v = "identical"
                same += 1
            else:
                v = "NEW" if not a else "REMOVED" if not b else "DIFFERS"
                differ += 1
            out.append(f"{f:16} {pretty[:64]:64} {a[1] if a else '-':>9} {b[1] if b else '-':>9}  {v}")
    out.append(f"# {same} kernels identical, {differ} new / removed / different")
    text = "\n".join(out) + "\n"
    if write:
        with open(os.path.join(ROOT, write), "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
