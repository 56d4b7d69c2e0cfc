#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Which lines of the KERNEL files does the emulated suite execute?  Input: the HIPEMU_COV_OUT.<pid> files a
run of the coverage build leaves (make -C tests/emu SAN=cov; tests/emu/covrt.cpp): one line per control-flow edge of the kernel
files, `library+offset hit`.  The files of a run's processes are merged (an edge is hit if any process hit it), the addresses
symbolized (llvm-symbolizer, innermost inlined frame), and per source file the report gives edges and lines reached and the
line ranges NO test reached -- code whose parity with the oracle nothing has ever checked, on the emulator or on hardware.
Lines are the product's own (the emulated build compiles mechanical rewrites of galah_amd/csrc/*.hip that keep line numbers).
usage: emu_coverage.py cov-file... [--write profiles/r06_emu_kernel_coverage.txt]"""
import collections
import os
import re
import subprocess
import sys

SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_path = sys.argv[sys.argv.index("--write") + 1] if "--write" in sys.argv else None
    if out_path:
        args = [a for a in args if a != out_path]
    hit = {}
    for path in args:
        for line in open(path):
            loc, h = line.split()
            lib, off = loc.rsplit("+", 1)
            key = (os.path.basename(lib), off)
            hit[key] = hit.get(key, 0) | int(h)
            hit.setdefault(("__lib__", os.path.basename(lib)), lib)
    libs = {k[1]: v for k, v in hit.items() if k[0] == "__lib__"}
    edges = [(k, v) for k, v in hit.items() if k[0] != "__lib__"]
    by_lib = collections.defaultdict(list)
    for (lib, off), h in edges:
        by_lib[lib].append((off, h))
    lines = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))   # file -> line -> [edges, hit edges]
    funcs = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))   # file -> function -> [edges, hit]
    for lib, offs in by_lib.items():
        r = subprocess.run([SYMBOLIZER, "--obj=" + libs[lib], "--functions=short", "--no-inlines"], input="\n".join(o for o, _ in offs) + "\n",
                           capture_output=True, text=True)
        blocks = [b for b in r.stdout.strip().split("\n\n")]
        assert len(blocks) == len(offs), (len(blocks), len(offs))
        for (off, h), b in zip(offs, blocks):
            parts = b.strip().splitlines()
            fn, loc = parts[0], parts[1]
            m = re.match(r"(.*):(\d+):(\d+)$", loc)
            if not m or m.group(2) == "0":
                continue
            f = os.path.basename(m.group(1)).replace(".hip.cpp", ".hip")
            e = lines[f][int(m.group(2))]
            e[0] += 1
            e[1] += 1 if h else 0
            g = funcs[f][fn]
            g[0] += 1
            g[1] += 1 if h else 0
    src_dir = os.path.join(ROOT, "galah_amd", "csrc")
    out = ["# scripts/emu_coverage.py: control-flow edges of the kernel files reached by the emulated suite (tests/emu, SAN=cov)",
           f"# {len(args)} process files merged; an edge counts as reached if any process reached it", ""]
    tot_e = tot_h = 0
    for f in sorted(lines):
        if not os.path.exists(os.path.join(src_dir, f)):
            continue   # the emulator's own headers
        le = lines[f]
        e = sum(v[0] for v in le.values())
        h = sum(v[1] for v in le.values())
        ln = len(le)
        lh = sum(1 for v in le.values() if v[1])
        tot_e += e
        tot_h += h
        out.append(f"{f}: edges {h}/{e} ({100.0 * h / max(e, 1):.1f} %), lines with code {lh}/{ln} ({100.0 * lh / max(ln, 1):.1f} %)")
        dead = sorted(fn for fn, v in funcs[f].items() if v[1] == 0)
        if dead:
            out.append("    functions never entered: " + ", ".join(dead))
        text = open(os.path.join(src_dir, f)).read().splitlines()
        miss = sorted(l for l, v in le.items() if v[1] == 0)
        runs, i = [], 0
        while i < len(miss):
            j = i
            while j + 1 < len(miss) and miss[j + 1] - miss[j] <= 2:
                j += 1
            runs.append((miss[i], miss[j]))
            i = j + 1
        for a, b in runs:
            first = text[a - 1].strip()[:110] if a - 1 < len(text) else ""
            out.append(f"    not reached: {f}:{a}" + (f"-{b}" if b != a else "") + f"    | {first}")
        out.append("")
    out.append(f"# all kernel files: edges {tot_h}/{tot_e} ({100.0 * tot_h / max(tot_e, 1):.1f} %)")
    res = "\n".join(out) + "\n"
    if out_path:
        open(os.path.join(ROOT, out_path) if not os.path.isabs(out_path) else out_path, "w").write(res)
    sys.stdout.write(res)


if __name__ == "__main__":
    main()
