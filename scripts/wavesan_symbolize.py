#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Turns the `library+0xoffset` code addresses of the wave race detector's reports (tests/emu/wavesan.cpp,
WAVESAN_LOG files or stderr) into function + file:line of the PRODUCT's source, via llvm-symbolizer.  The emulated build compiles
tests/emu/build/src/X.hip.cpp, a mechanical rewrite of galah_amd/csrc/X.hip: line numbers are the product's own (the rewrites keep
line counts) unless a rewrite changed them, so the text of the line is printed next to it.
usage: wavesan_symbolize.py report-file... > readable"""
import os
import re
import subprocess
import sys

SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def symbolize(lib, off):
    key = (lib, off)
    if key not in _cache:
        # the return address points behind the call: the instrumented access is the instruction before it
        out = subprocess.run([SYMBOLIZER, "--obj=" + lib, "--inlines", "--functions=short", hex(int(off, 16) - 1)], capture_output=True, text=True).stdout
        lines = [l for l in out.strip().splitlines() if l]
        frames = []
        for fn, loc in zip(lines[0::2], lines[1::2]):
            m = re.match(r"(.*):(\d+):(\d+)$", loc)
            text = ""
            if m and os.path.exists(m.group(1)):
                try:
                    text = open(m.group(1)).read().splitlines()[int(m.group(2)) - 1].strip()
                except Exception:
                    pass
            short = loc.replace(ROOT + "/", "").replace("tests/emu_tmp/", "tests/emu/")
            frames.append((fn, short, text))
        _cache[key] = frames
    return _cache[key]


def main():
    seen = set()
    for path in sys.argv[1:]:
        for line in open(path, errors="replace"):
            if not line.startswith("WAVESAN"):
                continue
            m = re.match(r"WAVESAN (.*?) in (.*?): (.*?) by wave (\d+) lane (\d+) at (\S+)\+(0x[0-9a-f]+)\s+vs\s+(.*?) by wave (\d+) at (\S+)\+(0x[0-9a-f]+)\s+\((.*)\)", line)
            if not m:
                print(line.rstrip())
                continue
            what, kernel, k1, w1, l1, lib1, off1, k2, w2, lib2, off2, rest = m.groups()
            a, b = symbolize(lib1, off1), symbolize(lib2, off2)
            key = (what, kernel, tuple(f[1] for f in a), tuple(f[1] for f in b))
            if key in seen:
                continue
            seen.add(key)
            print(f"{what} in {kernel}")
            for tag, kind, wave, frames in (("  this ", k1, w1, a), ("  other", k2, w2, b)):
                print(f"{tag}: {kind} by wave {wave}")
                for fn, loc, text in frames:
                    print(f"          {loc}  {fn}    | {text[:150]}")
            print(f"         ({rest})")
    print(f"# {len(seen)} distinct reports by source location")


if __name__ == "__main__":
    main()
