#!/usr/bin/env python3
"""Static resource usage of every gfx950 kernel of the library, as hipcc reports it (-Rpass-analysis=kernel-resource-usage):
VGPRs, AGPRs, SGPRs, scratch bytes per lane (spills), LDS bytes, occupancy in waves per SIMD.  No GPU needed.
usage: kernel_resources.py [--write profiles/r06_kernel_resources.txt]"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "galah_amd", "csrc")
FILES = ["sketch.hip", "pairs.hip", "pairs_probe.hip", "pairs_join.hip", "ani.hip", "gz_inflate.hip"]
FIELDS = [("Function Name", "name"), ("VGPRs", "vgprs"), ("AGPRs", "agprs"), ("SGPRs", "sgprs"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occupancy"), ("LDS Size [bytes/block]", "lds")]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        if len(out) != len(names):
            return names
        res = []
        for d in out:
            d = re.sub(r"^void ", "", d.replace("(anonymous namespace)::", ""))
            depth, cut = 0, len(d)
            for i, ch in enumerate(d):
                if ch == "<":
                    depth += 1
                elif ch == ">":
                    depth -= 1
                elif ch == "(" and depth == 0:
                    cut = i
                    break
            res.append(d[:cut])
        return res
    except OSError:
        return names


def analyse(f):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(CSRC, f), "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows, cur = [], {}
    for line in r.stderr.splitlines():
        for label, key in FIELDS:
            m = re.search(re.escape(label) + r": (\S+)", line)
            if m:
                if key == "name" and cur:
                    rows.append(cur)
                    cur = {}
                cur[key] = m.group(1)
    if cur:
        rows.append(cur)
    for row, nm in zip(rows, demangle([r_["name"] for r_ in rows])):
        row["file"], row["name"] = f, nm
    return rows


def table():
    with ThreadPoolExecutor(len(FILES)) as ex:
        rows = [r for rs in ex.map(analyse, FILES) for r in rs]
    return sorted(rows, key=lambda r: (r["file"], r["name"]))


def render(rows):
    out = [f"{'file':16} {'kernel':58} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'waves/SIMD':>10}"]
    for r in rows:
        out.append(f"{r['file']:16} {r['name'][:58]:58} {r['vgprs']:>5} {r.get('agprs', '0'):>5} {r['sgprs']:>5} {r['scratch']:>8} {r['lds']:>7} {r['occupancy']:>10}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = render(table())
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, sys.argv[sys.argv.index("--write") + 1]), "w") as f:
            f.write("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (static; scripts/kernel_resources.py)\n" + text)
    sys.stdout.write(text)
