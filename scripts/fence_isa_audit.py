#!/usr/bin/env python3
"""Static view of what orders memory BETWEEN workgroups in the gfx950 code of every kernel (no GPU needed): the library's .hip
files are compiled to device assembly (hipcc -S --offload-device-only) and, kernel by kernel, the instructions that matter
for an agent-scope hand-over are listed in program order --

    buffer_wbl2 sc1         write back this XCD's L2 (what an agent-scope RELEASE needs before the flag is raised)
    buffer_inv sc1          invalidate L1 / L2 lines (what an agent-scope ACQUIRE needs after the flag is seen)
    global_atomic_*         the flag / counter / reservation itself
    global_load/store ... sc1   agent-scope atomic loads and stores (bypass the non-coherent levels)
    s_sleep                 a spin loop

The emulator's race detector (tests/emu/wavesan.cpp) checks the SOURCE's release / acquire structure; this shows that the
compiler turned it into the cache maintenance the hardware needs across XCDs (each XCD has its own L2: a flag raised without
buffer_wbl2, or read without buffer_inv, works on one XCD and fails across two).  A kernel with atomics but neither wbl2 nor
inv only reserves / counts (no data is handed over through memory inside the launch) -- listed with that verdict.
usage: fence_isa_audit.py [--write profiles/r06_fence_isa.txt]"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "galah_amd", "csrc")
FILES = ["sketch.hip", "pairs.hip", "pairs_probe.hip", "pairs_join.hip", "ani.hip", "gz_inflate.hip"]
WANT = re.compile(r"\b(buffer_wbl2|buffer_inv|global_atomic_\w+|flat_atomic_\w+|s_sleep)\b|\bglobal_(load|store)_\w+ .*\bsc1\b")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"^void ", "", d.replace("(anonymous namespace)::", "")).split("(")[0] for d in out] if len(out) == len(names) else names


def audit(f):
    with tempfile.NamedTemporaryFile(suffix=".s") as out:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--offload-device-only", os.path.join(CSRC, f), "-o", out.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        text = open(out.name).read()
    rows = []
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        seq = []
        for line in m.group(2).splitlines():
            line = line.split(";")[0].strip()
            if WANT.search(line):
                op = line.split()[0]
                tail = " ".join(t for t in line.split() if t in ("sc0", "sc1", "nt"))
                seq.append((op + " " + tail).strip())
        rows.append((m.group(1), seq))
    names = demangle([r[0] for r in rows])
    return [(f, n, s) for n, (_, s) in zip(names, rows)]


def verdict(seq):
    ops = " ".join(seq)
    atomics = "atomic" in ops
    if not atomics and "buffer_" not in ops and "sc1" not in ops:
        return None
    rel, acq = "buffer_wbl2 sc1" in ops, "buffer_inv sc1" in ops
    if rel and acq:
        return "hand-over between workgroups: release (wbl2) and acquire (inv) both present"
    if rel or acq:
        return "one-sided: " + ("release only (publishes; consumed by a LATER launch or by the host)" if rel else "acquire only")
    return "atomics only: reserves / counts, no data handed over inside the launch"


def compress(seq):
    out, i = [], 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i]:
            j += 1
        out.append(seq[i] + (f" x{j - i}" if j - i > 1 else ""))
        i = j
    return out


def main():
    with ThreadPoolExecutor(6) as ex:
        per_file = list(ex.map(audit, FILES))
    lines = ["# scripts/fence_isa_audit.py: ordering-relevant instructions of every gfx950 kernel that has any, in program order (hipcc -O3, static)"]
    n = 0
    for rows in per_file:
        for f, name, seq in rows:
            v = verdict(seq)
            if v is None:
                continue
            n += 1
            lines.append(f"{f}  {name}")
            lines.append(f"    {v}")
            c = compress(seq)
            for k in range(0, len(c), 6):
                lines.append("      " + " ; ".join(c[k:k + 6]))
    lines.append(f"# {n} kernels with atomics / fences of {sum(len(r) for r in per_file)}")
    text = "\n".join(lines) + "\n"
    if "--write" in sys.argv:
        open(os.path.join(ROOT, sys.argv[sys.argv.index("--write") + 1]), "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
