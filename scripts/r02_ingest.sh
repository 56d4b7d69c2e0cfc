export TMPDIR=/tmp
python scripts/ingest_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "^\[ingest\]" | tail -6
for PL in 1 0; do for TP in 8 12 16 24; do
echo "== GHIP_PIPELINE=$PL plain threads $TP"
GHIP_PIPELINE=$PL GHIP_INGEST_THREADS_PLAIN=$TP python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.getcwd())
import numpy as np
import galah_amd
from concurrent.futures import ThreadPoolExecutor
from bench import _fasta_bytes
n, L = 1000, 5_000_000
ctx = galah_amd.Context(0)
d = "/dev/shm/ghip_probe2"
if not os.path.isdir(d):
    os.makedirs(d)
    g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)
    seqs = [g.to_host(i) for i in range(n)]
    del g
    def write(i):
        with open(os.path.join(d, f"g{i:05d}.fna"), "wb") as f: f.write(_fasta_bytes(seqs[i], f"genome{i}"))
    with ThreadPoolExecutor(64) as ex: list(ex.map(write, range(n)))
paths = [os.path.join(d, f"g{i:05d}.fna") for i in range(n)]
ts = []
for rep in range(4):
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=64)
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=64)
    t0 = time.perf_counter(); c = galah_amd.cluster(paths, pre, cl); ts.append(time.perf_counter() - t0)
print("cluster() from files:", " ".join(f"{t*1e3:.0f}" for t in ts), "ms;", len(c), "clusters")
PY
done; done
rm -rf /dev/shm/ghip_probe2
