export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_mirror.py -m gpu -q -x -k "ingest or batched or gunzip or stats or golden or cluster or mirror or files" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import galah_amd
from concurrent.futures import ThreadPoolExecutor
from bench import _fasta_bytes
n, L = 1000, 5_000_000
ctx = galah_amd.Context(0)
d = "/dev/shm/ghip_probe2"
os.makedirs(d, exist_ok=True)
g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)
seqs = [g.to_host(i) for i in range(n)]
del g
def write(i):
    with open(os.path.join(d, f"g{i:05d}.fna"), "wb") as f: f.write(_fasta_bytes(seqs[i], f"genome{i}"))
with ThreadPoolExecutor(64) as ex: list(ex.map(write, range(n)))
PY
for MM in 1 0 1 0; do
echo "== GHIP_INGEST_MMAP=$MM"
GHIP_INGEST_MMAP=$MM GHIP_INGEST_DEBUG=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, sys, time
sys.path.insert(0, os.getcwd())
import galah_amd
n=1000
ctx = galah_amd.Context(0)
d = "/dev/shm/ghip_probe2"
paths = [os.path.join(d, f"g{i:05d}.fna") for i in range(n)]
ts = []
for rep in range(5):
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=64)
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=64)
    t0 = time.perf_counter(); c = galah_amd.cluster(paths, pre, cl); ts.append(time.perf_counter() - t0)
os.environ["GHIP_PIPELINE"]="0"
ti=[]
for rep in range(4):
    t0=time.perf_counter(); gg=ctx.genomes_from_files(paths,64); ti.append(time.perf_counter()-t0); gg.free()
print("cluster() from files:", " ".join(f"{t*1e3:.0f}" for t in ts), "ms; ingest only:", " ".join(f"{t*1e3:.0f}" for t in ti))
PY
done
rm -rf /dev/shm/ghip_probe2
