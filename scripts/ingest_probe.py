"""Where the files -> HBM time goes: writes 1 000 synthetic 5 Mb FASTA files to /dev/shm, then times ghip_genomes_from_files
for several thread counts (GHIP_INGEST_DEBUG prints the thread-second breakdown) next to the plain H2D rate of this box."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _fasta_bytes

n, L = 1000, 5_000_000
ctx = galah_amd.Context(0)
d = tempfile.mkdtemp(prefix="ghip_probe_", dir="/dev/shm")
g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)
seqs = [g.to_host(i) for i in range(n)]
del g
def write(i):
    p = os.path.join(d, f"g{i:05d}.fna")
    with open(p, "wb") as f: f.write(_fasta_bytes(seqs[i], f"genome{i}"))
    return p
with ThreadPoolExecutor(64) as ex: paths = list(ex.map(write, range(n)))
del seqs
total = sum(os.path.getsize(p) for p in paths)
# reference: H2D of the same volume, pinned and pageable, one big copy and 1000 small ones
host = torch.empty(total, dtype=torch.uint8).pin_memory(); dev = torch.empty(total, dtype=torch.uint8, device="cuda")
for name, src in (("pinned", host), ("pageable", torch.empty(total, dtype=torch.uint8))):
    src.fill_(65); torch.cuda.synchronize()
    t0 = time.perf_counter(); dev.copy_(src, non_blocking=True); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"H2D one {total/1e9:.2f} GB copy, {name}: {t*1e3:.1f} ms = {total/t/1e9:.1f} GB/s")
del host, dev
torch.cuda.empty_cache()
os.environ["GHIP_INGEST_DEBUG"] = "1"
for T in (16, 32, 64, 128):
    for rep in range(2):
        t0 = time.perf_counter(); gg = ctx.genomes_from_files(paths, T); t = time.perf_counter() - t0
        gg.free()
    print(f"threads {T}: ingest {t*1e3:.1f} ms = {total/t/1e9:.1f} GB/s", flush=True)
shutil.rmtree(d)
