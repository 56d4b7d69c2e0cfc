"""Where the first files -> clusters call of a process spends its time: each stage timed in a fresh context, first call
and second call.  usage: first_call_probe.py [n_genomes=1000]"""
import os, sys, time, tempfile, shutil
t_import0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
from concurrent.futures import ThreadPoolExecutor
t_import = time.perf_counter() - t_import0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = 5_000_000
d = tempfile.mkdtemp(prefix="ghip_first_", dir="/dev/shm")
rng = np.random.default_rng(1)
def write(i):
    seq = rng.integers(0, 4, size=L, dtype=np.uint8)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[seq]
    body = np.concatenate([seq.reshape(-1, 80), np.full((L // 80, 1), 10, np.uint8)], axis=1).tobytes()
    p = os.path.join(d, f"g{i:05d}.fna")
    with open(p, "wb") as f: f.write(b">g\n" + body)
    return p
base = write(0)
paths = [base]
for i in range(1, n):   # the same bytes under n names: the ingest does not care, the page cache holds one copy each
    p = os.path.join(d, f"g{i:05d}.fna"); shutil.copyfile(base, p); paths.append(p)
os.environ["GHIP_INGEST_DEBUG"] = "1"
t0 = time.perf_counter(); ctx = galah_amd.Context(0); t_ctx = time.perf_counter() - t0
print(f"import {t_import*1e3:.0f} ms, context {t_ctx*1e3:.0f} ms")
if os.environ.get("GHIP_PROBE_PREWARM"):   # a few files first: are the first call's extra 0.2 s per-thread first touches?
    t0 = time.perf_counter(); gg = ctx.genomes_from_files(paths[:int(os.environ["GHIP_PROBE_PREWARM"])], 64); gg.free()
    print(f"prewarm ingest of {os.environ['GHIP_PROBE_PREWARM']} files: {1e3*(time.perf_counter()-t0):.1f} ms")
for call in range(3):
    t = [time.perf_counter()]
    g = ctx.genomes_from_files(paths, 64); t.append(time.perf_counter())
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0); t.append(time.perf_counter())
    pairs = ctx.precluster(sk, np.float32(0.9)); t.append(time.perf_counter())
    pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
    ani = ctx.ani_pairs(idx, pi, 0.15) if len(pi) else None; t.append(time.perf_counter())
    print(f"call {call}: ingest {1e3*(t[1]-t[0]):.1f}  sketch+index {1e3*(t[2]-t[1]):.1f}  precluster {1e3*(t[3]-t[2]):.1f}  ani {1e3*(t[4]-t[3]):.1f} ms  ({len(pairs)} pairs)")
    g.free(); sk.free(); idx.free()
shutil.rmtree(d)
