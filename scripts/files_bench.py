"""Wall-clock of `cluster` from FASTA files on disk to clusters (BASELINE metric 2) on synthetic genomes.
Files are written from the device generator (no oracle), 80 columns per line; then
galah_amd.cluster(paths, FinchPreclusterer, HipAniClusterer) is timed, split into host ingest and GPU work.
usage: files_bench.py [n_genomes=200] [length=5000000] [io_threads=64] [gz=0]"""
import gzip, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
from concurrent.futures import ThreadPoolExecutor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
gz = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
ctx = galah_amd.Context(0)
d = tempfile.mkdtemp(prefix="ghip_files_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)

def write(i):
    seq = g.to_host(i)
    pad = (-len(seq)) % 80
    body = np.concatenate([seq, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 80)
    body = np.concatenate([body, np.full((body.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes()
    body = body[: len(body) - pad - 1] + b"\n" if pad else body
    p = os.path.join(d, f"g{i:05d}.fna" + (".gz" if gz else ""))
    data = f">genome{i} synthetic\n".encode() + body
    with (gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) as f:
        f.write(data)
    return p

t0 = time.perf_counter()
paths = [write(i) for i in range(n)]   # to_host goes through the one context: keep it serial
print(f"wrote {n} files ({n * L / 1e9:.2f} GB{' gz' if gz else ''}) in {time.perf_counter() - t0:.1f}s")
del g

thread_list = [int(x) for x in os.environ.get("GHIP_FILES_BENCH_THREADS", str(T)).split(",")]
for rep in range(int(os.environ.get("GHIP_FILES_BENCH_REPS", "2")) * len(thread_list)):
    T = thread_list[rep % len(thread_list)]
    if os.environ.get("GHIP_FILES_BENCH_PLAIN_CAPS"):   # sweep the reader cap for plain files
        caps = os.environ["GHIP_FILES_BENCH_PLAIN_CAPS"].split(",")
        ctx.set_options(io_threads_plain=int(caps[rep % len(caps)]))
        print("plain cap", caps[rep % len(caps)], end="  ")
    if os.environ.get("GHIP_FILES_BENCH_ALTERNATE"):   # A/B of the batch pipeline inside one process
        ctx.set_options(pipeline_pieces=1 if rep % 2 == 0 else 0)
        print("pipeline_pieces =", 1 if rep % 2 == 0 else 0, end="  ")
    t0 = time.perf_counter()
    gg = ctx.genomes_from_files(paths, T)
    t_ingest = time.perf_counter() - t0
    gg.free()
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=T)
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=T)
    t0 = time.perf_counter()
    clusters = galah_amd.cluster(paths, pre, cl)
    t_all = time.perf_counter() - t0
    print(f"run {rep}: cluster() from files {t_all:.3f}s for {n} genomes -> {len(clusters)} clusters "
          f"({n * (n - 1) // 2 / t_all:.3e} genome-pairs/s end to end); ingest alone (parse + H2D, {T} threads) {t_ingest:.3f}s "
          f"= {n * L / t_ingest / 1e9:.2f} GB/s; everything after ingest {t_all - t_ingest:.3f}s")
import shutil; shutil.rmtree(d)
