"""Pair stage (join form) on N synthetic genomes, for rocprofv3 --kernel-trace --stats.  usage: join_profile.py [species=800] [length=100000] [reps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
species = int(sys.argv[1]) if len(sys.argv) > 1 else 800
length = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, species, 10, length, 0.0253)
sk = ctx.sketch_genomes(g, 21, 1000, 0)
ctx.precluster(sk, np.float32(0.9))
ctx.profile(True); ctx.profile_reset()
t = time.perf_counter()
for _ in range(reps):
    p = ctx.precluster(sk, np.float32(0.9))
dt = (time.perf_counter() - t) / reps
print(f"N={species * 10}: precluster {dt * 1e3:.2f} ms/call, {len(p)} pairs;", {k: round(ms / n, 3) for k, (n, ms) in ctx.kernel_stats().items() if n})
