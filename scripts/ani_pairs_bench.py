"""ani_pairs kernel time on the bench workload (1 000 x 5 Mb, 4 500 precluster pairs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, 100, 10, 5_000_000, 0.0253)
sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
pairs = ctx.precluster(sk, np.float32(0.9))
pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
out = []
for m in (len(pi), 1536, 384, 96, 8):
    ctx.ani_pairs(idx, pi[:m], 0.15)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5):
        ctx.ani_pairs(idx, pi[:m], 0.15)
    ctx.profile(False)
    nl, ms = ctx.kernel_stats()["ani_pairs"]
    out.append("%d pairs: %.3f ms" % (m, ms / nl))
print("ani_pairs " + "   ".join(out))
