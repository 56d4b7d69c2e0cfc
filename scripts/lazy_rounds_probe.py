"""The lazy ANI rounds of one 10 000-genome step: pairs per round, time inside ghip_ani_pairs, kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
from galah_amd.engine import cluster_pairs_lazy
n_species = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, n_species, 10, 5_000_000, 0.0253)
sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
pairs = ctx.precluster(sk, np.float32(0.9))
pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
for rep in range(3):
    rounds = []
    def ani_of(edges):
        t0 = time.perf_counter(); sel = pi[edges]; t1 = time.perf_counter()
        ctx.profile(True); ctx.profile_reset()
        out = ctx.ani_pairs(idx, sel, 0.15)
        ctx.profile(False)
        t2 = time.perf_counter()
        nl, ms = ctx.kernel_stats()["ani_pairs"]
        rounds.append((len(edges), (t1 - t0) * 1e3, (t2 - t1) * 1e3, ms))
        return out
    t0 = time.perf_counter()
    clusters, asked = cluster_pairs_lazy(len(g), pairs, np.float32(95.0), ani_of)
    total = (time.perf_counter() - t0) * 1e3
    inside = sum(r[2] + r[1] for r in rounds)
    print(f"rep {rep}: {len(pairs)} pairs, {asked} asked in {len(rounds)} rounds, total {total:.2f} ms, in callbacks {inside:.2f}, clusterer itself {total - inside:.2f}")
    for r in rounds: print("   pairs %6d  index %.3f ms  ghip_ani_pairs %.3f ms  (kernel %.3f)" % r)
