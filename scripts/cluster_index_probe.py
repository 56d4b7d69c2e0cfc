"""Clustering stage on one GPU, two drivers of the same lazy rounds: ghip_cluster_index (native: the rounds never leave the
library) and ghip_cluster_lazy with a host-language callback per round (+ the quality-order renumbering in numpy).
usage: cluster_index_probe.py [species=5000] [members=10] [length=200000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
mem = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
n = ns * mem
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, ns, mem, L, 0.0253)
sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
pairs = ctx.precluster(sk, np.float32(0.9))
order = np.random.default_rng(1).permutation(n).astype(np.uint32)
thr = np.float32(95.0)


def host_driven(order):
    t0 = time.perf_counter()
    p = pairs
    pi = np.stack([p["i"], p["j"]], axis=1).astype(np.uint32)
    if order is not None:
        rank_of = np.empty(n, np.uint32); rank_of[order] = np.arange(n, dtype=np.uint32)
        a, b = rank_of[p["i"]], rank_of[p["j"]]
        p = p.copy(); p["i"], p["j"] = np.minimum(a, b), np.maximum(a, b)
        perm = np.lexsort((p["j"], p["i"])); p, pi = p[perm], pi[perm]
    t_ani = [0.0]

    def ani_of(e):
        a0 = time.perf_counter(); out = ctx.ani_pairs(idx, pi[e], 0.15); t_ani[0] += time.perf_counter() - a0
        return out
    c, asked = galah_amd.cluster_pairs_lazy(n, p, thr, ani_of)
    return c, (time.perf_counter() - t0) * 1e3, t_ani[0] * 1e3, asked


def native(order):
    t0 = time.perf_counter()
    c, st = ctx.cluster_index(idx, n, pairs, thr, 0.15, order)
    return c, (time.perf_counter() - t0) * 1e3, st["ani_ms"], st["asked"], st["total_ms"], st["rounds"]


for name, od in (("genome order", None), ("quality order", order)):
    for rep in range(3):
        a = host_driven(od); b = native(od)
        assert a[0] == b[0] and a[3] == b[3]
        print("%-13s n=%d pairs=%d asked=%d rounds=%d | host-driven %.2f ms (ani %.2f) | native %.2f ms (ani %.2f, in library %.2f)"
              % (name, n, len(pairs), a[3], b[5], a[1], a[2], b[1], b[2], b[4]))
