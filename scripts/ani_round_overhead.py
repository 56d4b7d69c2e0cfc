"""What the ANI stage spends OUTSIDE its kernel: ghip_cluster_index's time in ANI rounds against the ani_pairs kernel time of
the same rounds, and one ghip_ani_pairs call of a round's size, at configs[4]'s pair counts.
usage: ani_round_overhead.py [species=5000] [members=10] [length=200000]"""
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
thr = np.float32(95.0)
for rep in range(4):
    t0 = time.perf_counter()
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
    t1 = time.perf_counter()
    pairs = ctx.precluster(sk, np.float32(0.9))
    t2 = time.perf_counter()
    ctx.profile(True); ctx.profile_reset()
    c, st = ctx.cluster_index(idx, n, pairs, thr, 0.15, None)
    t3 = time.perf_counter()
    ks = ctx.kernel_stats(); ctx.profile(False)
    nl, kms = ks.get("ani_pairs", (0, 0.0))
    bl, bms = ks.get("ani_bin", (0, 0.0))
    print("rep %d: sketch+index %.1f ms, precluster %.1f ms, cluster_index %.1f ms: asked %d in %d rounds, in ANI %.2f ms of which kernel %.2f ms (%d launches) -> %.2f ms outside the kernel; ani_bin seen here %.2f"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, st["asked"], st["rounds"], st["ani_ms"], kms, nl, st["ani_ms"] - kms, bms))
    pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)[: st["asked"] // max(st["rounds"], 1)]
    ctx.profile(True); ctx.profile_reset()
    a0 = time.perf_counter(); ctx.ani_pairs(idx, pi, 0.15); a1 = time.perf_counter()
    ks = ctx.kernel_stats(); ctx.profile(False)
    print("       one ghip_ani_pairs of %d pairs: %.2f ms, kernel %.2f ms" % (len(pi), (a1 - a0) * 1e3, ks["ani_pairs"][1]))
    idx.free(); sk.free()
