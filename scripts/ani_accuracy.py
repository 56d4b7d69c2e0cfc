"""Accuracy of the build-defined ANI estimator against the TRUE identity of synthetic genome
pairs: members of one species are independent substitution copies of an ancestor, so the identity of a pair is
measured exactly by comparing their bases.  Every pair comes from its own species (its own ancestor), so the pairs
are independent samples.  usage: ani_accuracy.py [length=2000000] [pairs=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = galah_amd.Context(0)
print(f"{'sub_rate':>9} {'true %':>8} {'est mean':>9} {'bias':>7} {'+-SE':>6} {'max |err|':>9} {'AF':>5}  ({S} independent pairs of {L} bp)")
for rate in (0.0005, 0.0025, 0.005, 0.0102, 0.0155, 0.0253, 0.0363, 0.0417, 0.0527, 0.0640):
    g = ctx.genomes_synthetic(1234, S, 2, L, rate)
    idx = ctx.ani_index_build(g)
    pairs = np.array([(2 * sp, 2 * sp + 1) for sp in range(S)], dtype=np.uint32)
    ani, af = ctx.ani_pairs(idx, pairs, 0.15, want_af=True)
    true = np.array([100.0 * np.mean(g.to_host(int(a)) == g.to_host(int(b))) for a, b in pairs])
    err = ani - true
    print(f"{rate:9.4f} {true.mean():8.3f} {ani.mean():9.3f} {err.mean():+7.3f} {err.std(ddof=1) / np.sqrt(S):6.3f} {np.abs(err).max():9.3f} {af.mean():5.2f}")
    idx.free(); g.free()
