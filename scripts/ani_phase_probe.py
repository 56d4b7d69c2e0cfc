"""Cycle counters around the phases of ani_pairs (build variant `phases` of scripts/ani_variants.sh): one launch of `m` pairs
of the bench workload, counters of wave 0 of pair 0 printed by the library.  usage: ani_phase_probe.py [m ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, 100, 10, 5_000_000, 0.0253)
sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
pairs = ctx.precluster(sk, np.float32(0.9))
pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
for m in [int(a) for a in sys.argv[1:]] or [8, 768, 4500]:
    sys.stderr.write("-- %d pairs (counters are cumulative)\n" % m)
    ctx.ani_pairs(idx, pi[:m], 0.15)
