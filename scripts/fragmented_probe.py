"""Fused sketch pass on FRAGMENTED genomes: the same 5 Mb of bases cut into contigs of a given length (one invalid position --
the record separator -- between contigs, as the ingest lays records out), against the unbroken genome.  The careful variant
of sketch_kmers21 runs for every 16-position group in which some lane of the wave sees an invalid base.
usage: fragmented_probe.py [n_genomes=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = galah_amd.Context(0)
g0 = ctx.genomes_synthetic(3, n // 10, 10, 5_000_000, 0.02)
base = [g0.to_host(i) for i in range(n)]
for contig in (0, 100_000, 20_000, 5_000, 1_000):
    if contig:
        streams = []
        for b in base:
            k = len(b) // contig
            a = b[: k * contig].reshape(k, contig)
            streams.append(np.concatenate([a, np.full((k, 1), ord("N"), np.uint8)], axis=1).reshape(-1))
    else:
        streams = base
    g = ctx.genomes_from_host(streams)
    for _ in range(3): ctx.sketch_and_index(g, 21, 1000, 0)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5): ctx.sketch_and_index(g, 21, 1000, 0)
    ctx.profile(False)
    nl, ms = ctx.kernel_stats()["sketch_kmers"]
    print("contigs of %7s bases: sketch_kmers %.3f ms per %d x 5 Mb = %.0f Gbases/s" % (contig or "whole", ms / nl, n, n * 5e6 / (ms / nl * 1e-3) / 1e9))
    g.free()
