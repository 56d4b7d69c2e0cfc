"""Sketch-stage micro benchmark on synthetic genomes: MinHash only, fused MinHash + seeds, standalone seeds
(kernel time from HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(7, n // 10, 10, L, 0.02)
for _ in range(4):   # let the clocks settle: the first measurements of a process run ~8 % slow
    ctx.sketch_and_index(g, 21, 1000, 0)
for name, fn in (("minhash", lambda: ctx.sketch_genomes(g, 21, 1000, 0)),
                 ("fused", lambda: ctx.sketch_and_index(g, 21, 1000, 0)),
                 ("seeds", lambda: ctx.ani_index_build(g))):

    fn()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5):
        r = fn()
    ctx.profile(False)
    st = {k: ms / nl for k, (nl, ms) in ctx.kernel_stats().items() if nl}
    main = st.get("sketch_kmers", st.get("ani_seeds", 1.0))
    print(name, {k: round(v, 3) for k, v in st.items()}, "k-mer pass: %.1f Gbases/s" % (n * L / (main * 1e-3) / 1e9))
