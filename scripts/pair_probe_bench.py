"""pair_intersect_tile (dense probe form) on BASELINE configs[1]'s sketch matrix: 1 000 sketches, s = 1000, all 499 500 pairs.
Random sketches with planted families stand in for genomes (the kernel's time does not depend on where the hashes came from)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
os.environ["GHIP_PAIR_KERNEL"] = "probe"
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, n // 10, 10, 100_000, 0.0253)
sk = ctx.sketch_genomes(g, 21, 1000, 0)
p = ctx.precluster(sk, np.float32(0.9))
ctx.profile(True); ctx.profile_reset()
for _ in range(10):
    p2 = ctx.precluster(sk, np.float32(0.9))
ctx.profile(False)
assert p2.tobytes() == p.tobytes()
st = ctx.kernel_stats()
nl, ms = st["pair_intersect_tile"]
print("pair_intersect_tile n=%d: %.3f ms per launch (%d pairs listed) = %.3e pairs/s" % (n, ms / nl, len(p), n * (n - 1) / 2 / (ms / nl * 1e-3)))
