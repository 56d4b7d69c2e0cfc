"""Pair-stage micro benchmark on random sketches (kernel time from HIP events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
s = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(1)
# sorted random sketches whose maxima mimic ~5 Mb genomes; 10-member families share ~35 % of hashes
fam = n // 10 + 1
pools = rng.integers(0, 2**51, size=(fam, s), dtype=np.uint64)
h = np.empty((n, s), dtype=np.uint64)
for i in range(n):
    own = rng.integers(0, 2**51, size=s, dtype=np.uint64)
    mix = np.where(rng.random(s) < 0.35, pools[i // 10], own)
    u = np.unique(mix)
    while len(u) < s:
        u = np.unique(np.concatenate([u, rng.integers(0, 2**51, size=s - len(u), dtype=np.uint64)]))
    h[i] = u[:s]
lens = np.full(n, s, dtype=np.uint32)
ctx = galah_amd.Context(0)
sk = ctx.sketches_from_host(h, lens, 21)
ctx.precluster(sk, np.float32(0.9))
ctx.profile(True); ctx.profile_reset()
for _ in range(5):
    p = ctx.precluster(sk, np.float32(0.9))
ctx.profile(False)
st = ctx.kernel_stats()
nl, ms = st["pair_join"] if st["pair_join"][0] and not st["pair_intersect_tile"][0] else st["pair_intersect_tile"]
nl = max(nl, 1)
pairs = n * (n - 1) // 2
print(f"form={os.environ.get('GHIP_PAIR_KERNEL','auto')} n={n} s={s} hits={len(p)} avg {ms/nl:.3f} ms  {pairs/(ms/nl*1e-3):.3e} pairs/s  {pairs*16*s/(ms/nl*1e-3)/1e9:.0f} GB/s alg")
