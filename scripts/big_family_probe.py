"""Pair stage on a collection with ONE very large family (a hash shared by more genomes than an element bucket of the join
holds): the hybrid (join + a dense pass over the family's rows) against the dense form over everything.
usage: big_family_probe.py [n=50000] [family=1500]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
s = 1000
rng = np.random.default_rng(5)
hashes = np.sort(rng.integers(0, 2**63, size=(n, s), dtype=np.uint64), axis=1)
groups = (n - fam) // 10
pool = np.sort(rng.integers(0, 2**63, size=(groups, 600), dtype=np.uint64), axis=1)
for i in range(fam, n):          # small families of ten sharing 600 of their 1000 hashes
    hashes[i, :600] = pool[(i - fam) % groups]
core = rng.integers(0, 2**63, size=600, dtype=np.uint64)
hashes[:fam, :600] = core        # the large family
hashes = np.sort(hashes, axis=1)
lens = np.full(n, s, dtype=np.uint32)
ctx = galah_amd.Context(0)
sk = ctx.sketches_from_host(hashes, lens, 21)
for form in ("default", "probe"):
    if form == "probe": ctx.set_options(pair_form="probe")
    p = ctx.precluster(sk, np.float32(0.9))
    ctx.synchronize(); t0 = time.perf_counter(); p = ctx.precluster(sk, np.float32(0.9)); ctx.synchronize()
    print("%-8s n=%d family=%d: %d pairs listed in %.1f ms" % (form, n, fam, len(p), (time.perf_counter() - t0) * 1e3))
    if form == "default": ref = p
assert ref.tobytes() == p.tobytes()
