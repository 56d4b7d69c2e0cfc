"""Pair stage at BASELINE configs[2] size (10 000 genomes, s = 1000) on ONE GPU, timed as each of 8 ranks would run it:
  join, whole            the inverted-index form over all pairs (what --gpus 1 and GHIP_JOIN_RANKS=replicate run)
  join, share r of 8     records emitted only for pairs with (i + j) mod 8 == r (the default multi-rank form)
  dense probe, tile share the N^2/2 kernel dealt by tile (what north_star's wording describes)
Times are wall ms of ghip_precluster_shard (kernels + host recheck + sort), best of 3; the candidate gather that the
sharded forms need afterwards is timed by bench.py's stage `allgather_pairs`.  usage: join_shard_bench.py [n=10000] [len=100000]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import galah_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
ctx = galah_amd.Context(0)
g = ctx.genomes_synthetic(42, n // 10, 10, L, 0.0253)
sk = ctx.sketch_genomes(g, 21, 1000, 0)
g.free()

def best(fn, reps=3):
    out, t = None, 1e9
    for _ in range(reps):
        ctx.synchronize(); t0 = time.perf_counter(); out = fn(); ctx.synchronize(); t = min(t, time.perf_counter() - t0)
    return out, t * 1e3

res = {"n": n, "pairs": n * (n - 1) // 2}
os.environ["GHIP_PAIR_KERNEL"] = "join"
whole, t = best(lambda: ctx.precluster(sk, np.float32(0.9)))
res["join_whole_ms"] = t; res["candidates"] = len(whole)
shares, ts = [], []
for r in range(8):
    p, t = best(lambda: ctx.precluster(sk, np.float32(0.9), r, 8))
    shares.append(p); ts.append(t)
res["join_share_of_8_ms"] = {"min": min(ts), "max": max(ts), "per_rank": ts}
allp = np.concatenate(shares)
key = (allp["i"].astype(np.uint64) << np.uint64(32)) | allp["j"].astype(np.uint64)
assert allp[np.argsort(key, kind="stable")].tobytes() == whole.tobytes(), "shares do not add up to the whole list"
os.environ["GHIP_PAIR_KERNEL"] = "probe"
_, t = best(lambda: ctx.precluster(sk, np.float32(0.9)), 2)
res["dense_whole_ms"] = t
ts = []
dshares = []
for r in range(8):
    p, t = best(lambda: ctx.precluster(sk, np.float32(0.9), r, 8), 2)
    ts.append(t); dshares.append(p)
res["dense_tile_share_of_8_ms"] = {"min": min(ts), "max": max(ts)}
allp = np.concatenate(dshares)
key = (allp["i"].astype(np.uint64) << np.uint64(32)) | allp["j"].astype(np.uint64)
assert allp[np.argsort(key, kind="stable")].tobytes() == whole.tobytes()
del os.environ["GHIP_PAIR_KERNEL"]
print(json.dumps(res))
