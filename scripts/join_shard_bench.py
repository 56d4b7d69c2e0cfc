"""Pair stage at BASELINE configs[2] size (10 000 genomes, s = 1000) on ONE GPU, timed as each of 8 ranks would run it:
  join, whole            the inverted-index form over all pairs (what --gpus 1 and GHIP_JOIN_RANKS=replicate run)
  join, hash share r of 8  the default multi-rank form (ghip_precluster_comm): rank r partitions the hashes whose first-level
                         digit is r mod 8, the per-pair partial counts of all ranks are exchanged, rank r finishes the pairs
                         with (i + j) mod 8 == r.  Each rank is run ALONE on the GPU through a host-callback communicator
                         whose all-gather replays what the other ranks contributed in a recording pass -- so the time is
                         one rank's stage 1 + stage 2 + the staging of the callback transport, without seven neighbours
                         on the same device
  join, share r of 8     records emitted only for pairs with (i + j) mod 8 == r, element stage replicated (round 2's form;
                         GHIP_JOIN_RANKS=records)
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
ctx.set_options(pair_form="join")
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
# ---- the hash-sharded form, one rank at a time
import ctypes as C
from galah_amd import _lib
from galah_amd._lib import PAIR_DTYPE
Lb = _lib.lib()
W = 8
sent = [[] for _ in range(W)]   # per rank: the blocks it contributed to its collectives, in call order


cb_time = [0.0]


def run_rank(r, replay):
    calls = [0]

    def allgather(_user, send, nbytes, recv):
        t_in = time.perf_counter()
        try:
            return _allgather(send, nbytes, recv)
        finally:
            cb_time[0] += time.perf_counter() - t_in

    def _allgather(send, nbytes, recv):
        k = calls[0]
        calls[0] += 1
        mine = C.string_at(send, nbytes)
        if not replay:
            sent[r].append(mine)
            # stand-ins: only this rank's own contribution is real (the entry blocks of the others are padding, so that
            # no pair's count exceeds its true common)
            blocks = [mine] * W if k == 0 else [b"\xff" * nbytes] * W
            blocks[r] = mine
        else:
            blocks = []
            for q in range(W):
                b = sent[q][k]
                if k == 1:                                       # the entry blocks: re-pad the recorded one to this call's block size
                    real = np.frombuffer(sent[q][0], dtype=np.uint64)[2]
                    b = b[: int(real) * 16] + b"\xff" * (nbytes - int(real) * 16)
                blocks.append(b)
            blocks[r] = mine
        C.memmove(recv, b"".join(blocks), nbytes * W)
        return 0

    cb = _lib.ALLGATHER_FN(allgather)
    h = C.c_void_p()
    _lib.check(Lb.ghip_comm_init_callback(ctx._h, r, W, C.cast(cb, C.c_void_p), None, C.byref(h)))
    try:
        def once():
            calls[0] = 0
            cb_time[0] = 0.0
            p, n_, rep = C.c_void_p(), C.c_size_t(0), C.c_int(0)
            _lib.check(Lb.ghip_precluster_comm(h, sk._h, np.float32(0.9), C.byref(p), C.byref(n_), C.byref(rep)), ctx._h)
            return ctx._take_pairs(p, n_)
        if not replay:
            return once(), 0.0, 0.0
        # best of 3 by the time spent in the library: wall minus the Python stand-in for the collective
        out, t, tl = None, 1e9, 1e9
        for _ in range(3):
            ctx.synchronize(); t0 = time.perf_counter(); out = once(); ctx.synchronize(); w = time.perf_counter() - t0
            if w - cb_time[0] < tl: t, tl = w, w - cb_time[0]
        return out, t * 1e3, tl * 1e3
    finally:
        Lb.ghip_comm_destroy(h)


for r in range(W):
    run_rank(r, False)
hshares, ts, tls = [], [], []
for r in range(W):
    p, t, tl = run_rank(r, True)
    hshares.append(p); ts.append(t); tls.append(tl)
res["join_hash_share_of_8_ms"] = {"min": min(tls), "max": max(tls), "per_rank": tls,
                                  "what": "stage 1 + stage 2 + host-callback staging of one rank, the Python stand-in for the all-gather excluded "
                                          "(with it: per_rank_wall_ms); on RCCL / peer copies the exchange is two collectives of 24 B and ~0.7 MB per rank",
                                  "per_rank_wall_ms": ts,
                                  "entries_per_rank": [int(np.frombuffer(sent[q][0], dtype=np.uint64)[2]) for q in range(W)]}
allp = np.concatenate(hshares)
key = (allp["i"].astype(np.uint64) << np.uint64(32)) | allp["j"].astype(np.uint64)
assert allp[np.argsort(key, kind="stable")].tobytes() == whole.tobytes(), "hash-sharded shares do not add up to the whole list"
ctx.set_options(pair_form="probe")
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
ctx.set_options(pair_form="auto")
print(json.dumps(res))
