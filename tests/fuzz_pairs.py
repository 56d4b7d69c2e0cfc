"""Randomised differential test of the three forms of the pair stage (probe / merge / join) against the oracle's pair
loop, whole and sharded (run by hand on the GPU box: python tests/fuzz_pairs.py [rounds=60] [seed=1])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import galah_amd  # noqa: E402
import oracle  # noqa: E402
from conftest import random_sketches  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = galah_amd.Context(0)
done = 0
for r in range(rounds):
    n = int(rng.choice([2, 3, 9, 33, 100, 257, 400]))
    s = int(rng.choice([8, 64, 256, 1000, 1024, 2000]))
    groups = int(rng.choice([0, 1, 3, 20]))
    min_len = None if rng.random() < 0.4 else int(rng.integers(0 if rng.random() < 0.2 else 1, s + 1))
    hashes, lens = random_sketches(rng, n, s, shared_groups=groups, min_len=min_len)
    if rng.random() < 0.3 and n > 4:                      # duplicates and prefixes
        hashes[1], lens[1] = hashes[0].copy(), lens[0]
        keep = max(int(lens[2]) // 2, 0)
        hashes[3] = np.uint64(0xFFFFFFFFFFFFFFFF); hashes[3, :keep] = hashes[2, :keep]; lens[3] = keep
    thr = np.float32(rng.choice([0.0, 0.5, 0.9, 0.99]))
    want = oracle.distances_from_sketches(hashes, lens, thr, threads=16)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    for form in ("probe", "merge", "join"):
        ctx.set_options(pair_form=form)
        got = ctx.precluster(sk, thr)
        assert got.tobytes() == want.tobytes(), (r, form, n, s, groups, min_len, float(thr))
        world = int(rng.choice([2, 3, 5]))
        parts, compared = [], 0
        for rk in range(world):
            parts.append(ctx.precluster(sk, thr, rk, world))
            compared += ctx.last_pairs_compared
        merged = np.sort(np.concatenate(parts), order=["i", "j"])
        assert merged.tobytes() == want.tobytes(), (r, form, "sharded", world)
        assert compared == n * (n - 1) // 2, (r, form, compared)
        done += 1
    sk.free()
print(f"fuzz ok: {rounds} rounds, {done} (matrix, form) combinations")
