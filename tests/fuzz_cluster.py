"""Randomised differential test of the host clusterer (ghip_cluster; no GPU) against the oracle's literal
restatement of src/clusterer.rs, including tied precluster ANIs, tied clusterer ANIs, zeros and None values
(python tests/fuzz_cluster.py [rounds=300] [seed=1])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from galah_amd import PAIR_DTYPE, cluster_pairs  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for r in range(rounds):
    n = int(rng.integers(1, 90))
    groups = int(rng.integers(1, 9))
    density = rng.uniform(0.05, 1.0)
    rows = []
    for i in range(n):
        for j in range(i + 1, n):
            same = (i % groups) == (j % groups)
            if rng.random() < (density if same else density * rng.choice([0.0, 0.02, 0.3])):
                pre = np.float32(rng.choice([0.9, 0.95, 0.97]) if rng.random() < 0.4 else rng.uniform(0.9, 1.0))  # tied precluster ANIs
                rows.append((i, j, 0, 0, pre))
    pairs = np.array(rows, dtype=PAIR_DTYPE) if rows else np.zeros(0, dtype=PAIR_DTYPE)
    vals = np.round(rng.uniform(93, 100, size=(n, n)), int(rng.integers(0, 3))).astype(np.float32)   # many ties at 0 decimals
    vals = np.minimum(vals, vals.T)
    vals[rng.random((n, n)) < 0.08] = 0.0
    vals = np.minimum(vals, vals.T)
    thr = np.float32(rng.choice([95.0, 97.0, 99.5]))
    want = oracle.cluster(n, oracle.Cache.from_pairs(pairs), thr, lambda a, b: float(vals[a, b]))
    pair_ani = np.array([vals[p["i"], p["j"]] for p in pairs], dtype=np.float32)
    got = cluster_pairs(n, pairs, thr, pair_ani)
    assert got == want, (r, n, "batched")
    got = cluster_pairs(n, pairs, thr, None, False, ani_callback=lambda a, b: float(vals[a, b]))
    assert got == want, (r, n, "callback")
    thr2 = np.float32(rng.choice([0.9, 0.95, 0.97]))
    assert cluster_pairs(n, pairs, thr2, None, True) == oracle.cluster(n, oracle.Cache.from_pairs(pairs), thr2, None, True), (r, "skip")
print(f"fuzz ok: {rounds} rounds")
