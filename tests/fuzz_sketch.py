"""Randomised differential test of the sketch / seeding kernels against the oracle (run by hand on the GPU box:
python tests/fuzz_sketch.py [rounds=40] [seed=1]).  Not collected by pytest; test_gpu_parity.py holds the fixed cases."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conftest  # noqa: E402,F401  (the test harness' emulator switch, tests/conftest.py: GALAH_TEST_EMU)
import galah_amd  # noqa: E402
import oracle  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = galah_amd.Context(0)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
junk = np.frombuffer(b"NN-nacgtRY\x00\xff", dtype=np.uint8)
checked = 0
for r in range(rounds):
    n = int(rng.integers(1, 12))
    streams = []
    for _ in range(n):
        kind = rng.integers(0, 6)
        L = int(rng.choice([0, 1, 20, 21, 22, 63, 64, 65, 84, 85, 16383, 16384, 16385, 16404, 16405, 32768 + 20])) if kind == 0 \
            else int(rng.integers(1, 200_000))
        s = rng.choice(acgt, size=L)
        if kind in (1, 2) and L:
            bad = rng.random(L) < rng.choice([1e-4, 1e-3, 0.02, 0.1])
            s[bad] = rng.choice(junk, size=int(bad.sum()))
        if kind == 3 and L > 100:
            s = np.tile(s[: int(rng.integers(30, 3000))], L // 30 + 1)[:L]      # repetitive: few distinct k-mers
        streams.append(np.ascontiguousarray(s))
    g = ctx.genomes_from_host(streams)
    k = 21 if rng.random() < 0.8 else int(rng.integers(1, 33))
    s_ = int(rng.choice([1, 16, 100, 1000, 4096]))
    seed = 0 if rng.random() < 0.7 else int(rng.integers(1, 2**32))
    c = int(rng.choice([1, 7, 125]))
    chunk = int(rng.choice([1000, 20000]))
    sk, idx = ctx.sketch_and_index(g, k, s_, seed, 15, c, chunk)
    sk2 = ctx.sketch_genomes(g, k, s_, seed)
    h, l = sk.to_host()
    h2, l2 = sk2.to_host()
    assert np.array_equal(h, h2) and np.array_equal(l, l2), ("fused != separate", r)
    glen, cap, cnt = idx.meta()
    for i, st in enumerate(streams):
        want = oracle.sketch_bytes(st, k, s_, seed)
        assert l[i] == len(want) and np.array_equal(h[i, : l[i]], want), ("sketch", r, i, k, s_, seed, len(st))
        o = oracle.AniSketch.from_bytes(st, 15, c, chunk)
        assert cnt[i] == o.nseeds, ("seed count", r, i, c, chunk, len(st), cnt[i], o.nseeds)
        checked += 1
    if n >= 2:
        pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.uint32)
        ani = ctx.ani_pairs(idx, pairs, 0.1)
        osk = [oracle.AniSketch.from_bytes(st, 15, c, chunk) for st in streams]
        for x, (a, b) in enumerate(pairs[:20]):
            assert np.float32(oracle.ani_pair(osk[a], osk[b], 0.1)[0]) == ani[x], ("ani", r, a, b)
    for hnd in (sk, sk2, idx, g):
        hnd.free()
print(f"fuzz ok: {rounds} rounds, {checked} genomes checked")
