"""Randomised differential test of the ANI stage (ani_seeds / ani_bin / ani_pairs + host finalisation) against the
oracle on RELATED genomes: families grown from a random ancestor by substitutions, insertions, deletions, segment
shuffles, repeats and runs of N, over random k / seed density / chunk length / aligned-fraction gates.  ANI and both
aligned fractions must be bit-identical for every ordered pair.  Run by hand on the GPU box:
python tests/fuzz_ani.py [rounds=30] [seed=1].  Not collected by pytest; test_gpu_parity.py holds the fixed cases."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conftest  # noqa: E402,F401  (the test harness' emulator switch, tests/conftest.py: GALAH_TEST_EMU)
import galah_amd  # noqa: E402
import oracle  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = galah_amd.Context(0)
if "--tall-below" in sys.argv:   # ghip_options.ani_tall_below: 0 = every launch in the 8-wave shape
    ctx.set_options(ani_tall_below=int(sys.argv[sys.argv.index("--tall-below") + 1]))
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(anc: np.ndarray) -> np.ndarray:
    s = anc.copy()
    L = len(s)
    if L == 0:
        return s
    rate = float(rng.choice([0.0, 0.001, 0.01, 0.03, 0.05, 0.08, 0.15]))
    hit = rng.random(L) < rate
    s[hit] = rng.choice(acgt, size=int(hit.sum()))
    for _ in range(int(rng.integers(0, 4))):            # indels move everything behind them across chunk borders
        p = int(rng.integers(0, len(s) + 1))
        if rng.random() < 0.5:
            s = np.concatenate([s[:p], rng.choice(acgt, size=int(rng.integers(1, 3000))), s[p:]])
        else:
            s = np.concatenate([s[:p], s[p + int(rng.integers(1, 3000)):]])
    if rng.random() < 0.3 and len(s) > 2000:               # segment shuffle (rearrangement)
        cuts = np.sort(rng.integers(0, len(s), size=3))
        parts = [s[:cuts[0]], s[cuts[0]:cuts[1]], s[cuts[1]:cuts[2]], s[cuts[2]:]]
        s = np.concatenate([parts[i] for i in rng.permutation(4)])
    if rng.random() < 0.3 and len(s) > 500:                # a repeat: the same segment several times
        p, w = int(rng.integers(0, len(s) - 400)), int(rng.integers(50, 400))
        s = np.concatenate([s, np.tile(s[p:p + w], int(rng.integers(2, 20)))])
    if rng.random() < 0.4 and len(s) > 100:                # contig breaks / runs of N
        for _ in range(int(rng.integers(1, 6))):
            p = int(rng.integers(0, len(s)))
            s[p:p + int(rng.choice([1, 1, 10, 500]))] = ord("N")
    if rng.random() < 0.2 and len(s) > 1000:               # a fragment only
        p = int(rng.integers(0, len(s) // 2))
        s = s[p:p + int(rng.integers(200, len(s) // 2 + 200))]
    return np.ascontiguousarray(s)


checked = 0
for r in range(rounds):
    k = int(rng.choice([15, 15, 15, 16, 12, 9]))
    c = int(rng.choice([1, 5, 30, 125, 125]))
    chunk = int(rng.choice([500, 2000, 20000, 20000]))
    min_af = float(rng.choice([0.0, 0.15, 0.5, 0.9]))
    streams = []
    for fam in range(int(rng.integers(1, 4))):
        L = int(rng.choice([0, 8, 40, 700, 5000, 20000, 19999, 40001, 120000, 300000]))
        anc = rng.choice(acgt, size=L)
        for _ in range(int(rng.integers(1, 5))):
            streams.append(mutate(anc))
    if rng.random() < 0.3:
        streams.append(streams[0].copy())                  # an identical copy
    n = len(streams)
    g = ctx.genomes_from_host(streams)
    idx = ctx.ani_index_build(g, k, c, chunk)
    pairs = np.array([(i, j) for i in range(n) for j in range(n)], dtype=np.uint32)   # a genome with itself too
    ani, af = ctx.ani_pairs(idx, pairs, min_af, want_af=True)
    osk = [oracle.AniSketch.from_bytes(st, k, c, chunk) for st in streams]
    for x, (a, b) in enumerate(pairs):
        o, afq, afr = oracle.ani_pair(osk[a], osk[b], min_af)
        assert np.float32(o) == ani[x] and np.float32(afq) == af[x, 0] and np.float32(afr) == af[x, 1], \
            ("ani", r, int(a), int(b), k, c, chunk, min_af, len(streams[a]), len(streams[b]), o, float(ani[x]), afq, afr, af[x].tolist())
        checked += 1
    idx.free(); g.free()
print(f"fuzz ok: {rounds} rounds, {checked} ordered pairs checked")
