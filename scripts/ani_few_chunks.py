"""The ANI estimator on records of ONE to TEN chunks (20 kb .. 200 kb), against the counted identity of the pair: the lower
median of the per-chunk containments next to the pooled count, for pooling limits 0 (always the median: the definition up
to round 4), 3, 5 and "always pooled".  CPU only (the oracle is the definition; the device reproduces it bit for bit).
Scenarios: plain substitutions (the device's synthetic generator), indels (1-30 bp, one per ~2 kb), island (a foreign
10 % insertion in one genome).  usage: ani_few_chunks.py [pairs per cell = 48]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
RATES = (0.0005, 0.0025, 0.005, 0.0102, 0.0155, 0.0253, 0.0363, 0.0417, 0.0527, 0.0640)
LENGTHS = tuple(int(x) for x in os.environ.get("LENGTHS", "20000,30000,45000,70000,100000,200000").split(","))
LIMITS = tuple(int(x) for x in os.environ.get("LIMITS", "0,3,5,1073741824").split(","))


def substitute(rng, anc, rate):
    s = anc.copy()
    hit = rng.random(len(s)) < rate
    s[hit] = ACGT[(np.searchsorted(ACGT, s[hit]) + rng.integers(1, 4, size=int(hit.sum()))) % 4]
    return s


def pair(rng, name, rate, L, seed, sp):
    if name == "plain":
        a, b = oracle.synth_genome(seed, sp, 0, L, rate), oracle.synth_genome(seed, sp, 1, L, rate)
        return a, b, 100.0 * float(np.mean(a == b))
    anc = rng.choice(ACGT, size=L)
    a, b = substitute(rng, anc, rate), substitute(rng, anc, rate)
    true = 100.0 * float(np.mean(a == b))
    if name == "island":
        p = int(rng.integers(0, len(a)))
        a = np.concatenate([a[:p], rng.choice(ACGT, size=L // 10), a[p:]])
    if name == "indels":
        out, at = [], 0
        for p in np.sort(rng.integers(0, len(b), size=max(1, len(b) // 2000))):
            if p <= at:
                continue
            out.append(b[at:p])
            w = int(rng.integers(1, 31))
            if rng.random() < 0.5:
                out.append(rng.choice(ACGT, size=w)); at = p
            else:
                at = min(len(b), p + w)
        out.append(b[at:])
        b = np.concatenate(out)
    return a, b, true


for name in ("plain", "indels", "island"):
    print(f"== {name}: mean error (max |error|) in ANI points over {S} independent pairs per cell; columns = pooling limit")
    print(f"{'L':>7} {'true %':>7} " + " ".join(f"{('median' if l == 0 else 'pooled' if l > 99 else 'pool<%d' % l):>16}" for l in LIMITS) + "   chunks")
    for L in LENGTHS:
        worst = {l: [0.0, 0.0] for l in LIMITS}
        for rate in RATES:
            rng = np.random.default_rng(1000 + int(rate * 1e5) + L)
            errs = {l: [] for l in LIMITS}
            truths, nch = [], []
            for sp in range(S):
                a, b, true = pair(rng, name, rate, L, 1234 + L, sp + int(rate * 1e5) * 100)
                qa, qb = oracle.AniSketch.from_bytes(a), oracle.AniSketch.from_bytes(b)
                truths.append(true)
                for l in LIMITS:
                    v, _, _, d = oracle.ani_pair_pool_below(qa, qb, min(l, 0xffffffff), 0.15)
                    errs[l].append(v - true)
                nch.append(d[2])
            cells = []
            for l in LIMITS:
                e = np.array(errs[l])
                cells.append(f"{e.mean():+7.3f} ({np.abs(e).max():5.2f})")
                worst[l][0] = max(worst[l][0], abs(e.mean())); worst[l][1] = max(worst[l][1], np.abs(e).max())
            print(f"{L:7d} {np.mean(truths):7.2f} " + " ".join(f"{c:>16}" for c in cells) + f"   {np.mean(nch):.1f}")
        print(f"{L:7d}   worst " + " ".join(f"{worst[l][0]:7.3f} ({worst[l][1]:5.2f})".rjust(16) for l in LIMITS))
