"""Accuracy of the build-defined ANI estimator on genome pairs with STRUCTURE, against the counted identity of their
orthologous bases (CPU, oracle only -- the device reproduces the oracle bit for bit, tests/fuzz_ani.py):
  plain            substitutions only
  repeats          + a 1.5 kb insertion-sequence family, 30 copies scattered through both genomes
  island           + a foreign 200 kb island in one genome (no counterpart in the other)
  rearranged       + 12 segments of 50-300 kb inverted or moved in one genome
  fragmented       + both genomes cut into ~150 contigs, shuffled, half of them reverse-complemented (a MAG)
  plasmid only     two UNRELATED genomes sharing one 100 kb element (expected: 0, the aligned-fraction gate)
Columns: previous estimator (seeds matched by value against the whole other genome, analytic chance term) and the
current one (matches must be colinear inside a 20 kb chunk).  usage: ani_scenarios.py [length=2000000] [pairs=6] [old_lib.so]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 6
OLD = sys.argv[3] if len(sys.argv) > 3 else None
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


class OldOracle:
    def __init__(self, path):
        self.L = C.CDLL(path)
        self.L.go_ani_sketch_bytes.restype = C.c_void_p
        self.L.go_ani_sketch_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]
        self.L.go_ani_pair.restype = C.c_float
        self.L.go_ani_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]

    def ani(self, a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        sa = self.L.go_ani_sketch_bytes(a.ctypes.data, a.size, 15, 125, 20000)
        sb = self.L.go_ani_sketch_bytes(b.ctypes.data, b.size, 15, 125, 20000)
        return float(self.L.go_ani_pair(sa, sb, np.float32(0.15), None, None))


old = OldOracle(OLD) if OLD else None


def new_ani(a, b):
    return oracle.ani_pair(oracle.AniSketch.from_bytes(a), oracle.AniSketch.from_bytes(b), 0.15)[0]


def substitute(rng, anc, rate):
    s = anc.copy()
    hit = rng.random(len(s)) < rate
    s[hit] = (np.searchsorted(acgt, s[hit]) + rng.integers(1, 4, size=int(hit.sum()))) % 4
    s[hit] = acgt[s[hit]]
    return s


def revcomp(s):
    return COMP[s[::-1]]


def scenario(rng, name, rate):
    anc = rng.choice(acgt, size=L)
    a, b = substitute(rng, anc, rate), substitute(rng, anc, rate)
    true = 100.0 * float(np.mean(a == b))   # identity of the orthologous bases (what ANI means)
    if name in ("repeats", "island", "rearranged", "fragmented"):
        ins = rng.choice(acgt, size=1500)
        for g in (0, 1):
            s = a if g == 0 else b
            for p in sorted(rng.integers(0, len(s), size=30).tolist(), reverse=True):
                s = np.concatenate([s[:p], substitute(rng, ins, 0.01), s[p:]])
            if g == 0: a = s
            else: b = s
    if name in ("island", "rearranged", "fragmented"):
        p = int(rng.integers(0, len(a)))
        a = np.concatenate([a[:p], rng.choice(acgt, size=200_000), a[p:]])
    if name in ("rearranged", "fragmented"):
        for _ in range(12):
            w = int(rng.integers(50_000, 300_000))
            p = int(rng.integers(0, len(b) - w))
            seg = b[p:p + w]
            rest = np.concatenate([b[:p], b[p + w:]])
            if rng.random() < 0.5:
                seg = revcomp(seg)
            t = int(rng.integers(0, len(rest))) if rng.random() < 0.5 else p
            b = np.concatenate([rest[:t], seg, rest[t:]])
    if name == "fragmented":
        def mag(s):
            cuts = np.sort(rng.choice(np.arange(1, len(s)), size=150, replace=False))
            parts = np.split(s, cuts)
            order = rng.permutation(len(parts))
            out = []
            for i in order:
                out.append(revcomp(parts[i]) if rng.random() < 0.5 else parts[i])
                out.append(np.frombuffer(b"N", dtype=np.uint8))
            return np.concatenate(out)
        a, b = mag(a), mag(b)
    return a, b, true


print(f"{'scenario':<14}{'true %':>8}{'old mean':>10}{'old max|e|':>11}{'new mean':>10}{'new max|e|':>11}   ({S} pairs of {L} bp, substitution rate 0.0253 per copy)")
for name in ("plain", "repeats", "island", "rearranged", "fragmented"):
    rng = np.random.default_rng(99)
    tr, eo, en = [], [], []
    for _ in range(S):
        a, b, true = scenario(rng, name, 0.0253)
        tr.append(true)
        en.append(new_ani(a, b) - true)
        if old: eo.append(old.ani(a, b) - true)
    o1 = f"{np.mean(eo):+10.3f}{np.max(np.abs(eo)):11.3f}" if old else f"{'-':>10}{'-':>11}"
    print(f"{name:<14}{np.mean(tr):8.3f}{o1}{np.mean(en):+10.3f}{np.max(np.abs(en)):11.3f}")
rng = np.random.default_rng(5)
res_o, res_n = [], []
for _ in range(S):
    a, b = rng.choice(acgt, size=L), rng.choice(acgt, size=L)
    pl = rng.choice(acgt, size=100_000)
    a = np.concatenate([a[:L // 3], pl, a[L // 3:]])
    b = np.concatenate([b[:L // 2], substitute(rng, pl, 0.001), b[L // 2:]])
    res_n.append(new_ani(a, b))
    if old: res_o.append(old.ani(a, b))
print(f"{'plasmid only':<14}{'(0)':>8}{(str(max(res_o)) if old else '-'):>10}{'':>11}{max(res_n):>10}{'':>11}   largest value reported over {S} unrelated pairs")
