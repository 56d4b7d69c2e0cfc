"""CPU Monte-Carlo behind DESIGN.md section 4 (fused join): does a first-level bucket of the capacity form hold its load?
Elements: N s hash digits -> 256 buckets of capacity mean + 12 sqrt(mean) + 1 024; equal hashes travel together (an ancestral
k-mer of a species is kept by Binomial(10, 0.584) of its members at 95 % identity), which makes the load's variance ~4x a Poisson's.  Records: every record of a genome pair
(i, j) carries the pair's digit (pairs_join.hip: RecSrc::mix), so a bucket's load is a sum of lumps of `common` records ->
capacity 1.5 mean + 8 s + 2 048.  Workload: the bench's generator (species of 10 members, all 45 pairs of a species share
~350 +- 60 hashes).  usage: join_capacity.py [species=1000]"""
import sys

import numpy as np

n_species = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
members, s = 10, 1000
rng = np.random.default_rng(1)


def mix(i, j):
    return ((((i.astype(np.uint64) * 0x9E3779B1) & 0xffffffff) ^ ((j.astype(np.uint64) * 0x85EBCA77) & 0xffffffff)) >> 12).astype(np.uint32)


n = n_species * members
ii, jj = [], []
for a in range(members):
    for b in range(a + 1, members):
        ii.append(np.arange(n_species) * members + a)
        jj.append(np.arange(n_species) * members + b)
i, j = np.concatenate(ii), np.concatenate(jj)
worst = 0.0
for trial in range(20):
    common = rng.normal(350, 60, len(i)).clip(50, s).astype(np.int64)
    load = np.bincount(mix(i, j) >> 12, weights=common, minlength=256)
    mean = common.sum() / 256
    cap = 1.5 * mean + 8 * s + 2048
    worst = max(worst, load.max() / cap)
print(f"records, {n} genomes: mean bucket {mean:.0f}, capacity {cap:.0f}, fullest bucket over 20 trials at {100 * worst:.0f} % of it")
E = n * s
mean = E / 256
cap = mean + 12 * np.sqrt(mean) + 1024
worst, sig = 0.0, 0.0
for trial in range(20):
    # distinct hashes with their multiplicities: ancestral k-mers (kept by m ~ Binomial(10, 0.584) members, m >= 1) supply
    # 58.4 % of the elements, private (mutated) k-mers the rest
    n_anc = int(0.584 * E / 5.84)
    m = rng.binomial(members, 0.584, size=int(n_anc * 1.02))
    m = m[m > 0][:n_anc]
    mult = np.concatenate([m, np.ones(max(E - int(m.sum()), 0), dtype=np.int64)])
    load = np.bincount(rng.integers(0, 256, size=len(mult)), weights=mult, minlength=256)
    worst = max(worst, load.max() / cap)
    sig = max(sig, load.std())
print(f"elements, {n} genomes: mean bucket {mean:.0f} (sigma {sig:.0f}, Poisson {np.sqrt(mean):.0f}), capacity {cap:.0f} = mean + {(cap - mean) / sig:.1f} sigma, "
      f"fullest bucket over 20 trials at {100 * worst:.1f} % of it")
