"""Round-4 measurement: the frozen band-vote ANI estimator against an ORDERED COLINEAR CHAIN (gap-cost
DP over a chunk's anchors, Shaw & Yu 2023) and against span-limited denominators -- oracle only, CPU
(oracle/galah_oracle_ani.c: go_ani_pair_mode; nothing in the product uses these modes).

  band        the frozen definition (tests/golden/ani_golden.json)
  chain       matches = seeds of the kept chains, denominators = whole chunks          -> isolates chain vs band vote
  chain/span  denominators = the seeds inside the chains' query spans (what skani's per-chain estimate does)
  c/s pooled  the same with sum M / sum T instead of the lower median
  band/span   the band vote's matches, denominators between the first and last voting seed of a chunk
  sub D=n     the band vote's matches, denominators = the chunk's 2 kb sub-blocks that hold a matched seed, runs of <= n
              unmatched sub-blocks between matched ones (or the chunk's edge) filled in: a covered chunk keeps its whole
              denominator, an uncovered stretch longer than n sub-blocks leaves it (a form the device could afford)

Sections: (1) the reference's fixture genomes (every pair with ANI > 0), (2) the reference's contig fixtures and what its
tests expect of them (tests/test_cmdline.rs:461-609), (3) synthetic pairs against counted identity, plain and structured
(the generators of scripts/ani_scenarios.py).  usage: ani_chain_vs_band.py [length=2000000] [pairs=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import fasta, fasta_records  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 6
MODES = [("chain", oracle.CHAIN), ("chain/span", oracle.CHAIN_SPAN), ("c/s pooled", oracle.CHAIN_SPAN | oracle.AGG_POOLED),
         ("band/span", oracle.BAND_SPAN), ("sub D=1", oracle.BAND_SUB | 1 << 8), ("sub D=2", oracle.BAND_SUB | 2 << 8)]
N = np.frombuffer(b"N", dtype=np.uint8)


def all_modes(a, b, min_af=0.15):
    return [oracle.ani_pair_detail(a, b, min_af)[0]] + [oracle.ani_pair_mode(a, b, fl, min_af)[0] for _, fl in MODES]


def header(first):
    print(f"{first:<44}{'band':>8}" + "".join(f"{n:>12}" for n, _ in MODES))


print("== (1) fixture genomes: pairs with ANI > 0 under any mode; '*' = the >= 95 decision differs from the band vote's")
GENOMES = ["set1_1mbp", "set1_500kb", "set2_1mbp", "set2_half", "abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13",
           "antonio_MAG52", "antonio_MAG189", "clash_500kb", "abisko_S2D10", "abisko_S1D21", "abisko_S2M16"]
sk = [oracle.AniSketch.from_file(fasta(g)) for g in GENOMES]
header("pair")
flips = 0
for i in range(len(sk)):
    for j in range(i + 1, len(sk)):
        v = all_modes(sk[i], sk[j])
        if max(v) > 0:
            mark = "".join("*" if (x >= 95.0) != (v[0] >= 95.0) or (x >= 99.0) != (v[0] >= 99.0) else " " for x in v[1:])
            flips += mark.count("*")
            print(f"{GENOMES[i] + ' / ' + GENOMES[j]:<44}{v[0]:8.2f}" + "".join(f"{x:11.2f}{m}" for x, m in zip(v[1:], mark)))
print(f"decisions at 95 % or 99 % that differ from the band vote's: {flips}")

print("\n== (2) contig fixtures (every record a genome; density per genome: all of them are seeded with every 15-mer)")
for files, note in ((["contigs"], "expected (:461-480): {13024, 13024_2} {50844} {37820}"),
                    (["contigs", "contigs_extra"], "expected (:546-567): {13024, _2, _3} {50844} {37820}"),
                    (["contigs_rep_bug"], "expected --large-contigs (:570-588): one cluster of three; --small-contigs (:591-609): NODE_1070 apart, "
                                          "i.e. ANI(k141_313035, NODE_1070) >= 95 with skani -c 125 and < 95 with -c 30"),
                    (["contigs_specific"], "expected (:482-505): the 96 % variant with the contig, the 94 % variant apart")):
    names, seqs = [], []
    for f in files:
        n, s = fasta_records(f)
        names += n; seqs += s
    csk = [oracle.AniSketch.from_bytes(np.concatenate([s, N])) for s in seqs]
    print("--", "+".join(files), ":", note)
    header("pair")
    for i in range(len(csk)):
        for j in range(i + 1, len(csk)):
            v = all_modes(csk[i], csk[j])
            if max(v) > 0 and (files != ["contigs_specific"] or i == 0):
                print(f"{names[i][-20:] + ' / ' + names[j][-20:]:<44}{v[0]:8.2f}" + "".join(f"{x:12.2f}" for x in v[1:]))

print(f"\n== (3) synthetic pairs against counted identity ({S} pairs of {L} bp per row): mean error / largest |error|, ANI points")
sys.argv = [sys.argv[0], str(L), str(S)]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ani_scenarios_gen", os.path.join(ROOT, "scripts", "ani_scenarios.py"))
src = open(os.path.join(ROOT, "scripts", "ani_scenarios.py")).read().split("\nprint(f\"{'scenario'")[0]   # the generators only
gen = {"__file__": os.path.join(ROOT, "scripts", "ani_scenarios.py")}
exec(compile(src, "ani_scenarios.py", "exec"), gen)
header("scenario (true %)")
for name, rate in (("plain", 0.0051), ("plain", 0.0253), ("plain", 0.0527), ("repeats", 0.0253), ("island", 0.0253), ("rearranged", 0.0253),
                   ("fragmented", 0.0253), ("fragmented", 0.0208), ("fragmented", 0.0300)):
    rng = np.random.default_rng(99)
    errs, tr, side = [], [], []
    for _ in range(S):
        a, b, true = gen["scenario"](rng, name, rate)
        v = all_modes(oracle.AniSketch.from_bytes(a), oracle.AniSketch.from_bytes(b))
        errs.append([x - true for x in v]); tr.append(true)
        side.append([(x >= 95.0) == (true >= 95.0) for x in v])
    e = np.array(errs)
    wrong = (~np.array(side)).sum(axis=0)
    print(f"{name + ' (%.2f)' % np.mean(tr):<44}" + "".join(f"{e[:, m].mean():+7.3f}/{np.abs(e[:, m]).max():.2f}{'!' * int(wrong[m] > 0):1}" + " " * (0 if m == 0 else 0) for m in range(e.shape[1])))
# partially overlapping genomes: an incomplete MAG (60 % of the genome, in 40 contigs) against the complete genome
print("-- a 60 %-complete MAG (40 contigs) of one genome against a 97.5 / 95.8 / 94.2 % relative, complete")
for rate in (0.0126, 0.0212, 0.0295):
    rng = np.random.default_rng(7)
    errs = []
    for _ in range(S):
        anc = rng.choice(gen["acgt"], size=L)
        a, b = gen["substitute"](rng, anc, rate), gen["substitute"](rng, anc, rate)
        cuts = np.sort(rng.choice(np.arange(1, L), size=66, replace=False))
        parts = [p for x, p in enumerate(np.split(np.arange(L), cuts)) if x % 5 < 3]
        order = rng.permutation(len(parts))
        mag = np.concatenate([np.concatenate([(gen["revcomp"](a[parts[i]]) if rng.random() < 0.5 else a[parts[i]]), N]) for i in order])
        true = 100.0 * float(np.mean(np.concatenate([a[p] == b[p] for p in parts])))
        v = all_modes(oracle.AniSketch.from_bytes(mag), oracle.AniSketch.from_bytes(b))
        errs.append([x - true for x in v])
    e = np.array(errs)
    print(f"{'MAG vs complete (%.2f)' % (100 * (1 - rate) ** 2 + 100 * rate * rate / 3):<44}" + "".join(f"{e[:, m].mean():+7.3f}/{np.abs(e[:, m]).max():.2f} " for m in range(e.shape[1])))
