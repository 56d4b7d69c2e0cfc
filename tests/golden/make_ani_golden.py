#!/usr/bin/env python3
"""Freezes the build-defined ANI estimator (oracle/galah_oracle_ani.c; skani's own floats are unpinned):
writes tests/golden/ani_golden.json = for every pair of the fixture genomes under tests/golden/fasta (copies of the
reference's tests/data FASTA files, see make_golden.py), every pair of the records of contigs_specific.fna
(tests/test_cmdline.rs:482-505) and (round 4, "anchor_contig_pairs") every pair of the records of contigs.fna + contigs_extra.fna
+ contigs_rep_bug.fna (:461-480, :546-567, :570-609), what the oracle computes at min_aligned_fraction 0.15 (galah's default, src/lib.rs:78):

    M, T            matched / total seeds of the lower-median chunk
    chunks          aligned chunks of both directions
    bases_q/_r      aligned bases of either genome (AF = bases / length)
    c_pair          seed density the pair was evaluated at (per-genome tiers, go_ani_density)
    ani_bits        the f32 the clusterer sees (percent; 0 below the aligned-fraction gate)

Every value is ORACLE-DERIVED -- the reference asserts no skani float.  The file exists so that a change of the
estimator's definition shows up as a reviewed diff of this file instead of "oracle and tests follow": the CPU suite
checks oracle == file, the GPU suite checks device == file.  Run it only to change the definition on purpose."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import fasta, fasta_records  # noqa: E402

GENOMES = ["set1_1mbp", "set1_500kb", "set2_1mbp", "set2_half", "abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13",
           "antonio_MAG52", "antonio_MAG189", "clash_500kb", "abisko_S2D10", "abisko_S1D21", "abisko_S2M16"]
MIN_AF = 0.15


def rows_of(names, sketches):
    rows = []
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            ani, afq, afr, d = oracle.ani_pair_detail(sketches[i], sketches[j], MIN_AF)
            rows.append({"q": names[i], "r": names[j], "M": d[0], "T": d[1], "chunks": d[2], "bases_q": d[3], "bases_r": d[4],
                         "c_pair": d[5], "ani": float(np.float32(ani)), "ani_bits": int(np.float32(ani).view(np.uint32))})
    return rows


def build():
    sk = [oracle.AniSketch.from_file(fasta(g)) for g in GENOMES]
    genomes = {g: {"length": int(s.length), "density": int(s.density), "seeds": int(s.nseeds)} for g, s in zip(GENOMES, sk)}
    cn, cs = fasta_records("contigs_specific")
    # a record's stream is its bases followed by one 'N' (what the ingest makes of a one-record file)
    csk = [oracle.AniSketch.from_bytes(np.concatenate([s, np.frombuffer(b"N", dtype=np.uint8)])) for s in cs]
    contigs = {n: {"length": int(s.length), "density": int(s.density), "seeds": int(s.nseeds)} for n, s in zip(cn, csk)}
    # round 4: the records of the reference's other three contig fixtures (tests/test_cmdline.rs:461-480, 546-567, 570-609)
    an, as_ = [], []
    for f in ("contigs", "contigs_extra", "contigs_rep_bug"):
        n, s = fasta_records(f, full_names=True)
        an += n
        as_ += s
    ask = [oracle.AniSketch.from_bytes(np.concatenate([s, np.frombuffer(b"N", dtype=np.uint8)])) for s in as_]
    anchors = {n: {"length": int(s.length), "density": int(s.density), "seeds": int(s.nseeds)} for n, s in zip(an, ask)}
    return {"definition_version": oracle.ani_definition_version(), "k": 15, "c": 125, "chunk": 20000, "min_aligned_fraction": MIN_AF, "source": "oracle-derived (no reference float exists)",
            "genomes": genomes, "genome_pairs": rows_of(GENOMES, sk), "contigs": contigs, "contig_pairs": rows_of(cn, csk),
            "anchor_contigs": anchors, "anchor_contig_pairs": rows_of(an, ask)}


if __name__ == "__main__":
    out = build()
    with open(os.path.join(HERE, "ani_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    hit = [r for r in out["genome_pairs"] if r["ani"] > 0]
    print(f"{len(out['genome_pairs'])} genome pairs ({len(hit)} with ANI > 0), {len(out['contig_pairs'])} contig pairs")
    for r in hit:
        print(r["q"], r["r"], r["M"], r["T"], r["chunks"], r["c_pair"], r["ani"])
