#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference's own test data (run in the authoring container,
where /root/reference exists; the GPU box only sees the committed outputs).

Fixtures are DATA only:
  * FASTA inputs the reference's tests use (tests/data/set1, set2, abisko4 subset, antonio_mags),
    gzip-compressed copies -- inputs, not reference source code;
  * golden.json: for each genome pair of the reference's finch-path tests, the integers
    (i, j, common, total) and the f32 ANI.  Row "set1" is REFERENCE-EMITTED
    (src/finch.rs:111-119 asserts Some(0.9808188)); every other row is ORACLE-DERIVED
    (oracle/galah_oracle.c), consistent with the membership asserts of src/clusterer.rs:538-690.
  * sketches.npz: the oracle's MinHash sketches (k=21, s=1000, seed 0) of every fixture genome.
"""
import gzip
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/tests/data"
GENOMES = {
    "set1_1mbp": "set1/1mbp.fna",
    "set1_500kb": "set1/500kb.fna",
    "set2_1mbp": "set2/1mbp.fna",
    "set2_half": "set2/1mbp.half_aligned.fna",
    "abisko_S1X13": "abisko4/73.20120800_S1X.13.fna",
    "abisko_S2D19": "abisko4/73.20120600_S2D.19.fna",
    "abisko_S3X12": "abisko4/73.20120700_S3X.12.fna",
    "abisko_S2D13": "abisko4/73.20110800_S2D.13.fna",
    "antonio_MAG52": "antonio_mags/BE_RX_R2_MAG52.fna",
    "antonio_MAG189": "antonio_mags/BE_RX_R3_MAG189.fna",
    "clash_500kb": "set1_name_clash/500kb.fna",
    "abisko_S2D10": "abisko4/73.20110600_S2D.10.fna",   # src/genome_stats.rs:61-73 golden
    "abisko_S1D21": "abisko4/73.20120800_S1D.21.fna",   # tests/test_cmdline.rs:12-61,304-352 (quality order)
    "abisko_S2M16": "abisko4/73.20110800_S2M.16.fna",
}
# multi-record files whose RECORDS the reference clusters (--cluster-contigs); copied as data, not sketched here
CONTIG_FILES = {"contigs_specific": "contigs/contigs_specific.fna",   # tests/test_cmdline.rs:482-505
                "contigs": "contigs/contigs.fna",                     # :461-480 (--large-contigs), :546-567 with contigs_extra
                "contigs_extra": "contigs/contigs_extra.fna",         # :546-567 (--small-contigs)
                "contigs_rep_bug": "contigs/contigs_rep_bug.fna"}     # :570-588 (--large-contigs), :591-609 (--small-contigs)
PAIRS = [("set1_1mbp", "set1_500kb"), ("abisko_S1X13", "abisko_S2D19"), ("abisko_S1X13", "abisko_S3X12"),
         ("abisko_S1X13", "abisko_S2D13"), ("abisko_S2D19", "abisko_S3X12"), ("abisko_S2D19", "abisko_S2D13"),
         ("abisko_S3X12", "abisko_S2D13"), ("antonio_MAG52", "antonio_MAG189"), ("set2_1mbp", "set2_half"),
         ("set1_500kb", "clash_500kb"), ("abisko_S1X13", "antonio_MAG52"), ("abisko_S1D21", "abisko_S2M16")]


def main():
    fasta_dir = os.path.join(HERE, "fasta")
    os.makedirs(fasta_dir, exist_ok=True)
    sketches = {}
    for name, rel in GENOMES.items():
        src = os.path.join(REF, rel)
        dst = os.path.join(fasta_dir, name + ".fna.gz")
        with open(src, "rb") as f, gzip.GzipFile(dst, "wb", compresslevel=9, mtime=0) as g:
            shutil.copyfileobj(f, g)
        sk_plain = oracle.sketch_file(src)
        sk_gz = oracle.sketch_file(dst)
        assert np.array_equal(sk_plain, sk_gz), name
        sketches[name] = sk_plain
    np.savez_compressed(os.path.join(HERE, "sketches.npz"), **sketches)
    for name, rel in CONTIG_FILES.items():
        with open(os.path.join(REF, rel), "rb") as f, gzip.GzipFile(os.path.join(fasta_dir, name + ".fna.gz"), "wb", compresslevel=9, mtime=0) as g:
            shutil.copyfileobj(f, g)
    rows = []
    for a, b in PAIRS:
        common, total = oracle.raw_distance(sketches[a], sketches[b])
        c2, t2 = oracle.raw_distance(sketches[a], sketches[b], closed_form=True)
        assert (common, total) == (c2, t2)
        ani = np.float32(oracle.mash_ani(common, total, 21)) if total else None
        rows.append({"a": a, "b": b, "len_a": int(len(sketches[a])), "len_b": int(len(sketches[b])),
                     "common": common, "total": total,
                     "ani_f32": None if ani is None else float(ani),
                     "ani_f32_bits": None if ani is None else int(np.float32(ani).view(np.uint32)),
                     "source": "reference src/finch.rs:111-119" if (a, b) == PAIRS[0] else "oracle-derived"})
    assert rows[0]["ani_f32_bits"] == int(np.float32(0.9808188).view(np.uint32)), rows[0]
    # genome statistics: both rows are REFERENCE-EMITTED (src/genome_stats.rs:61-86)
    stats = {"abisko_S2D10": [161, 6506, 8289], "set1_1mbp": [1, 0, 1000000]}
    for name, want in stats.items():
        assert list(oracle.genome_stats(os.path.join(fasta_dir, name + ".fna.gz"))) == want, name
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump({"k": 21, "s": 1000, "seed": 0, "pairs": rows, "genome_stats": stats}, f, indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
