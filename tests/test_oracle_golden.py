"""Pins the CPU oracle to the reference's golden vectors.  CPU only."""
import numpy as np

import oracle
from conftest import fasta


def test_reference_golden_finch_rs_111_119():
    # src/finch.rs:111-119: distances(["set1/1mbp.fna","set1/500kb.fna"], 0.9, 1000, 21)
    #   == {(0,1): Some(0.9808188)}
    p = oracle.distances([fasta("set1_1mbp"), fasta("set1_500kb")], 0.9, 1000, 21)
    assert len(p) == 1
    assert (p["i"][0], p["j"][0]) == (0, 1)
    assert p["ani"][0].tobytes() == np.float32(0.9808188).tobytes()
    assert (p["common"][0], p["total"][0]) == (502, 1000)


def test_reference_golden_empty_at_099():
    # src/finch.rs:121-128
    assert len(oracle.distances([fasta("set1_1mbp"), fasta("set1_500kb")], 0.99, 1000, 21)) == 0


def test_golden_table(golden, golden_sketches):
    for row in golden["pairs"]:
        a, b = golden_sketches[row["a"]], golden_sketches[row["b"]]
        for closed in (False, True):
            assert oracle.raw_distance(a, b, closed_form=closed) == (row["common"], row["total"])
        ani = np.float32(oracle.mash_ani(row["common"], row["total"], 21))
        assert int(ani.view(np.uint32)) == row["ani_f32_bits"]


def test_oracle_resketch_matches_committed_sketches(golden_sketches):
    for name in ("set1_500kb", "abisko_S3X12"):  # single-contig and multi-contig-with-N inputs
        assert np.array_equal(oracle.sketch_file(fasta(name)), golden_sketches[name])


def test_sketch_extremes_from_survey(golden_sketches):
    assert int(golden_sketches["set1_1mbp"][0]) == 3491212462166
    assert int(golden_sketches["set1_1mbp"][-1]) == 18302639908299747
    assert int(golden_sketches["set1_500kb"][-1]) == 38405002408362914


def test_genome_stats_reference_goldens(golden):
    # src/genome_stats.rs:61-86: both rows are asserted by the reference's own unit tests
    assert oracle.genome_stats(fasta("abisko_S2D10")) == (161, 6506, 8289)
    assert oracle.genome_stats(fasta("set1_1mbp")) == (1, 0, 1_000_000)
    for name, want in golden["genome_stats"].items():
        assert list(oracle.genome_stats(fasta(name))) == want


def test_reference_membership_tests_finch_plus_skani():
    """src/clusterer.rs:631-690 (test_minhash_skani_hello_world / _two_clusters_same_ani): the four abisko
    genomes, finch precluster at 0.9, skani clusterer with min_aligned_threshold 0.2 ->
    [[0,1,2,3]] at 95 and [[0,1,3],[2]] at 99.  The reference needs the skani binary for these; the
    build-defined ANI estimator (median per-chunk containment) reproduces both memberships."""
    names = ["abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13"]  # order of the reference test
    paths = [fasta(n) for n in names]
    pairs = oracle.distances(paths, np.float32(0.9))
    assert len(pairs) == 6  # every pair clears the 0.9 precluster threshold
    sks = [oracle.AniSketch.from_file(p) for p in paths]
    ani = lambda a, b: oracle.ani_pair(sks[a], sks[b], 0.2)[0]
    cache = oracle.Cache.from_pairs(pairs)
    assert sorted(oracle.cluster(4, cache, 95.0, ani)) == [[0, 1, 2, 3]]
    assert sorted(oracle.cluster(4, cache, 99.0, ani)) == [[0, 1, 3], [2]]


# CheckM rows the reference's CLI tests use (tests/data/abisko4/abisko4.csv via tests/test_cmdline.rs:12-61;
# the values are quoted in the comment at test_cmdline.rs:14-16): completeness %, contamination %.
CHECKM = {"abisko_S1D21": (95.21, 0.00), "abisko_S2M16": (95.92, 0.65)}


def test_reference_cli_quality_order_and_skani_cluster():
    """tests/test_cmdline.rs:36-61 (Parks2020_reduced picks S2M.16 over S1D.21) and :304-352 (finch precluster +
    skani clusterer put both in one cluster whose representative is S2M.16).  test_github53 (:612-631) is the same pair as
    gzip files with a CheckM2 report whose completeness / contamination for the two genomes are these very numbers
    (tests/data/abisko4/abisko4_quality_report.tsv rows 8 and 17): the fixtures here ARE gzip files, so it is this test."""
    import galah_amd.quality as q
    names = ["abisko_S1D21", "abisko_S2M16"]          # order of the CLI's --genome-fasta-files
    st = [oracle.genome_stats(fasta(n)) for n in names]
    comp = np.array([CHECKM[n][0] for n in names], np.float32) / np.float32(100)
    cont = np.array([CHECKM[n][1] for n in names], np.float32) / np.float32(100)
    order = q.quality_order_parks2020_reduced(comp, cont, [s[0] for s in st], [s[1] for s in st])
    assert list(order) == [1, 0]
    # completeness-4contamination (test_cmdline.rs:12-34) prefers S1D.21 instead
    assert np.argmax(comp - np.float32(4) * cont) == 0
    paths = [fasta(names[i]) for i in order]
    pairs = oracle.distances(paths, np.float32(0.9))
    assert len(pairs) == 1
    sks = [oracle.AniSketch.from_file(p) for p in paths]
    ani = lambda a, b: oracle.ani_pair(sks[a], sks[b], 0.15)[0]   # CLI default --min-aligned-fraction 15
    assert oracle.cluster(2, oracle.Cache.from_pairs(pairs), 95.0, ani) == [[0, 1]]   # rep = S2M.16


def test_reference_cli_min_aligned_fraction():
    """tests/test_cmdline.rs:262-302: 1mbp.fna vs 1mbp.half_aligned.fna, finch precluster, ANI 95:
    --min-aligned-fraction 0.2 -> one representative (1mbp.fna); 0.6 -> two."""
    paths = [fasta("set2_1mbp"), fasta("set2_half")]
    pairs = oracle.distances(paths, np.float32(0.9))
    assert len(pairs) == 1
    sks = [oracle.AniSketch.from_file(p) for p in paths]
    for min_af, want in ((0.2, [[0, 1]]), (0.6, [[0], [1]])):
        ani = lambda a, b: oracle.ani_pair(sks[a], sks[b], min_af)[0]
        assert sorted(oracle.cluster(2, oracle.Cache.from_pairs(pairs), 95.0, ani)) == want


# tests/test_cmdline.rs:482-505 (test_contig_cluster_specific): the reference's expected clusters of the nine records of
# contigs_specific.fna at 95 % -- the ~1 kb contig, its four 100 %-identity variants and the 96 % variant together; the
# 94 % variant and the two unrelated contigs alone.
CONTIG_CLUSTERS = [[0, 1, 2, 3, 4, 5], [6], [7], [8]]


def test_reference_contig_expectation_with_automatic_density():
    """On 1 kb contigs a FracMinHash ANI at skani's --small-genomes density (c = 30) keeps ~35 seeds: the estimate's
    standard error (~1.5 ANI points) cannot be relied on to separate 96 % from 94 % at the 95 % threshold.  The seed density
    is chosen per genome (go_ani_density): a contig this short is seeded with every 15-mer, whatever base density the
    caller names, and the build-defined estimator reproduces the reference's expected contig clusters."""
    from conftest import fasta_records
    names, seqs = fasta_records("contigs_specific")
    assert names[5].startswith("96ANI") and names[6].startswith("94ANI") and len(names) == 9
    sk = [oracle.sketch_bytes(s, 21, 1000, 0) for s in seqs]
    n = len(seqs)
    cache = oracle.Cache()
    for i in range(n):
        for j in range(i + 1, n):
            c, t = oracle.raw_distance(sk[i], sk[j])
            ani = oracle.mash_ani(c, t, 21)
            if ani >= float(np.float32(0.9)):
                cache.insert((i, j), np.float32(ani))
    for base_c in (125, 30, 1):
        dense = [oracle.AniSketch.from_bytes(s, 15, base_c, 20000) for s in seqs]
        assert all(d.density == 1 and d.nseeds >= 0.95 * len(s) for d, s in zip(dense, seqs))
        got = oracle.cluster(n, cache, 95.0, lambda a, b: oracle.ani_pair(dense[a], dense[b], 0.15)[0])
        assert got == CONTIG_CLUSTERS
        assert oracle.ani_pair(dense[0], dense[5], 0.15)[0] >= 95.0 > oracle.ani_pair(dense[0], dense[6], 0.15)[0]


def test_ani_density_tiers():
    """c_g = c; while (c_g > 1 and L < 8192 c_g) c_g = max(1, c_g // 4): 125 -> 31 -> 7 -> 1."""
    assert [oracle.ani_density(L) for L in (0, 1000, 57343, 57344, 253951, 253952, 1023999, 1024000, 5_000_000)] == [1, 1, 1, 7, 7, 31, 31, 125, 125]
    assert [oracle.ani_density(L, 30) for L in (1000, 57343, 57344, 245759, 245760, 5_000_000)] == [1, 1, 7, 7, 30, 30]
    assert oracle.ani_density(10, 1) == 1 and oracle.ani_density(10**9, 1) == 1


def test_ani_mixed_density_pair_is_evaluated_at_the_sparser_density():
    """A 300 kb replicon (density 31) against a 2 Mb genome that contains a 97 % copy of it (density 125): the pair uses the
    seeds both samples hold -- those below the sparser threshold -- and is symmetric."""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    small = rng.choice(acgt, size=300_000)
    big = rng.choice(acgt, size=2_000_000)
    cp = small.copy()
    hit = rng.random(len(cp)) < 0.03
    cp[hit] = acgt[(np.searchsorted(acgt, cp[hit]) + rng.integers(1, 4, size=int(hit.sum()))) % 4]
    big[500_000:800_000] = cp
    a, b = oracle.AniSketch.from_bytes(small), oracle.AniSketch.from_bytes(big)
    assert (a.density, b.density) == (31, 125)
    ani, afq, afr, d = oracle.ani_pair_detail(a, b, 0.15)
    ani2, afq2, afr2, d2 = oracle.ani_pair_detail(b, a, 0.15)
    assert d[5] == 125 and ani == ani2 and (afq, afr) == (afr2, afq2) and d[:3] == d2[:3]
    true = 100.0 * float(np.mean(small == cp))
    assert afq > 0.9 and afr < 0.2 and abs(ani - true) < 0.5, (ani, true, afq, afr)
    # the same replicon against itself stays at its own density
    assert oracle.ani_pair_detail(a, oracle.AniSketch.from_bytes(cp), 0.15)[3][5] == 31


def test_ani_golden_file_is_what_the_oracle_computes():
    """tests/golden/ani_golden.json freezes the build-defined estimator on the reference's fixture genomes: a change of
    its definition has to change that file (tests/golden/make_ani_golden.py), in a diff a reviewer sees."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_ani_golden
    with open(os.path.join(here, "golden", "ani_golden.json")) as f:
        want = json.load(f)
    got = make_ani_golden.build()
    assert got["genomes"] == want["genomes"] and got["contigs"] == want["contigs"]
    assert got["genome_pairs"] == want["genome_pairs"]
    assert got["contig_pairs"] == want["contig_pairs"]
    assert len(want["genome_pairs"]) == 91 and len(want["contig_pairs"]) == 36
    # round 4: the records of contigs.fna + contigs_extra.fna + contigs_rep_bug.fna (added keys; nothing above changed)
    assert got["anchor_contigs"] == want["anchor_contigs"] and got["anchor_contig_pairs"] == want["anchor_contig_pairs"]
    assert len(want["anchor_contig_pairs"]) == 28


# The reference's remaining contig membership expectations (all of them held by skani: `skani triangle` at -c 125 for
# --large-contigs, -c 30 for --small-contigs; src/skani.rs:414-484), in record order with every record a genome.
#   tests/test_cmdline.rs:461-480  contigs.fna, --large-contigs            {13024, 13024_2} {50844} {37820}
#   :546-567  contigs.fna + contigs_extra.fna, --small-contigs             {13024, _2, _3} {50844} {37820}
#   :570-588  contigs_rep_bug.fna, --large-contigs                         {k141_313035, k141_401621, NODE_1070}
#   :591-609  contigs_rep_bug.fna, --small-contigs                         {k141_313035, k141_401621} {NODE_1070}
CONTIG_ANCHORS = [(["contigs"], 125, [[0, 1], [2], [3]]), (["contigs", "contigs_extra"], 30, [[0, 1, 4], [2], [3]])]
REP_BUG_EXPECTED = {125: [[0, 1, 2]], 30: [[0, 1], [2]]}
# what the build-defined estimator gives for contigs_rep_bug.fna at EITHER base density (all three records, 28-42 kb, are
# seeded with every 15-mer by the per-genome rule): the --small-contigs answer (:591-609), since round 5 (few-chunk pooling)
REP_BUG_GOT = [[0, 1], [2]]


def _contig_flow(files, base_c):
    from conftest import fasta_records
    names, seqs = [], []
    for f in files:
        n, s = fasta_records(f, full_names=True)
        names += n
        seqs += s
    sk = [oracle.sketch_bytes(s, 21, 1000, 0) for s in seqs]
    n = len(seqs)
    cache = oracle.Cache()
    for i in range(n):
        for j in range(i + 1, n):
            c, t = oracle.raw_distance(sk[i], sk[j])
            ani = oracle.mash_ani(c, t, 21)
            if ani >= float(np.float32(0.9)):
                cache.insert((i, j), np.float32(ani))
    stop = np.frombuffer(b"N", dtype=np.uint8)
    dense = [oracle.AniSketch.from_bytes(np.concatenate([s, stop]), 15, base_c, 20000) for s in seqs]
    return names, dense, oracle.cluster(n, cache, 95.0, lambda a, b: oracle.ani_pair(dense[a], dense[b], 0.15)[0])


def test_reference_contig_anchors_contigs_and_contigs_extra():
    for files, base_c, want in CONTIG_ANCHORS:
        for c in (base_c, 125, 30):   # the per-genome density makes the caller's base density immaterial at these lengths
            names, dense, got = _contig_flow(files, c)
            assert all(d.density == 1 for d in dense)
            assert sorted(sorted(x) for x in got) == want, (files, c, got)
            assert got[0][0] == 0 and names[0] == "73.20110600_S2D.10_contig_13024"   # the representative the reference prints first


def test_reference_contig_anchor_rep_bug_gives_the_small_contigs_answer():
    """contigs_rep_bug.fna has TWO expected answers in the reference (skani -c 125, :570-588: one cluster of three; -c 30,
    :591-609: NODE_1070 apart) -- skani's own estimate of ANI(k141_313035, NODE_1070) straddles 95 % with its seed density.
    All three records (28-42 kb) are seeded with every 15-mer by the per-genome rule, so the build-defined estimator has ONE
    answer at either base density, and since round 5 it is the --small-contigs one: with fewer than nine aligned chunks the
    pooled counts replace the lower median (of two chunks that was the minimum: chunk 0 of the 28 kb contig holds 8 kb
    without a homologue in NODE_1070 and pulled the pair to 92.35, and k141_401621 to NODE_1070's side -- NEITHER answer;
    profiles/r05_ani_few_chunks.txt has the measurement the change rests on).  Pooled: 96.37 / 94.18 / 96.27 --
    k141_401621 stays with k141_313035, NODE_1070 (94.18 < 95) is its own representative.  The --large-contigs answer
    needs the pair above 95, which skani reaches only through the noise of its sparser seeding: documented as not
    reproducible (anchor table)."""
    for c in (125, 30):
        names, dense, got = _contig_flow(["contigs_rep_bug"], c)
        assert names[0].startswith("k141_313035 flag=1") and names[2].startswith("NODE_1070")
        assert got == REP_BUG_GOT and got == REP_BUG_EXPECTED[30]   # representative first, members in the reference's order
        assert sorted(sorted(x) for x in got) != REP_BUG_EXPECTED[125]
        a = {(i, j): oracle.ani_pair(dense[i], dense[j], 0.15)[0] for i in range(3) for j in range(i + 1, 3)}
        assert [round(a[k], 2) for k in ((0, 1), (0, 2), (1, 2))] == [96.37, 94.18, 96.27]
        # the rule it replaced, still reachable for measurement: the lower median of the two chunks
        m = {(i, j): oracle.ani_pair_pool_below(dense[i], dense[j], 0, 0.15)[0] for i in range(3) for j in range(i + 1, 3)}
        assert [round(m[k], 2) for k in ((0, 1), (0, 2), (1, 2))] == [95.94, 92.35, 96.21]


def _finch_plus_ani(names, thr, min_af):
    paths = [fasta(n) for n in names]
    sks = [oracle.AniSketch.from_file(p) for p in paths]
    pairs = oracle.distances(paths, np.float32(0.9))
    return oracle.cluster(len(names), oracle.Cache.from_pairs(pairs), thr, lambda a, b: oracle.ani_pair(sks[a], sks[b], min_af)[0])


def test_reference_cli_representative_list_and_github7():
    """tests/test_cmdline.rs:161-181: [name_clash/500kb, set1/500kb, set1/1mbp] -> representatives set1/500kb, then
    name_clash/500kb (the bigger precluster first; 1mbp joins 500kb).  :417-440 (github issue 7): the two antonio MAGs at
    --min-aligned-fraction 60 -> one cluster, representative MAG52."""
    assert _finch_plus_ani(["clash_500kb", "set1_500kb", "set1_1mbp"], 95.0, 0.15) == [[1, 2], [0]]
    # :62-119 (symlink-directory tests): 500kb.fna is the only representative of [set1/500kb, set1/1mbp];
    # :120-160 and :184-216 repeat the three-genome outcome above (both 500kb files representatives, 1mbp.fna not)
    assert _finch_plus_ani(["set1_500kb", "set1_1mbp"], 95.0, 0.15) == [[0, 1]]
    # :1100-1125 (needs CheckM2 for the order, so only the membership is checked): the four genomes fall into
    # {S2M.16, S1D.21} and {500kb, 1mbp}
    got = _finch_plus_ani(["set1_1mbp", "set1_500kb", "abisko_S1D21", "abisko_S2M16"], 95.0, 0.15)
    assert sorted(sorted(c) for c in got) == [[0, 1], [2, 3]]
    assert _finch_plus_ani(["antonio_MAG52", "antonio_MAG189"], 95.0, 0.6) == [[0, 1]]


# sha256 of tests/golden/ani_golden.json's content (everything but "definition_version", as canonical JSON), by version of the
# ANI estimator's definition.  ADD an entry when the definition changes on purpose; never edit an old one.
ANI_GOLDEN_DIGESTS = {
    5: "707b9da04eba047a9f1a65f96be703b71d23535e7642d7fefe2397fd938d7e43",   # round 5: pooled counts below 9 aligned chunks
}


def test_ani_definition_is_versioned():
    """the oracle IS the definition of the ANI estimator (no skani float exists to pin it), so the definition
    must stop moving silently.  Oracle, device library and golden file name one version; the golden file's digest is recorded per
    version -- a regenerated file with other numbers fails here until GO_ANI_DEFINITION_VERSION (oracle/galah_oracle.h) and
    GHIP_ANI_DEFINITION_VERSION (include/galah_hip.h) are bumped and the new version's digest is added above."""
    import hashlib
    import json
    import os
    import re
    from conftest import GOLDEN, ROOT
    from galah_amd import _lib
    with open(os.path.join(GOLDEN, "ani_golden.json")) as f:
        d = json.load(f)
    v = oracle.ani_definition_version()
    assert v == d["definition_version"] == int(_lib.lib().ghip_ani_definition_version()), "oracle, golden file and device library name different versions"
    assert v == max(ANI_GOLDEN_DIGESTS), "a new version needs its digest recorded in ANI_GOLDEN_DIGESTS"
    rows = {k: x for k, x in d.items() if k != "definition_version"}
    digest = hashlib.sha256(json.dumps(rows, sort_keys=True, separators=(",", ":")).encode()).hexdigest()
    assert digest == ANI_GOLDEN_DIGESTS[v], ("tests/golden/ani_golden.json changed but the ANI definition's version did not: bump GO_ANI_DEFINITION_VERSION and "
                                             "GHIP_ANI_DEFINITION_VERSION, regenerate with tests/golden/make_ani_golden.py, add the new digest")
    # the constants of the definition that oracle and device each spell out are the same numbers
    osrc = open(os.path.join(ROOT, "oracle", "galah_oracle_ani.c")).read()
    dsrc = open(os.path.join(ROOT, "galah_amd", "csrc", "ghip_internal.h")).read()
    for name in ("POOL_BELOW", "SEEDS_WANTED"):
        o = re.search(r"#define GO_ANI_%s (\d+)" % name, osrc)
        g = re.search(r"#define GHIP_ANI_%s (\d+)u" % name, dsrc)
        assert o and g and o.group(1) == g.group(1), name
