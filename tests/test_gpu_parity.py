"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures."""
import os

import numpy as np
import pytest

import galah_amd
import oracle
from conftest import DEV, fasta, random_sketches, never_run_on_hardware

pytestmark = pytest.mark.gpu

ALL = ["set1_1mbp", "set1_500kb", "set2_1mbp", "set2_half", "abisko_S1X13", "abisko_S2D19", "abisko_S3X12",
       "abisko_S2D13", "antonio_MAG52", "antonio_MAG189", "clash_500kb", "abisko_S1D21", "abisko_S2M16"]


def test_reference_golden_through_hip(ctx):
    # src/finch.rs:111-128 run through FinchPreclusterer.distances on the GPU
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx)
    got = pre.distances([fasta("set1_1mbp"), fasta("set1_500kb")])
    want = galah_amd.SortedPairGenomeDistanceCache()
    want.insert((0, 1), 0.9808188)
    assert got == want
    assert repr(got) == "SortedPairGenomeDistanceCache { internal: {(0, 1): Some(0.9808188)} }"
    empty = galah_amd.FinchPreclusterer(0.99, 1000, 21, ctx=ctx).distances([fasta("set1_1mbp"), fasta("set1_500kb")])
    assert empty == galah_amd.SortedPairGenomeDistanceCache()


def test_sketches_bit_exact_on_fixture_genomes(ctx, golden_sketches):
    sk = ctx.sketch_files([fasta(n) for n in ALL], 21, 1000, 0, io_threads=4)
    hashes, lens = sk.to_host()
    for i, name in enumerate(ALL):
        want = golden_sketches[name]
        assert lens[i] == len(want)
        assert np.array_equal(hashes[i, : lens[i]], want), name
        assert np.all(hashes[i, lens[i]:] == np.uint64(0xFFFFFFFFFFFFFFFF))


def test_precluster_golden_table(ctx, golden):
    sk = ctx.sketch_files([fasta(n) for n in ALL], 21, 1000, 0, io_threads=4)
    pairs = ctx.precluster(sk, np.float32(0.0))  # threshold 0: every pair comes back
    assert len(pairs) == len(ALL) * (len(ALL) - 1) // 2
    look = {(int(p["i"]), int(p["j"])): p for p in pairs}
    for row in golden["pairs"]:
        i, j = ALL.index(row["a"]), ALL.index(row["b"])
        p = look[(min(i, j), max(i, j))]
        assert (int(p["common"]), int(p["total"])) == (row["common"], row["total"])
        assert int(p["ani"].view(np.uint32)) == row["ani_f32_bits"]
    hashes, lens = sk.to_host()
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.0))
    assert pairs.tobytes() == want.tobytes()


@pytest.mark.parametrize("n,s,min_len,thr", [(37, 1000, None, 0.9), (64, 1000, 1, 0.0), (130, 256, 1, 0.8),
                                               (9, 64, 0, 0.0), (2, 1000, None, 0.0), (100, 1024, 900, 0.95),
                                               (33, 2000, 1500, 0.9)])
def test_pairs_random_sketches_vs_oracle(ctx, n, s, min_len, thr):
    rng = np.random.default_rng(n * 1000 + s)
    hashes, lens = random_sketches(rng, n, s, shared_groups=5, min_len=min_len)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(thr))
    want = oracle.distances_from_sketches(hashes, lens, np.float32(thr))
    assert got.tobytes() == want.tobytes()
    assert ctx.last_pairs_compared == n * (n - 1) // 2


def test_pairs_edge_cases(ctx):
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    s = 8
    hashes = np.full((5, s), M, dtype=np.uint64)
    lens = np.array([0, 0, 3, 8, 8], dtype=np.uint32)  # two empty sketches, ragged, full
    hashes[2, :3] = [5, 9, 11]
    hashes[3] = [1, 5, 9, 11, 20, 30, 40, 50]
    hashes[4] = [5, 9, 11, 12, 13, 14, 15, 2**64 - 2]
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(0.0))
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.0))
    assert got.tobytes() == want.tobytes()
    # empty vs empty: total 0 -> NaN dropped by f64::max/min -> ANI 1.0 (documented reference quirk)
    assert float(got[0]["ani"]) == 1.0 and int(got[0]["total"]) == 0


def test_pair_shards_partition_the_triangle(ctx):
    rng = np.random.default_rng(3)
    hashes, lens = random_sketches(rng, 77, 1000, shared_groups=4)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    full = ctx.precluster(sk, np.float32(0.0))
    for world in (2, 3, 8):
        parts, compared = [], 0
        for r in range(world):
            parts.append(ctx.precluster(sk, np.float32(0.0), r, world))
            compared += ctx.last_pairs_compared
        assert compared == 77 * 76 // 2
        merged = np.sort(np.concatenate(parts), order=["i", "j"])
        assert merged.tobytes() == full.tobytes()


def _streams():
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = rng.choice(acgt, size=300_000)
    out = {"plain": base.copy()}
    withn = base.copy()
    withn[rng.integers(0, len(base), 300)] = ord("N")
    withn[5000:5100] = ord("N")
    withn[70000] = ord("-")
    out["with_N_runs"] = withn
    out["shorter_than_k"] = base[:20].copy()
    out["exactly_k"] = base[:21].copy()
    out["tiny_lt_s_kmers"] = base[:700].copy()
    out["empty"] = np.zeros(0, dtype=np.uint8)
    out["all_N"] = np.full(5000, ord("N"), dtype=np.uint8)
    out["repetitive"] = np.tile(base[:1500], 400)          # 600 kb, only ~1500 distinct k-mers
    out["homopolymer"] = np.full(100_000, ord("A"), dtype=np.uint8)
    out["chunk_boundary"] = base[:16384 + 21].copy()       # one k-mer straddles two blocks
    out["lowish"] = rng.choice(acgt, size=70_000)
    return out


def test_sketch_edge_cases_vs_oracle(ctx):
    streams = _streams()
    names = list(streams)
    for k, s in ((21, 1000), (21, 64), (15, 500), (32, 1000), (11, 256)):
        g = ctx.genomes_from_host([streams[n] for n in names])
        sk = ctx.sketch_genomes(g, k, s, 0)
        hashes, lens = sk.to_host()
        for i, n in enumerate(names):
            want = oracle.sketch_bytes(streams[n], k, s, 0)
            assert lens[i] == len(want), (n, k, s, lens[i], len(want))
            assert np.array_equal(hashes[i, : lens[i]], want), (n, k, s)


def test_sketch_nonzero_seed(ctx):
    rng = np.random.default_rng(2)
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=50_000)
    g = ctx.genomes_from_host([seq])
    for seed in (1, 42, 0xFFFFFFFF):
        hashes, lens = ctx.sketch_genomes(g, 21, 200, seed).to_host()
        assert np.array_equal(hashes[0, : lens[0]], oracle.sketch_bytes(seq, 21, 200, seed))


def test_synthetic_generator_matches_oracle(ctx):
    g = ctx.genomes_synthetic(42, 2, 3, 100_003, 0.0253)
    for i in range(6):
        assert np.array_equal(g.to_host(i), oracle.synth_genome(42, i // 3, i % 3, 100_003, 0.0253))


def test_ani_pairs_vs_oracle(ctx):
    seed, n_species, members, length = 9, 2, 4, 400_000
    g = ctx.genomes_synthetic(seed, n_species, members, length, 0.0253)
    n = n_species * members
    idx = ctx.ani_index_build(g)
    pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.uint32)
    ani, af = ctx.ani_pairs(idx, pairs, 0.15, want_af=True)
    osk = [oracle.AniSketch.from_bytes(g.to_host(i)) for i in range(n)]
    for x, (a, b) in enumerate(pairs):
        o, afq, afr = oracle.ani_pair(osk[a], osk[b], 0.15)
        assert np.float32(o) == ani[x], (a, b, o, ani[x])
        assert np.float32(afq) == af[x, 0] and np.float32(afr) == af[x, 1]
    same = [ani[x] for x, (a, b) in enumerate(pairs) if a // members == b // members]
    diff = [ani[x] for x, (a, b) in enumerate(pairs) if a // members != b // members]
    assert all(94.0 < v < 96.0 for v in same), same   # members are ~95 % identical by construction
    assert all(v == 0.0 for v in diff)                # unrelated genomes fail the aligned-fraction gate


def test_ani_pairs_both_workgroup_shapes(ctx, opts):
    """Short pair lists run 16 waves per pair, long ones 8 (ani.hip): the same values either way -- on genomes of very
    different sizes (rounds of 64 bins and of 1024), with sparse and dense seeds."""
    g = ctx.genomes_synthetic(4, 3, 4, 700_000, 0.03)
    small = ctx.genomes_synthetic(5, 4, 3, 9_000, 0.02)
    for genomes, c in ((g, 125), (g, 7), (small, 30)):
        n = len(genomes)
        idx = ctx.ani_index_build(genomes, 15, c, 20000)
        pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.uint32)
        opts(ani_tall_below=0)
        want, want_af = ctx.ani_pairs(idx, pairs, 0.15, want_af=True)
        assert (want > 90).sum() >= n
        opts(ani_tall_below=1000000)
        got, got_af = ctx.ani_pairs(idx, pairs, 0.15, want_af=True)
        assert np.array_equal(got, want) and np.array_equal(got_af, want_af), c
        opts(ani_tall_below=200)
        for m in (1, 3, 90, len(pairs)):
            assert np.array_equal(ctx.ani_pairs(idx, pairs[:m], 0.15), want[:m]), m
        idx.free()


def test_ani_on_fixture_files_vs_oracle(ctx):
    names = ["abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13", "antonio_MAG52"]
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=4)
    paths = [fasta(n) for n in names]
    cl.prepare(paths)
    osk = [oracle.AniSketch.from_file(p) for p in paths]
    for a in range(len(names)):
        for b in range(len(names)):
            if a == b:
                continue
            got = cl.calculate_ani(paths[a], paths[b])
            assert got == np.float32(oracle.ani_pair(osk[a], osk[b], 0.15)[0]), (names[a], names[b])


def test_calculate_ani_from_many_threads(ctx):
    """The reference calls calculate_ani from rayon workers (src/clusterer.rs:267-270,283-293,375-399; `C: Sync`,
    clusterer.rs:14).  Eight threads ask for every ordered pair at once, two of the genomes not indexed yet (so some
    calls re-index while others look up): every answer equals the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    names = ["abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13", "set1_1mbp", "set1_500kb"]
    paths = [fasta(n) for n in names]
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=4)
    cl.prepare(paths[:4])
    osk = [oracle.AniSketch.from_file(p) for p in paths]
    want = {(a, b): np.float32(oracle.ani_pair(osk[a], osk[b], 0.15)[0]) for a in range(6) for b in range(6) if a != b}
    jobs = [(a, b) for t in range(8) for (a, b) in sorted(want, key=lambda ab: (ab[0] * 7 + ab[1] * 5 + t * 3) % 11)]
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(lambda ab: cl.calculate_ani(paths[ab[0]], paths[ab[1]]), jobs))
    assert all(g == want[ab] for g, ab in zip(got, jobs))
    assert want[(0, 1)] > 95 and want[(4, 5)] > 95 and want[(0, 4)] == 0


def test_reused_ani_clusterer_is_checked_by_identity_not_by_count(ctx):
    """A HipAniClusterer prepared for another list of the same length, or grown by calculate_ani() (its index is then
    in first-seen order, not the caller's), must not answer edge indices of a new list from its stale index: cluster()
    re-prepares it.  A different context keeps the fused one-ingest path out of the way."""
    names = ["set1_500kb", "set1_1mbp", "antonio_MAG52", "antonio_MAG189"]
    paths = [fasta(n) for n in names]
    other = galah_amd.Context(0)
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=other, io_threads=2)
    want = galah_amd.cluster(paths, pre, galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=2))
    assert want == [[0, 1], [2, 3]]
    stale = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=2)
    stale.prepare(paths[::-1])                       # same genomes, reversed: same count, wrong positions
    assert not stale.prepared_for(paths) and stale.prepared_for(paths[::-1])
    assert galah_amd.cluster(paths, pre, stale) == want
    grown = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=2)
    grown.calculate_ani(paths[3], paths[2])          # index order: 3, 2
    grown.calculate_ani(paths[1], paths[0])          # ... then 1, 0
    assert len(grown._path_index) == 4 and not grown.prepared_for(paths)
    assert galah_amd.cluster(paths, pre, grown) == want
    # unrelated genomes (no precluster pair): singletons, through the batched clusterer
    lone = [fasta("abisko_S1X13"), fasta("antonio_MAG52"), fasta("set1_1mbp")]
    assert galah_amd.cluster(lone, pre, galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx)) == [[0], [1], [2]]
    other.close()


def test_cluster_end_to_end_vs_oracle(ctx):
    """Every fixture genome the reference's tests hold for this path (15 files; the multi-record contig file counts as
    one genome here), three ANI thresholds, two aligned-fraction gates: clusters, order and representatives equal the
    oracle's run of the reference's greedy algorithm."""
    names = ["abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13", "antonio_MAG52", "antonio_MAG189",
             "set1_1mbp", "set1_500kb", "abisko_S1D21", "abisko_S2M16", "abisko_S2D10", "clash_500kb",
             "contigs_specific", "set2_1mbp", "set2_half"]
    paths = [fasta(n) for n in names]
    osk = [oracle.AniSketch.from_file(p) for p in paths]
    want_pairs = oracle.distances(paths, np.float32(0.9))
    for thr, min_af in ((95.0, 0.15), (98.0, 0.15), (99.0, 0.15), (95.0, 0.6)):
        pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=4)
        cl = galah_amd.HipAniClusterer(thr, min_af, ctx=ctx, io_threads=4)
        got = galah_amd.cluster(paths, pre, cl)
        want = oracle.cluster(len(paths), oracle.Cache.from_pairs(want_pairs), thr,
                              lambda a, b: oracle.ani_pair(osk[a], osk[b], min_af)[0])
        assert got == want, (thr, min_af, got, want)
        assert sorted(x for c in got for x in c) == list(range(len(paths)))


@never_run_on_hardware
def test_join_fused_form_equals_exact_form_and_survives_an_outgrown_capacity(ctx, opts):
    """ghip_options.join_fused (round 4: 20 -> 12 launches, no host round trip inside a partition): byte-identical pair lists
    to the exact form and to the oracle at a size where the join is the automatic form (3 000 sketches, families of every
    size up to 60), whole and as the shares of 3 ranks; and a matrix built to overflow a first-level capacity -- every hash
    with the same eight digit bits -- raises the overflow flag and is answered by the exact form instead, same bytes."""
    rng = np.random.default_rng(2024)
    n, s = 3000, 1000
    hashes, lens = random_sketches(rng, n, s, shared_groups=120, min_len=600)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=32)
    assert len(want) > 500
    ctx.profile(True)
    for fused in (0, 1):
        opts(join_fused=fused)
        ctx.profile_reset()
        assert ctx.precluster(sk, np.float32(0.9)).tobytes() == want.tobytes(), fused
        st = ctx.kernel_stats()
        assert st["pair_join"][0] > 0 and st["pair_intersect_tile"][0] == 0      # the join form ran and did not decline
        parts = [ctx.precluster_ranks(sk, np.float32(0.9), r, 3)[0] for r in range(3)]
        assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes(), fused
    sk.free()
    # every hash in ONE first-level bucket: bits 12..19 of the hash are the first digit (pairs_join.hip: ElemSrc::mix)
    n = 1500
    hashes, lens = random_sketches(rng, n, 256, shared_groups=60, min_len=200)
    valid = np.arange(256)[None, :] < lens[:, None]
    skew = (hashes & ~np.uint64(0xFF000)) | np.uint64(0x5A000)
    hashes = np.where(valid, skew, hashes)
    hashes = np.sort(hashes, axis=1)      # (clearing bits may reorder a row; rows stay distinct with overwhelming probability)
    assert all(len(np.unique(hashes[i, : lens[i]])) == lens[i] for i in range(0, n, 97))
    sk = ctx.sketches_from_host(hashes, lens, 21)
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=32)
    for fused in (0, 1):
        opts(join_fused=fused)
        assert ctx.precluster(sk, np.float32(0.9)).tobytes() == want.tobytes(), fused
    ctx.profile(False)
    sk.free()


@pytest.mark.parametrize("n,s,min_len,thr", [(37, 1000, None, 0.9), (260, 1000, 1, 0.9), (130, 256, 1, 0.8), (64, 1000, 1, 0.0), (90, 700, 300, 0.95),
                                             (50, 12, 1, 0.5), (300, 200, 64, 0.9)])
@never_run_on_hardware
def test_probe_kernel_arranged_form_matches_oracle(ctx, opts, n, s, min_len, thr):
    """ghip_options.probe_arranged (round 4): the dense probe kernel with the second cuckoo choice in the first
    one's residue class and every B row dealt to the lanes by bucket residue (pairs_probe.hip: pair_arrange_kernel,
    pair_probe_arranged_kernel) -- the same bytes as the oracle's pair loop and as the free form, whole, as the tile shares
    of 3 ranks, and on the (new x all) rectangle of an incremental run; sketch sizes on both table sizes (s <= 256 and
    s <= 1024), ragged rows, a table too small for the constraint (s = 12)."""
    rng = np.random.default_rng(n * 13 + s)
    hashes, lens = random_sketches(rng, n, s, shared_groups=6, min_len=min_len)
    want = oracle.distances_from_sketches(hashes, lens, np.float32(thr))
    ctx.profile(True)
    for arranged in (1, 4, 0):   # by table size (3 bits at 1 024 buckets, 2 below) / 4 constrained bucket bits / the free form
        opts(pair_form="probe", probe_arranged=arranged)
        sk = ctx.sketches_from_host(hashes, lens, 21)      # (the form is fixed when a matrix's tables are built: a fresh matrix)
        ctx.profile_reset()
        assert ctx.precluster(sk, np.float32(thr)).tobytes() == want.tobytes(), arranged
        assert ctx.kernel_stats()["pair_intersect_tile"][0] > 0      # the probe kernel ran (no fall-back to the merge path)
        parts = [ctx.precluster(sk, np.float32(thr), r, 3) for r in range(3)]
        assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes(), arranged
        lo = n // 2
        inc = ctx.precluster_from(sk, lo, np.float32(thr))
        assert inc.tobytes() == want[want["j"] >= lo].tobytes(), arranged
        sk.free()
    ctx.profile(False)


def test_merge_path_kernel_also_matches_oracle(ctx, opts):
    """The 64-way merge-path kernel (fallback form of pair_intersect_tile) on the same inputs."""
    opts(pair_form="merge")
    for n, s, min_len, thr in ((37, 1000, None, 0.9), (130, 256, 1, 0.8), (64, 1000, 1, 0.0)):
        rng = np.random.default_rng(n * 7 + s)
        hashes, lens = random_sketches(rng, n, s, shared_groups=5, min_len=min_len)
        sk = ctx.sketches_from_host(hashes, lens, 21)
        got = ctx.precluster(sk, np.float32(thr))
        assert got.tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(thr)).tobytes()
        assert ctx.last_pairs_compared == n * (n - 1) // 2


def test_sketch_holding_the_empty_marker_falls_back_exactly(ctx):
    """A real hash equal to 2^64-1 (the cuckoo empty-slot marker / row padding) must not change results:
    the probe form detects it and the call runs the merge-path kernel with explicit index guards."""
    rng = np.random.default_rng(99)
    hashes, lens = random_sketches(rng, 20, 64, shared_groups=2)
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    for g in (3, 4, 11):           # same family members share the maximal hash
        hashes[g, lens[g] - 1] = M
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(0.0))
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.0))
    assert got.tobytes() == want.tobytes()
    row = got[(got["i"] == 3) & (got["j"] == 4)][0]
    a, b = set(hashes[3, : lens[3]].tolist()), set(hashes[4, : lens[4]].tolist())
    assert int(row["common"]) == len(a & b) and int(M) in (a & b)


def test_ingest_yields_reference_genome_stats(ctx):
    """Assembly statistics come out of the same FASTA parse as the base streams
    (src/genome_stats.rs:61-86 goldens, plus the oracle on every fixture)."""
    g = ctx.genomes_from_files([fasta(n) for n in ALL + ["abisko_S2D10"]], io_threads=4)
    assert g.stats(len(ALL)) == (161, 6506, 8289)
    assert g.stats(ALL.index("set1_1mbp")) == (1, 0, 1_000_000)
    for i, n in enumerate(ALL):
        assert g.stats(i) == oracle.genome_stats(fasta(n))
    with pytest.raises(galah_amd.GalahHipError):
        ctx.genomes_from_host([b"ACGT" * 10]).stats(0)   # only file-backed genomes carry statistics


FIXTURE_GENOMES = ["set1_1mbp", "set1_500kb", "set2_1mbp", "set2_half", "abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13",
                   "antonio_MAG52", "antonio_MAG189", "clash_500kb", "abisko_S2D10", "abisko_S1D21", "abisko_S2M16"]


def test_sketch_matrix_save_load_against_the_oracle(ctx, tmp_path):
    """A persisted matrix ("GHIPSK02": names, seed, checksum) loads to the same rows and names, and the pair stage on the
    LOADED matrix gives what the oracle computes from the files; a damaged file and a foreign file are refused; a
    round-1/2 "GHIPSK01" file (no names) still loads."""
    import struct
    paths = [fasta(n) for n in FIXTURE_GENOMES]
    sk = ctx.sketch_files(paths, 21, 1000, 0, 4)
    p = str(tmp_path / "matrix.ghipsk")
    sk.save(p, paths, 0)
    back, names, seed = ctx.sketches_load_named(p)
    assert names == paths and seed == 0 and (back.kmer, back.size, len(back)) == (21, 1000, len(paths))
    h0, l0 = sk.to_host()
    h1, l1 = back.to_host()
    assert np.array_equal(h0, h1) and np.array_equal(l0, l1)
    want = oracle.distances(paths, np.float32(0.9))
    assert ctx.precluster(back, np.float32(0.9)).tobytes() == want.tobytes() and len(want) >= 20
    back.free()
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    bad = str(tmp_path / "damaged.ghipsk")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(galah_amd.GalahHipError, match="damaged"):
        ctx.sketches_load(bad)
    # a truncated file, and a header whose counts ask for terabytes: refused by SIZE before anything is
    # allocated -- an error code, never std::bad_alloc through the C boundary
    open(bad, "wb").write(bytes(raw[: len(raw) - 4096]))
    with pytest.raises(galah_amd.GalahHipError, match="truncated"):
        ctx.sketches_load(bad)
    huge = bytearray(open(p, "rb").read())
    huge[24:32] = struct.pack("<Q", (1 << 32) - 1)          # n: 4 294 967 295 rows of 1 000 hashes = 34 TB
    open(bad, "wb").write(bytes(huge))
    with pytest.raises(galah_amd.GalahHipError, match="GHIP_EIO"):
        ctx.sketches_load(bad)
    huge = bytearray(open(p, "rb").read())
    huge[32:40] = struct.pack("<Q", (1 << 40) - 1)          # names: 1 TB
    open(bad, "wb").write(bytes(huge))
    with pytest.raises(galah_amd.GalahHipError, match="GHIP_EIO"):
        ctx.sketches_load(bad)
    open(bad, "wb").write(b">not a matrix\nACGT\n")
    with pytest.raises(galah_amd.GalahHipError, match="not a sketch matrix"):
        ctx.sketches_load(bad)
    old = str(tmp_path / "v1.ghipsk")
    with open(old, "wb") as f:   # the round-1/2 format: magic, k, s, n, lens, hashes
        f.write(b"GHIPSK01" + struct.pack("<IIQ", 21, 1000, len(paths)) + l0.astype("<u4").tobytes() + h0.astype("<u8").tobytes())
    v1, v1_names, _ = ctx.sketches_load_named(old)
    assert v1_names == [""] * len(paths) and np.array_equal(v1.to_host()[0], h0)
    v1.free(); sk.free()


@pytest.mark.parametrize("form", [None, "merge", "join"])
def test_incremental_dereplication_on_a_saved_matrix(ctx, tmp_path, opts, form):
    """docs/preludes/cluster_prelude.md:13-15's workflow without re-sketching: a first run over 8 of the reference's fixture
    genomes persists its sketch matrix; a second run names the matrix and 6 NEW files, reads and sketches only those, runs
    the pair stage on the (new x all) rectangle and -- given the first run's pairs -- returns the cache a full run over all
    14 files produces, checked against oracle.distances; in every form of the pair stage."""
    if form:
        opts(pair_form=form)
    paths = [fasta(n) for n in FIXTURE_GENOMES]
    old, new = paths[:8], paths[8:]
    full = oracle.distances(paths, np.float32(0.9))
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=4)
    m1, m2 = str(tmp_path / "first.ghipsk"), str(tmp_path / "grown.ghipsk")
    first = pre.distances_and_save(old, m1)
    first_pairs = pre.last_pairs
    assert first_pairs.tobytes() == oracle.distances(old, np.float32(0.9)).tobytes() and len(first.items()) == len(first_pairs)
    names, cache = pre.distances_incremental(m1, new, saved_pairs=first_pairs, save_to=m2)
    assert names == paths and pre.last_pairs.tobytes() == full.tobytes()
    assert repr(cache) == repr(galah_amd.SortedPairGenomeDistanceCache.from_pairs(full))
    assert ctx.last_pairs_compared == 14 * 13 // 2 - 8 * 7 // 2   # the rectangle only
    _, only_new = pre.distances_incremental(m1, new)                # without the first run's pairs: those touching a new genome
    assert pre.last_pairs.tobytes() == full[full["j"] >= 8].tobytes()
    # the grown matrix serves the next run: nothing new to sketch, an empty rectangle
    names3, cache3 = pre.distances_incremental(m2, [], saved_pairs=full)
    assert names3 == paths and pre.last_pairs.tobytes() == full.tobytes()
    # a matrix made with other parameters is refused, not silently mixed
    with pytest.raises(RuntimeError, match="was made with"):
        galah_amd.FinchPreclusterer(0.9, 500, 21, ctx=ctx).distances_incremental(m1, new)


def test_sketch_sizes_beyond_the_lds_tiles(ctx, opts):
    """num_kmers has no bound in the reference (src/finch.rs:55-61): s = 10 000 -- candidate lists sorted in global memory,
    the pair stage through the inverted index, and through the global-memory dense kernel where the join declines
    (threshold 0, an empty sketch) or is switched off -- sketches and pair lists against the oracle."""
    s = 10_000
    seqs = [oracle.synth_genome(3, i // 3, i % 3, 400_000 + 1000 * i, 0.01) for i in range(6)] + [np.frombuffer(b"ACGTACGTAC", dtype=np.uint8)]
    g = ctx.genomes_from_host(seqs)
    sk = ctx.sketch_genomes(g, 21, s, 0)
    hashes, lens = sk.to_host()
    for i, q in enumerate(seqs):
        o = oracle.sketch_bytes(q, 21, s, 0)
        assert lens[i] == len(o) and np.array_equal(hashes[i, : lens[i]], o), i
    assert lens[0] == s and lens[6] == 0
    for thr, form in ((0.9, None), (0.0, None), (0.9, "merge"), (0.95, "merge")):
        opts(pair_form=form or "auto")
        got = ctx.precluster(sk, np.float32(thr))
        assert got.tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(thr)).tobytes(), (thr, form)
        opts(pair_form="auto")
    assert len(ctx.precluster(sk, np.float32(0.9))) >= 6   # the members of a species pair up (and the empty sketch with everything: NaN quirk)
    with pytest.raises(galah_amd.GalahHipError, match="65535"):
        ctx.sketch_genomes(g, 21, 70_000, 0)
    sk.free(); g.free()


def test_fused_sketch_and_index_equals_separate_passes(ctx):
    """ghip_sketch_and_index (one pass over the bases) == ghip_sketch_genomes + ghip_ani_index_build,
    on streams with N runs, short genomes and a repetitive genome (forces the sketch retry path)."""
    streams = _streams()
    names = [n for n in streams if n != "empty"] + ["empty"]
    g = ctx.genomes_from_host([streams[n] for n in names])
    sk1 = ctx.sketch_genomes(g, 21, 1000, 0)
    idx1 = ctx.ani_index_build(g, 15, 125, 20000)
    sk2, idx2 = ctx.sketch_and_index(g, 21, 1000, 0, 15, 125, 20000)
    h1, l1 = sk1.to_host()
    h2, l2 = sk2.to_host()
    assert np.array_equal(h1, h2) and np.array_equal(l1, l2)
    for a, b in zip(idx1.meta(), idx2.meta()):
        assert np.array_equal(a, b)
    n = len(names)
    pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.uint32)
    a1, f1 = ctx.ani_pairs(idx1, pairs, 0.0, want_af=True)
    a2, f2 = ctx.ani_pairs(idx2, pairs, 0.0, want_af=True)
    assert np.array_equal(a1, a2) and np.array_equal(f1, f2)
    osk = [oracle.AniSketch.from_bytes(streams[nm]) for nm in names]
    for x, (i, j) in enumerate(pairs):
        assert np.float32(oracle.ani_pair(osk[i], osk[j], 0.0)[0]) == a2[x], (names[i], names[j])
    # other k: the fused kernel only exists for k = 21, the call must still work
    sk3, idx3 = ctx.sketch_and_index(g, 15, 200, 0, 15, 125, 20000)
    h3, l3 = sk3.to_host()
    for i, nm in enumerate(names):
        o = oracle.sketch_bytes(streams[nm], 15, 200, 0)
        assert l3[i] == len(o) and np.array_equal(h3[i, : l3[i]], o)
    assert np.array_equal(ctx.ani_pairs(idx3, pairs, 0.0), a1)


def test_reference_membership_tests_through_hip(ctx):
    """The reference's finch+skani cluster tests (src/clusterer.rs:631-690) through the GPU path:
    cluster(genomes, FinchPreclusterer{0.9,1000,21}, clusterer{threshold, 0.2}) -> [[0,1,2,3]] @95, [[0,1,3],[2]] @99."""
    names = ["abisko_S1X13", "abisko_S2D19", "abisko_S3X12", "abisko_S2D13"]
    paths = [fasta(n) for n in names]
    for thr, want in ((95.0, [[0, 1, 2, 3]]), (99.0, [[0, 1, 3], [2]])):
        pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=4)
        cl = galah_amd.HipAniClusterer(thr, 0.2, ctx=ctx, io_threads=4)
        got = galah_amd.cluster(paths, pre, cl)
        assert sorted(got) == want   # the reference sorts before comparing (clusterer.rs:562-564)


def test_reference_cli_expectations_through_hip(ctx):
    """tests/test_cmdline.rs through the GPU path: :36-61 + :304-352 (Parks2020_reduced order from the ingest's
    own genome statistics, then finch + ANI -> one cluster, representative S2M.16) and :262-302
    (--min-aligned-fraction 0.2 -> one representative, 0.6 -> two); the representatives the output tests
    :62-160 and :184-216 expect."""
    names = ["abisko_S1D21", "abisko_S2M16"]
    checkm = {"abisko_S1D21": (95.21, 0.00), "abisko_S2M16": (95.92, 0.65)}   # test_cmdline.rs:14-16
    g = ctx.genomes_from_files([fasta(n) for n in names], io_threads=2)
    st = [g.stats(i) for i in range(2)]
    comp = np.array([checkm[n][0] for n in names], np.float32) / np.float32(100)
    cont = np.array([checkm[n][1] for n in names], np.float32) / np.float32(100)
    order = galah_amd.quality_order_parks2020_reduced(comp, cont, [s[0] for s in st], [s[1] for s in st])
    assert list(order) == [1, 0]
    paths = [fasta(names[i]) for i in order]
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=2)
    got = galah_amd.cluster(paths, pre, galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=2))
    assert got == [[0, 1]]
    paths = [fasta("set2_1mbp"), fasta("set2_half")]
    for min_af, want in ((0.2, [[0, 1]]), (0.6, [[0], [1]])):
        cl = galah_amd.HipAniClusterer(95.0, min_af, ctx=ctx, io_threads=2)
        assert sorted(galah_amd.cluster(paths, pre, cl)) == want
    # :62-119 (symlink-directory tests): [set1/500kb, set1/1mbp] -> 500kb.fna is the only representative;
    # :120-160 and :184-216 (name clash): both 500kb files are representatives, 1mbp.fna is not
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=2)
    assert galah_amd.cluster([fasta("set1_500kb"), fasta("set1_1mbp")], pre, cl) == [[0, 1]]
    got = galah_amd.cluster([fasta("clash_500kb"), fasta("set1_500kb"), fasta("set1_1mbp")], pre, cl)
    assert sorted(c[0] for c in got) == [0, 1] and [1, 2] in got
    # :1100-1125 (membership only; the order there comes from CheckM2)
    got = galah_amd.cluster([fasta(n) for n in ("set1_1mbp", "set1_500kb", "abisko_S1D21", "abisko_S2M16")], pre, cl)
    assert sorted(sorted(c) for c in got) == [[0, 1], [2, 3]]


def _dirty_streams(seed=5, n=6, length=20_000):
    """Streams where ~8 % of the bytes break k-mers (N, '-', lower case, arbitrary bytes), so that fewer than
    4096 valid 21-mers remain: with s = 4096 the sketch is the full k-mer set and every window is checked."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    junk = np.frombuffer(b"NN-nacgtRYx\x00\xff*", dtype=np.uint8)
    out = []
    for i in range(n):
        L = length + int(rng.integers(0, 700))
        s = rng.choice(acgt, size=L)
        bad = rng.random(L) < 0.08
        s[bad] = rng.choice(junk, size=int(bad.sum()))
        for edge in (64, 128, 16384, 16384 + 64):      # bad bytes right at lane / block boundaries, all alignments
            for d in range(-3, 4):
                if rng.random() < 0.5 and 0 <= edge + d < L:
                    s[edge + d] = ord("N")
        out.append(s)
    return out


def _clean(stream):
    ok = np.isin(stream, np.frombuffer(b"ACGT", dtype=np.uint8))
    return np.where(ok, stream, np.uint8(ord("N")))


def test_every_window_of_dirty_streams_minhash(ctx):
    streams = _dirty_streams()
    g = ctx.genomes_from_host(streams)
    for i, st in enumerate(streams):
        assert np.array_equal(g.to_host(i), _clean(st))          # caller bytes are sanitised on upload
    for seed in (0, 7):
        hashes, lens = ctx.sketch_genomes(g, 21, 4096, seed).to_host()
        sk2, _ = ctx.sketch_and_index(g, 21, 4096, seed)
        h2, l2 = sk2.to_host()
        assert np.array_equal(hashes, h2) and np.array_equal(lens, l2)
        for i, st in enumerate(streams):
            want = oracle.sketch_bytes(st, 21, 4096, seed)
            assert 1000 < len(want) < 4096                       # the whole k-mer set, nothing truncated
            assert lens[i] == len(want) and np.array_equal(hashes[i, : lens[i]], want), i


@pytest.mark.parametrize("c", [1, 3, 125])
def test_every_seed_of_dirty_streams(ctx, c):
    """Seed sets (code, packed location = chunk, strand, offset) of the fused and the standalone seeding pass against the
    oracle, dense (c = 1 keeps every valid 15-mer and overflows the per-block LDS buffers) to sparse."""
    import torch
    streams = _dirty_streams(seed=6, n=4)
    g = ctx.genomes_from_host(streams)
    osk = [oracle.AniSketch.from_bytes(_clean(s), 15, c, 5000) for s in streams]
    for idx in (ctx.ani_index_build(g, 15, c, 5000), ctx.sketch_and_index(g, 21, 1000, 0, 15, c, 5000)[1]):
        lay = idx.layout()
        glen, cap, cnt = idx.meta()
        code = torch.empty(int(lay.n_seed_slots), dtype=torch.int32, device=DEV)
        loc = torch.empty(int(lay.n_seed_slots), dtype=torch.int32, device=DEV)
        ctx.memcpy_d2d(code.data_ptr(), lay.d_seed_code, code.numel() * 4)
        ctx.memcpy_d2d(loc.data_ptr(), lay.d_seed_loc, loc.numel() * 4)
        ctx.synchronize()
        code, loc = code.cpu().numpy().view(np.uint32), loc.cpu().numpy().view(np.uint32)
        start = 0
        for i, o in enumerate(osk):
            assert cnt[i] == len(o.seeds()), (c, i, cnt[i], len(o.seeds()))
            got = sorted(zip(code[start:start + cnt[i]].tolist(), loc[start:start + cnt[i]].tolist()))
            want = sorted(zip(o.seeds().astype(np.uint32).tolist(), o.locs(5000).tolist()))
            assert got == want, (c, i)
            start += int(cap[i])


@pytest.mark.parametrize("fused", [0, pytest.param(1, marks=never_run_on_hardware)])
def test_join_form_of_the_pair_stage_matches_oracle(ctx, opts, fused):
    """The inverted-index form (pairs_join.hip; automatic from N >= 1200) forced on small inputs: same bytes as
    the oracle's pair loop, whole and sharded; and the inputs it must decline (threshold 0, empty sketches)
    still give the dense kernels' answer.  fused: its partitions in the fused form (ghip_options.join_fused: the first
    level into fixed-capacity buckets, single-launch scans) -- the same bytes."""
    opts(pair_form="join", join_fused=fused)
    ctx.profile(True)
    for n, s, min_len, thr, groups in ((37, 1000, None, 0.9, 5), (130, 256, 1, 0.8, 5), (300, 1000, 700, 0.9, 40),
                                       (64, 1000, 1, 0.0, 5), (2, 1000, None, 0.9, 1)):
        rng = np.random.default_rng(n * 11 + s)
        hashes, lens = random_sketches(rng, n, s, shared_groups=groups, min_len=min_len)
        sk = ctx.sketches_from_host(hashes, lens, 21)
        ctx.profile_reset()
        got = ctx.precluster(sk, np.float32(thr))
        want = oracle.distances_from_sketches(hashes, lens, np.float32(thr))
        assert got.tobytes() == want.tobytes(), (n, s, thr)
        assert ctx.last_pairs_compared == n * (n - 1) // 2
        st = ctx.kernel_stats()
        used_join = st["pair_join"][0] > 0 and st["pair_intersect_tile"][0] == 0
        assert used_join == (thr > 0.0), (n, thr, st)  # threshold 0 must be declined
        if thr > 0.0:
            for world in (2, 3):
                parts, compared = [], 0
                for r in range(world):
                    parts.append(ctx.precluster(sk, np.float32(thr), r, world))
                    compared += ctx.last_pairs_compared
                assert compared == n * (n - 1) // 2
                merged = np.sort(np.concatenate(parts), order=["i", "j"])
                assert merged.tobytes() == want.tobytes()
                # the multi-rank entry point: by default the join form is sharded too (records only of the rank's
                # (i + j) mod world pairs); with join_ranks = replicate it hands every rank the whole list
                parts, compared = [], 0
                for r in range(world):
                    part, replicated = ctx.precluster_ranks(sk, np.float32(thr), r, world)
                    compared += ctx.last_pairs_compared
                    assert not replicated and all((part["i"] + part["j"]) % world == r)
                    parts.append(part)
                assert compared == n * (n - 1) // 2
                assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes()
                opts(join_ranks="replicate")
                compared = 0
                for r in range(world):
                    whole, replicated = ctx.precluster_ranks(sk, np.float32(thr), r, world)
                    compared += ctx.last_pairs_compared
                    assert replicated and whole.tobytes() == want.tobytes()
                assert compared == n * (n - 1) // 2
                opts(join_ranks="hash")
        else:  # declined: the dense forms give each rank its share
            parts = []
            for r in range(3):
                part, replicated = ctx.precluster_ranks(sk, np.float32(thr), r, 3)
                assert not replicated
                parts.append(part)
            assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes()
    ctx.profile(False)
    # two empty sketches: (empty, empty) has ANI 1.0 -- the join finds no such pair; the host adds the pairs of empty sketches
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    hashes = np.full((5, 8), M, dtype=np.uint64)
    lens = np.array([0, 0, 3, 8, 8], dtype=np.uint32)
    hashes[2, :3] = [5, 9, 11]
    hashes[3] = [1, 5, 9, 11, 20, 30, 40, 50]
    hashes[4] = [5, 9, 11, 12, 13, 14, 15, 2**64 - 2]
    sk = ctx.sketches_from_host(hashes, lens, 21)
    assert ctx.precluster(sk, np.float32(0.5)).tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(0.5)).tobytes()
    # ONE empty sketch is enough: the reference's NaN semantics pair it with every other sketch at ANI 1.0
    lens1 = lens.copy(); lens1[1] = 2; hashes1 = hashes.copy(); hashes1[1, :2] = [7, 9]
    sk = ctx.sketches_from_host(hashes1, lens1, 21)
    want = oracle.distances_from_sketches(hashes1, lens1, np.float32(0.5))
    assert ctx.precluster(sk, np.float32(0.5)).tobytes() == want.tobytes() and (want["total"] == 0).sum() == 4
    # ... at any size and in every share: 3 empty sketches among 300, joined (no dense pass), whole and as 2 / 3 shares
    rng = np.random.default_rng(77)
    hashes, lens = random_sketches(rng, 300, 256, shared_groups=30, min_len=100)
    for e in (0, 17, 299):
        lens[e] = 0; hashes[e] = M
    sk = ctx.sketches_from_host(hashes, lens, 21)
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.9))
    assert (want["total"] == 0).sum() == 3 * 299 - 3
    ctx.profile(True); ctx.profile_reset()
    assert ctx.precluster(sk, np.float32(0.9)).tobytes() == want.tobytes()
    st = ctx.kernel_stats()
    assert st["pair_join"][0] > 0 and st["pair_intersect_tile"][0] == 0
    ctx.profile(False)
    for world in (2, 3):
        parts = [ctx.precluster(sk, np.float32(0.9), r, world) for r in range(world)]
        assert all(all((p["i"] + p["j"]) % world == r) for r, p in enumerate(parts))
        assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes()
        parts = [ctx.precluster_ranks(sk, np.float32(0.9), r, world)[0] for r in range(world)]
        assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes()
    # one big family (every genome shares most hashes with every other): records outnumber pairs -> declined
    rng = np.random.default_rng(5)
    hashes, lens = random_sketches(rng, 200, 1000, shared_groups=1)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    assert ctx.precluster(sk, np.float32(0.9)).tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(0.9)).tobytes()


def test_join_with_one_very_large_family_goes_hybrid(ctx):
    """A hash shared by more genomes than an element bucket of the join holds (one family of 1 300 among 2 600 sketches):
    the join marks the genomes of such buckets, keeps every pair with an unmarked member, and the pairs of two marked
    genomes come from a dense pass over the marked rows -- same bytes as the oracle's pair loop, whole and in shares, with
    both the join and a dense kernel on the clock."""
    rng = np.random.default_rng(99)
    n, s, fam = 2600, 64, 1300
    hashes, lens = random_sketches(rng, n, s, shared_groups=130)
    core = np.sort(rng.integers(0, 2**62, size=40, dtype=np.uint64))
    for i in range(fam):
        own = rng.integers(0, 2**63, size=s - len(core), dtype=np.uint64)
        u = np.unique(np.concatenate([core, own]))[:s]
        hashes[i] = np.uint64(0xFFFFFFFFFFFFFFFF)
        hashes[i, : len(u)] = u
        lens[i] = len(u)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.9))
    assert len(want) >= fam * (fam - 1) // 2
    ctx.profile(True); ctx.profile_reset()
    got = ctx.precluster(sk, np.float32(0.9))
    st = ctx.kernel_stats()
    ctx.profile(False)
    assert got.tobytes() == want.tobytes()
    assert st["pair_join"][0] > 0 and st["pair_intersect_tile"][0] > 0     # the join AND the dense pass over the family
    assert ctx.last_pairs_compared == n * (n - 1) // 2
    for world in (2, 3):
        parts = [ctx.precluster(sk, np.float32(0.9), r, world) for r in range(world)]
        assert all(all((p["i"] + p["j"]) % world == r) for r, p in enumerate(parts))
        assert np.sort(np.concatenate(parts), order=["i", "j"]).tobytes() == want.tobytes()


def test_join_form_takes_over_at_scale_with_awkward_families(ctx):
    """N = 2500 (the automatic switch to the join form): identical genomes, subset sketches, short sketches,
    one family of 120 -- every pair result equal to the oracle's (parallel) pair loop."""
    rng = np.random.default_rng(77)
    n, s = 2500, 1000
    hashes, lens = random_sketches(rng, n, s, shared_groups=300, min_len=400)
    # 40 identical copies of one sketch, and 30 strict prefixes of another (smaller genome of the same species)
    for i in range(100, 140):
        hashes[i], lens[i] = hashes[100], lens[100]
    base, base_len = hashes[200].copy(), int(lens[200])
    for x, i in enumerate(range(200, 230)):
        keep = base_len - 10 * x
        hashes[i] = np.uint64(0xFFFFFFFFFFFFFFFF)
        hashes[i, :keep] = base[:keep]
        lens[i] = keep
    # a family of 120 genomes sharing most hashes (records >> pairs inside the family, still sparse overall)
    pool = np.unique(rng.integers(0, 2**62, size=1300, dtype=np.uint64))
    for i in range(300, 420):
        pick = np.sort(rng.choice(pool, size=1000, replace=False))
        hashes[i], lens[i] = pick, 1000
    sk = ctx.sketches_from_host(hashes, lens, 21)
    ctx.profile(True); ctx.profile_reset()
    got = ctx.precluster(sk, np.float32(0.9))
    st = ctx.kernel_stats()
    ctx.profile(False)
    assert st["pair_join"][0] > 0 and st["pair_intersect_tile"][0] == 0      # the join form ran and did not decline
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=32)
    assert got.tobytes() == want.tobytes()
    assert len(got) > 120 * 119 // 2 + 40 * 39 // 2


def test_reference_contig_expectation_through_hip(ctx):
    """tests/test_cmdline.rs:482-505 (expected clusters of the records of contigs_specific.fna at 95 %) with every record
    a genome: finch precluster on the GPU, ANI with dense seeds (c = 1; at c = 30 a 1 kb contig keeps ~30 seeds, see
    tests/test_oracle_golden.py), host clusterer -- and every ANI equal to the oracle's."""
    from conftest import fasta_records
    names, seqs = fasta_records("contigs_specific")
    g = ctx.genomes_from_host(seqs)
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0, 15, 1, 20000)
    pairs = ctx.precluster(sk, np.float32(0.9))
    pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
    ani = ctx.ani_pairs(idx, pi, 0.15)
    osk = [oracle.AniSketch.from_bytes(s, 15, 1, 20000) for s in seqs]
    for x, (a, b) in enumerate(pi):
        assert np.float32(oracle.ani_pair(osk[a], osk[b], 0.15)[0]) == ani[x], (names[a], names[b])
    assert galah_amd.cluster_pairs(len(seqs), pairs, np.float32(95.0), ani) == [[0, 1, 2, 3, 4, 5], [6], [7], [8]]
    # and with the parameters `--small-genomes` would use (c = 30, ~35 seeds per 1 kb contig: the estimate is noisy -- one
    # contig of the family of six can fall a few hundredths under 95 % -- but it is what it is on both sides):
    # device == oracle value by value, hence the same clusters
    idx30 = ctx.ani_index_build(g, 15, 30, 20000)
    ani30 = ctx.ani_pairs(idx30, pi, 0.15)
    osk30 = [oracle.AniSketch.from_bytes(s, 15, 30, 20000) for s in seqs]
    for x, (a, b) in enumerate(pi):
        assert np.float32(oracle.ani_pair(osk30[a], osk30[b], 0.15)[0]) == ani30[x], (names[a], names[b])
    got30 = galah_amd.cluster_pairs(len(seqs), pairs, np.float32(95.0), ani30)
    assert got30 == oracle.cluster(len(seqs), oracle.Cache.from_pairs(pairs), 95.0, lambda a, b: oracle.ani_pair(osk30[a], osk30[b], 0.15)[0])
    assert [c for c in got30 if len(c) == 1 and c[0] >= 6] == [[6], [7], [8]]   # the unrelated contigs stay alone


def test_reference_cli_representative_list_and_github7_through_hip(ctx):
    """tests/test_cmdline.rs:161-181 and :417-440 through the GPU path (see tests/test_oracle_golden.py)."""
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=2)
    got = galah_amd.cluster([fasta(n) for n in ("clash_500kb", "set1_500kb", "set1_1mbp")], pre,
                            galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx, io_threads=2))
    assert got == [[1, 2], [0]]
    got = galah_amd.cluster([fasta(n) for n in ("antonio_MAG52", "antonio_MAG189")], pre,
                            galah_amd.HipAniClusterer(95.0, 0.6, ctx=ctx, io_threads=2))
    assert got == [[0, 1]]


def test_batched_files_entry_point_equals_one_batch(ctx):
    """ghip_sketch_and_index_files with a tiny batch size (several batches, concatenated on the device) ==
    one batch == the separate calls; statistics as the oracle's; sketches-only form for ghip_sketch_files."""
    names = ALL + ["abisko_S2D10"]
    paths = [fasta(n) for n in names]
    sk1, idx1, st1 = ctx.sketch_and_index_files(paths, io_threads=4)                          # one batch
    sk2, idx2, st2 = ctx.sketch_and_index_files(paths, io_threads=4, batch_bytes=3_000_000)   # ~6 batches
    h1, l1 = sk1.to_host()
    h2, l2 = sk2.to_host()
    assert np.array_equal(h1, h2) and np.array_equal(l1, l2) and np.array_equal(st1, st2)
    for a, b in zip(idx1.meta(), idx2.meta()):
        assert np.array_equal(a, b)
    for i, n in enumerate(names):
        assert tuple(int(x) for x in st2[i]) == oracle.genome_stats(fasta(n))
    n = len(names)
    pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.uint32)
    a1 = ctx.ani_pairs(idx1, pairs, 0.15)
    a2 = ctx.ani_pairs(idx2, pairs, 0.15)
    assert np.array_equal(a1, a2) and (a1 > 0).sum() >= 8
    g = ctx.genomes_from_files(paths, 4)
    assert np.array_equal(ctx.ani_pairs(ctx.ani_index_build(g), pairs, 0.15), a1)
    sk3, none, _ = ctx.sketch_and_index_files(paths, io_threads=2, batch_bytes=1, want_index=False)  # one file per batch
    h3, l3 = sk3.to_host()
    assert none is None and np.array_equal(h3, h1) and np.array_equal(l3, l1)
    h4, l4 = ctx.sketch_files(paths, 21, 1000, 0, io_threads=2).to_host()
    assert np.array_equal(h4, h1) and np.array_equal(l4, l1)


def test_ingest_forms_agree(ctx, tmp_path, opts):
    """Plain, gzip and multi-member gzip files (whose trailer under-reports the stream, forcing the two-phase
    ingest), CRLF line ends, no final newline, an empty file, lower case and IUPAC codes: same sketches and
    statistics as the oracle, in the pipelined and in the two-phase form."""
    import gzip
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGTacgtNRYn-", dtype=np.uint8)
    recs = [rng.choice(acgt, size=int(L), p=[.22, .22, .22, .22, .02, .02, .02, .02, .01, .01, .005, .005, .01]).tobytes().decode()
            for L in (70_000, 31, 0, 12_345)]
    def fasta_text(eol, final_newline=True):
        out = []
        for i, r in enumerate(recs):
            out.append(f">rec{i} some description")
            out += [r[j:j + 60] for j in range(0, len(r), 60)]
        t = eol.join(out)
        return (t + eol if final_newline else t).encode()
    files = {}
    files["plain.fna"] = fasta_text("\n")
    files["crlf.fna"] = fasta_text("\r\n")
    files["nofinal.fna"] = fasta_text("\n", final_newline=False)
    files["one.fna.gz"] = gzip.compress(fasta_text("\n"))
    half = fasta_text("\n")
    cut = half.index(b">rec2")
    files["multi.fna.gz"] = gzip.compress(half[:cut]) + gzip.compress(half[cut:])   # two members
    files["empty.fna"] = b""
    paths = []
    for name, data in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    want_sk = [oracle.sketch_file(p) for p in paths]
    want_st = [oracle.genome_stats(p) for p in paths]
    for p in paths[1:5]:
        assert np.array_equal(oracle.sketch_file(p), want_sk[0])   # all forms hold the same records
    streams = {}
    for form in ("packed", "ascii", "two-phase"):   # packed = 2-bit codes over PCIe, copied into place on the device
        opts(ingest_form=form)
        g = ctx.genomes_from_files(paths, 3)
        hashes, lens = ctx.sketch_genomes(g, 21, 1000, 0).to_host()
        for i, p in enumerate(paths):
            assert lens[i] == len(want_sk[i]) and np.array_equal(hashes[i, : lens[i]], want_sk[i]), (form, p)
            assert g.stats(i) == want_st[i], (form, p)
        streams[form] = [g.to_host(i).tobytes() for i in range(len(paths))]
    # the resident streams are the same whichever way they travelled (IUPAC codes, gaps, the 'N' after a record).  The
    # resident form keeps 2-bit codes + "is it A/C/G/T" per position, so every other stream byte -- 'N', needletail's
    # '-' for gaps -- reads back as 'N'
    assert streams["packed"] == streams["ascii"] == streams["two-phase"]
    assert b"-" not in streams["packed"][0] and streams["packed"][0].count(b"N") > 100
    for i, p in enumerate(paths):   # ... and the host parser's (ghip_fasta_stream, no GPU involved)
        host = galah_amd.fasta_stream(p)[0].tobytes()
        assert (b"-" in host) == (i < 5) and streams["packed"][i] == host.replace(b"-", b"N"), p
    with pytest.raises(galah_amd.GalahHipError):
        ctx.genomes_from_files([str(tmp_path / "missing.fna")], 1)
    bad = tmp_path / "notfasta.fna"
    bad.write_bytes(b"ACGT\n")
    with pytest.raises(galah_amd.GalahHipError):
        ctx.genomes_from_files([str(bad)], 1)


def test_empty_and_single_inputs(ctx):
    """No genome, one genome: every entry point returns empty results instead of failing."""
    sk, idx, st = ctx.sketch_and_index_files([], io_threads=4)
    assert len(sk) == 0 and idx.layout().n == 0 and st.shape == (0, 3)
    assert len(ctx.precluster(sk, np.float32(0.9))) == 0
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=0)     # io_threads <= 0 means 1
    cl = galah_amd.HipAniClusterer(95.0, 0.15, ctx=ctx)
    assert galah_amd.cluster([], pre, cl) == []
    assert galah_amd.cluster([fasta("set1_500kb")], pre, cl) == [[0]]
    assert len(pre.distances([fasta("set1_500kb")])) == 0
    assert ctx.ani_pairs(idx, np.zeros((0, 2), np.uint32), 0.15).shape == (0,)


def test_randomised_differential_runs():
    """A short run of the fuzzers (tests/fuzz_sketch.py, tests/fuzz_pairs.py -- seed 4 of it is the run that found the
    empty-sketch case of the join form -- and tests/fuzz_ani.py: related genomes with indels, shuffles, repeats)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for script, args in (("fuzz_pairs.py", ["45", "4"]), ("fuzz_sketch.py", ["12", "21"]), ("fuzz_ani.py", ["25", "3"]),
                         ("fuzz_ingest.py", ["25", "2"])):   # random FASTA files through every ingest form
        r = subprocess.run([sys.executable, os.path.join(here, script)] + args, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_cluster_on_the_resident_index_native_rounds(ctx):
    """ghip_cluster_index (clusterer::cluster with the device index as ClusterDistanceFinder, src/clusterer.rs:56-152 +
    src/skani.rs:718-788, whole in native code) against the lazy clusterer driven from the host language and against the
    oracle's clusterer fed every pair's ANI -- in genome order and in a quality order (positions of the order)."""
    n_species, members = 40, 6
    n = n_species * members
    g = ctx.genomes_synthetic(11, n_species, members, 200_000, 0.0253)
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
    pairs = ctx.precluster(sk, np.float32(0.9))
    assert len(pairs) >= n_species * members * (members - 1) // 2 * 0.9
    pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
    thr = np.float32(95.0)   # ~95 % ANI between members: some join their species' representative, some found a new one
    every = ctx.ani_pairs(idx, pi, 0.15)
    assert 0.1 < float(np.mean(every >= thr)) < 0.9

    got, st = ctx.cluster_index(idx, n, pairs, thr, 0.15)
    want, asked = galah_amd.cluster_pairs_lazy(n, pairs, thr, lambda e: ctx.ani_pairs(idx, pi[e], 0.15))
    assert got == want and st["asked"] == asked and 1 <= st["rounds"] <= members + 1
    look = {(int(a), int(b)): float(v) for (a, b), v in zip(pi, every)}
    assert got == oracle.cluster(n, oracle.Cache.from_pairs(pairs), float(thr), lambda a, b: look[(min(a, b), max(a, b))])

    order = np.random.default_rng(5).permutation(n).astype(np.uint32)
    got_o, st_o = ctx.cluster_index(idx, n, pairs, thr, 0.15, order)
    rank_of = np.empty(n, np.uint32)
    rank_of[order] = np.arange(n, dtype=np.uint32)
    re = pairs.copy()
    a, b = rank_of[pairs["i"]], rank_of[pairs["j"]]
    re["i"], re["j"] = np.minimum(a, b), np.maximum(a, b)
    perm = np.lexsort((re["j"], re["i"]))
    re, every_o = re[perm], every[perm]
    look_o = {(int(p["i"]), int(p["j"])): float(v) for p, v in zip(re, every_o)}
    assert got_o == oracle.cluster(n, oracle.Cache.from_pairs(re), float(thr), lambda a, b: look_o[(min(a, b), max(a, b))])
    assert sorted(x for c in got_o for x in c) == list(range(n))
    # with `order` the edges are renumbered to positions but their ANI is asked for as (p.i, p.j) in GENOME order,
    # where the reference's calculate_ani takes (representative, genome) of the sorted list.  That is only right while the ANI
    # kernel is exactly symmetric -- so: (1) the same clusters as a run over the PHYSICALLY re-ordered genome list (its own
    # sketches, index, pair list, no `order`), and (2) ani(q, r) == ani(r, q) bit for bit, also for a pair of mixed density
    streams = [g.to_host(int(x)) for x in order]
    g2 = ctx.genomes_from_host(streams)
    sk2, idx2 = ctx.sketch_and_index(g2, 21, 1000, 0)
    pairs2 = ctx.precluster(sk2, np.float32(0.9))
    assert pairs2.tobytes() == re.tobytes()      # the renumbered, re-sorted list IS the pair list of the re-ordered genomes
    got_p, st_p = ctx.cluster_index(idx2, n, pairs2, thr, 0.15)
    assert got_p == got_o and st_p["asked"] == st_o["asked"]
    assert np.array_equal(ctx.ani_pairs(idx, pi[:, ::-1].copy(), 0.15), every)
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    small, big = acgt[rng.integers(0, 4, 300_000)], acgt[rng.integers(0, 4, 2_000_000)]
    cp = small.copy()
    hit = rng.random(len(cp)) < 0.03
    cp[hit] = acgt[(np.searchsorted(acgt, cp[hit]) + rng.integers(1, 4, size=int(hit.sum()))) % 4]
    big[500_000:800_000] = cp
    gm = ctx.genomes_from_host([small, big])
    im = ctx.ani_index_build(gm)
    fwd = ctx.ani_pairs(im, np.array([[0, 1]], dtype=np.uint32), 0.15)
    rev = ctx.ani_pairs(im, np.array([[1, 0]], dtype=np.uint32), 0.15)
    assert fwd[0] == rev[0] and 96.0 < fwd[0] < 98.0
    for h in (sk2, idx2, g2, im, gm):
        h.free()
    # no pairs: every genome its own cluster, no index needed
    assert ctx.cluster_index(None, 3, np.zeros(0, dtype=galah_amd.PAIR_DTYPE), thr)[0] == [[0], [1], [2]]
    with pytest.raises(galah_amd.GalahHipError):
        ctx.cluster_index(idx, n, pairs, thr, 0.15, np.zeros(n, np.uint32))   # not a permutation
