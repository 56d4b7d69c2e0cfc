"""BASELINE.json configs as parity-test cases on the GPU (configs[1] is the bench workload; its
full-size parity check -- all 499 500 pairs and 256 sketches against the oracle -- runs inside
bench.py's cpu_baseline leg).  Sketch matrices here are planted: families of 10 share a pool of
hashes, unrelated genomes share nothing, so the exact expected pair list is the union of the
oracle's per-family lists and scales to any N."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import galah_amd
import oracle
from conftest import never_run_on_hardware
from galah_amd import PAIR_DTYPE

pytestmark = pytest.mark.gpu


def planted_sketches(n, s, seed, fam=10, q=0.6, hi=2**52):
    rng = np.random.default_rng(seed)
    hashes = np.empty((n, s), dtype=np.uint64)
    lens = np.full(n, s, dtype=np.uint32)
    nf = (n + fam - 1) // fam
    for f in range(nf):
        pool = rng.integers(0, hi, size=s, dtype=np.uint64)
        for g in range(f * fam, min((f + 1) * fam, n)):
            own = rng.integers(0, hi, size=s, dtype=np.uint64)
            u = np.unique(np.where(rng.random(s) < q, pool, own))
            while len(u) < s:
                u = np.unique(np.concatenate([u, rng.integers(0, hi, size=s - len(u), dtype=np.uint64)]))
            hashes[g] = u[:s]
    return hashes, lens


def expected_pairs(hashes, lens, min_ani, fam=10):
    n = hashes.shape[0]
    out = []
    for f0 in range(0, n, fam):
        p = oracle.distances_from_sketches(hashes[f0:f0 + fam], lens[f0:f0 + fam], np.float32(min_ani))
        p = p.copy()
        p["i"] += f0
        p["j"] += f0
        out.append(p)
    return np.concatenate(out) if out else np.zeros(0, PAIR_DTYPE)


def test_config3_10k_genomes_pair_stage_full_oracle(ctx):
    """configs[2]: 10 000 genomes -- the whole 5e7-pair stage against the oracle's pair loop."""
    n, s = 10_000, 1000
    hashes, lens = planted_sketches(n, s, 3)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(0.9))
    assert ctx.last_pairs_compared == n * (n - 1) // 2
    want = oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=os.cpu_count())
    assert got.tobytes() == want.tobytes()
    assert len(got) == (n // 10) * 45
    # N x N tiles dealt over 8 ranks: the shards partition the result (what 8 GPUs would each compute)
    parts, compared = [], []
    for r in range(8):
        parts.append(ctx.precluster(sk, np.float32(0.9), r, 8))
        compared.append(ctx.last_pairs_compared)
    merged = np.sort(np.concatenate(parts), order=["i", "j"])
    assert merged.tobytes() == got.tobytes()
    assert sum(compared) == n * (n - 1) // 2
    assert max(compared) < 1.02 * min(compared)  # the block-cyclic deal balances the pair work


def test_config4_small_sketch_high_pair_count(ctx):
    """configs[3]: many short contigs, small sketch (s'=256, build-defined -- finch itself refuses
    contigs, src/finch.rs:26-33).  4.5e8 pairs; expected list from the planted families."""
    n, s = 30_000, 256
    hashes, lens = planted_sketches(n, s, 4, q=0.7)
    lens[::7] = 100   # ragged: some contigs have fewer than s k-mers
    for g in range(0, n, 7):
        hashes[g, 100:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(0.9))
    want = expected_pairs(hashes, lens, 0.9)
    assert got.tobytes() == want.tobytes()
    assert ctx.last_pairs_compared == n * (n - 1) // 2


def test_config4_contigs_end_to_end_small_genomes(ctx):
    """Short contigs (2-20 kb) as individual genomes: sketches, pairs and ANI against the oracle."""
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = []
    for f in range(40):
        L = int(np.exp(rng.uniform(np.log(2000), np.log(20000))))
        anc = rng.choice(acgt, size=L)
        for m in range(5):
            c = anc.copy()
            mut = rng.random(L) < 0.02
            c[mut] = rng.choice(acgt, size=int(mut.sum()))
            contigs.append(c)
    g = ctx.genomes_from_host(contigs)
    sk = ctx.sketch_genomes(g, 21, 256, 0)
    hashes, lens = sk.to_host()
    for i in (0, 1, 57, 199):
        o = oracle.sketch_bytes(contigs[i], 21, 256, 0)
        assert lens[i] == len(o) and np.array_equal(hashes[i, : lens[i]], o)
    got = ctx.precluster(sk, np.float32(0.9))
    assert got.tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(0.9)).tobytes()
    idx = ctx.ani_index_build(g, 15, 30, 20000)  # --small-genomes ~ c = 30
    pi = np.stack([got["i"], got["j"]], axis=1).astype(np.uint32)
    ani = ctx.ani_pairs(idx, pi, 0.15)
    sks = {}
    for x in range(0, len(pi), 37):
        a, b = int(pi[x, 0]), int(pi[x, 1])
        for y in (a, b):
            if y not in sks:
                sks[y] = oracle.AniSketch.from_bytes(contigs[y], 15, 30, 20000)
        assert np.float32(oracle.ani_pair(sks[a], sks[b], 0.15)[0]) == ani[x]


def test_config5_50k_genomes_quality_order_two_stage(ctx):
    """configs[4]: 50 000 genomes, CheckM2-style qualities, genomes ordered by Parks2020_reduced
    (src/cluster_argument_parsing.rs:1078-1092), 90 % precluster / 95 % ANI two-stage clustering."""
    n, s = 50_000, 1000
    hashes, lens = planted_sketches(n, s, 5)
    rng = np.random.default_rng(55)
    completeness = rng.uniform(70, 100, n).astype(np.float32) / np.float32(100)
    contamination = rng.uniform(0, 5, n).astype(np.float32) / np.float32(100)
    num_contigs = rng.integers(1, 400, n)
    ambiguous = rng.integers(0, 20000, n)
    order = galah_amd.quality_order_parks2020_reduced(completeness, contamination, num_contigs, ambiguous)
    score = completeness.astype(np.float64) * 100. - 5. * contamination.astype(np.float64) * 100. \
        - 5. * num_contigs / 100. - 5. * ambiguous / 100000.
    assert np.all(np.diff(score[order]) <= 0)  # best genome first -> becomes the representative
    hashes, lens = hashes[order], lens[order]
    sk = ctx.sketches_from_host(hashes, lens, 21)
    pairs = ctx.precluster(sk, np.float32(0.9))
    assert ctx.last_pairs_compared == n * (n - 1) // 2
    # expected list: families are scattered by the quality order, so build it by original family
    inv = np.empty(n, dtype=np.int64)
    inv[order] = np.arange(n)
    rows = []
    for f0 in range(0, n, 10):
        members = np.sort(inv[f0:f0 + 10])
        p = oracle.distances_from_sketches(hashes[members], lens[members], np.float32(0.9)).copy()
        p["i"], p["j"] = members[p["i"]], members[p["j"]]
        rows.append(p)
    want = np.sort(np.concatenate(rows), order=["i", "j"])
    assert pairs.tobytes() == want.tobytes()
    # second stage: a deterministic synthetic ANI per pair; the host clusterer must match the oracle on a
    # sub-problem and satisfy the greedy invariants on the full one
    ani = (np.float32(90) + (pairs["common"].astype(np.float32) % np.float32(97)) / np.float32(9.7)).astype(np.float32)
    clusters = galah_amd.cluster_pairs(n, pairs, np.float32(95.0), ani)
    assert sorted(x for c in clusters for x in c) == list(range(n))
    look = {(int(p["i"]), int(p["j"])): a for p, a in zip(pairs, ani)}
    reps = np.array([c[0] for c in clusters])
    is_rep = np.zeros(n, bool)
    is_rep[reps] = True
    for c in clusters[:3000]:
        for m in c[1:]:
            assert look[(min(c[0], m), max(c[0], m))] >= np.float32(95.0)  # members reach their representative
    for (i, j), a in list(look.items())[:200000]:
        if is_rep[i] and is_rep[j]:
            assert a < np.float32(95.0)  # two representatives are never within the threshold
    sub = 3000
    keep = (pairs["i"] < sub) & (pairs["j"] < sub)
    oc = oracle.cluster(sub, oracle.Cache.from_pairs(pairs[keep]), 95.0,
                        lambda a, b: float(look[(min(a, b), max(a, b))]))
    assert galah_amd.cluster_pairs(sub, pairs[keep], np.float32(95.0), ani[keep]) == oc


@never_run_on_hardware
def test_config5_second_stage_through_the_ani_kernel_on_3000_genomes(ctx):
    """configs[4]'s second stage with the ANI KERNEL (the values in the test above are synthetic): a
    3 000-genome sub-problem of the same shape -- 300 species x 10 of 100 kb, CheckM2-style qualities, Parks2020_reduced
    order, 90 % precluster / 95 % ANI -- sketch -> pairs -> the native clusterer's lazy ANI rounds on the resident index,
    against the oracle's sketches, pair loop, ANI and greedy clusterer in that order."""
    from test_gpu_e2e_scale import MEMBERS, RATE, SEED, oracle_end_to_end
    n_sp, length = 300, 100_000
    m = n_sp * MEMBERS
    q = np.random.default_rng(56)
    sub_order = galah_amd.quality_order_parks2020_reduced(q.uniform(70, 100, m).astype(np.float32) / np.float32(100),
                                                          q.uniform(0, 5, m).astype(np.float32) / np.float32(100),
                                                          q.integers(1, 400, m), q.integers(0, 20000, m))
    g = ctx.genomes_synthetic(SEED, n_sp, MEMBERS, length, RATE)
    sk2, idx2 = ctx.sketch_and_index(g, 21, 1000, 0)
    p2 = ctx.precluster(sk2, np.float32(0.9))
    got, st = ctx.cluster_index(idx2, m, p2, np.float32(95.0), 0.15, sub_order)
    oh, ol, op, olook, oclusters = oracle_end_to_end(n_sp, length, order=sub_order)
    h2, l2 = sk2.to_host()
    assert np.array_equal(h2, oh) and np.array_equal(l2, ol) and p2.tobytes() == op.tobytes()
    assert got.tolist() == oclusters and st["asked"] <= len(op)
    for h in (sk2, idx2, g):
        h.free()


def test_config2_full_length_genomes_sample_vs_oracle(ctx):
    """configs[1] at full genome length (5 Mb) on a sample of 20 genomes: sketches, pairs, ANI."""
    seed, members, length, rate = 42, 10, 5_000_000, 0.0253
    g = ctx.genomes_synthetic(seed, 2, members, length, rate)
    sk = ctx.sketch_genomes(g, 21, 1000, 0)
    hashes, lens = sk.to_host()
    streams = [oracle.synth_genome(seed, i // members, i % members, length, rate) for i in range(20)]
    with ThreadPoolExecutor(min(20, os.cpu_count() or 1)) as ex:
        osk = list(ex.map(lambda b: oracle.sketch_bytes(b, 21, 1000, 0), streams))
        ask = list(ex.map(lambda b: oracle.AniSketch.from_bytes(b), streams))
    for i in range(20):
        assert lens[i] == 1000 and np.array_equal(hashes[i], osk[i])
        assert np.all(np.diff(hashes[i].astype(np.float64)) > 0)
    pairs = ctx.precluster(sk, np.float32(0.9))
    assert pairs.tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(0.9)).tobytes()
    assert len(pairs) == 90
    idx = ctx.ani_index_build(g)
    pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
    ani = ctx.ani_pairs(idx, pi, 0.15)
    for x in range(0, 90, 7):
        a, b = pi[x]
        assert np.float32(oracle.ani_pair(ask[a], ask[b], 0.15)[0]) == ani[x]
    assert np.all((ani > 94.5) & (ani < 95.5))


def test_ani_large_genomes_narrow_rounds(ctx):
    """Genomes above ~5 Mb join fewer bins per round (32 / 16 / 8) so that a round's seeds still fit the LDS stage, and
    hold more chunks than the default votes area: 9 Mb and 22 Mb pairs against the oracle, both orders."""
    for length, rate in ((9_000_000, 0.02), (22_000_000, 0.03)):
        g = ctx.genomes_synthetic(77, 1, 2, length, rate)
        idx = ctx.ani_index_build(g)
        ani, af = ctx.ani_pairs(idx, np.array([[0, 1], [1, 0]], dtype=np.uint32), 0.15, want_af=True)
        sks = [oracle.AniSketch.from_bytes(g.to_host(i)) for i in range(2)]
        want = oracle.ani_pair(sks[0], sks[1], 0.15)
        assert ani[0] == np.float32(want[0]) and ani[1] == ani[0], (length, ani, want)
        assert af[0, 0] == np.float32(want[1]) and af[0, 1] == np.float32(want[2])
        assert 90.0 < ani[0] < 99.0
        idx.free()
        g.free()


def test_ani_genomes_beyond_the_lds_votes_area(ctx):
    """A pair whose chunks (both genomes together) do not fit the LDS votes area -- above 58 Mb at the default 20 kb chunk --
    takes the general form of ani_pairs (per-chunk state in a global scratch region) instead of being refused: the
    reference hands any FASTA to skani (src/skani.rs:730-744).  31 Mb against 31 Mb (3 100 chunks), 62 Mb against 62 Mb
    and against a 4 Mb part of itself, both orders, against the oracle; short pair lists and long ones mix the two forms."""
    g = ctx.genomes_synthetic(78, 1, 2, 31_000_000, 0.03)
    idx = ctx.ani_index_build(g)
    sks = [oracle.AniSketch.from_bytes(g.to_host(i)) for i in range(2)]
    want = oracle.ani_pair_detail(sks[0], sks[1], 0.15)
    ani, af = ctx.ani_pairs(idx, np.array([[0, 1], [1, 0]], dtype=np.uint32), 0.15, want_af=True)
    assert ani[0] == np.float32(want[0]) and ani[1] == ani[0] and 90.0 < ani[0] < 99.0
    assert af[0, 0] == np.float32(want[1]) and af[0, 1] == np.float32(want[2])
    assert ctx.ani_pairs_detail(idx, np.array([[0, 1]], dtype=np.uint32))[0].tolist() == want[3]
    idx.free(); g.free()
    big = ctx.genomes_synthetic(79, 1, 2, 62_000_000, 0.02)
    a, b = big.to_host(0), big.to_host(1)
    part = b[30_000_000:34_000_000].copy()
    big.free()
    seqs = [a, b, part, oracle.synth_genome(80, 0, 0, 4_000_000, 0.0), oracle.synth_genome(80, 0, 1, 4_000_000, 0.01)]
    g = ctx.genomes_from_host(seqs)
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0, 15, 125, 20000)   # the fused pass seeds them too
    sks = [oracle.AniSketch.from_bytes(s) for s in seqs]
    pairs = np.array([(0, 1), (1, 0), (0, 2), (2, 1), (3, 4), (2, 3), (4, 3)], dtype=np.uint32)
    want = [oracle.ani_pair_detail(sks[x], sks[y], 0.15) for x, y in pairs]
    ani, af = ctx.ani_pairs(idx, pairs, 0.15, want_af=True)
    assert [float(v) for v in ani] == [float(np.float32(w[0])) for w in want], (ani, [w[0] for w in want])
    assert np.array_equal(af, np.float32([[w[1], w[2]] for w in want]))
    assert ctx.ani_pairs_detail(idx, pairs).tolist() == [w[3] for w in want]
    assert 95.5 < ani[0] < 96.7 and 95.0 < ani[2] < 97.0 and ani[4] > 98.5 and ani[5] == 0   # two copies at 2 % each: ~96 %
    many = np.concatenate([pairs] * 40)   # long list: the LDS form runs 8 waves per pair, the general form beside it
    assert np.array_equal(ctx.ani_pairs(idx, many, 0.15), np.concatenate([ani] * 40))
    sk.free(); idx.free(); g.free()


def test_ani_tandem_repeats_skewed_segments(ctx):
    """Genomes that are mostly ONE short unit repeated: a handful of distinct seed codes, hundreds of copies each.  The
    seeding pass files seeds under 8 segments of the bin hash with equal capacities -- here nearly all land in two or
    three, the segment overflows and the index is rebuilt with exact capacities; a bin then holds hundreds of seeds, so a
    round's run exceeds the LDS stage and ani_pairs walks it in global memory.  Fused and standalone passes, dense and
    sparse seeds, against the oracle."""
    rng = np.random.default_rng(11)
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1500)
    flank = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=400_000)
    def variant(rate):
        g = np.concatenate([flank[:200_000]] + [unit] * 300 + [flank[200_000:]])
        m = rng.random(len(g)) < rate
        g[m] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(m.sum()))
        return g
    seqs = [variant(0.0), variant(0.01), variant(0.03), rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=400_000),
            np.concatenate([unit] * 300)]   # the last one: nothing but the repeat
    g = ctx.genomes_from_host(seqs)
    pairs = np.array([(a, b) for a in range(5) for b in range(5) if a != b], dtype=np.uint32)
    for c in (125, 9):
        osk = [oracle.AniSketch.from_bytes(s, 15, c, 20000) for s in seqs]
        want = np.float32([oracle.ani_pair(osk[a], osk[b], 0.15)[0] for a, b in pairs])
        sk, fused = ctx.sketch_and_index(g, 21, 1000, 0, 15, c, 20000)
        alone = ctx.ani_index_build(g, 15, c, 20000)
        for idx in (fused, alone):
            assert np.array_equal(ctx.ani_pairs(idx, pairs, 0.15), want), c
            idx.free()
        sk.free()
    assert want[0] > 95 and want[2] == 0   # the variants are related (their unique flanks align), the random genome is not


def test_config4_full_size_100k_contig_sketches(ctx):
    """configs[3] at its stated size: 100 000 contigs, small sketches (s' = 256): 5e9 pairs through the join form; the
    expected list from the planted families (the oracle's pair loop per family); every 9th contig ragged."""
    n, s = 100_000, 256
    hashes, lens = planted_sketches(n, s, 14, q=0.7)
    lens[::9] = 90
    hashes[::9, 90:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    sk = ctx.sketches_from_host(hashes, lens, 21)
    got = ctx.precluster(sk, np.float32(0.9))
    assert ctx.last_pairs_compared == n * (n - 1) // 2
    assert got.tobytes() == expected_pairs(hashes, lens, 0.9).tobytes()
    sk.free()


def test_config4_5000_real_contigs_files_to_clusters_small_genomes(ctx, tmp_path):
    """5 000 contigs of 2-20 kb as FASTA files through the whole path with the small-genomes seed density
    (FinchPreclusterer + HipAniClusterer(small_genomes=True), one ingest): clusters equal the oracle's run of the
    reference's greedy algorithm on the oracle's sketches and ANI."""
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs, paths = [], []
    for f in range(1000):
        L = int(np.exp(rng.uniform(np.log(2000), np.log(20000))))
        anc = rng.choice(acgt, size=L)
        for m in range(5):
            c = anc.copy()
            mut = rng.random(L) < (0.01 if m < 3 else 0.04)
            c[mut] = rng.choice(acgt, size=int(mut.sum()))
            contigs.append(c)
    for i, c in enumerate(contigs):
        p = tmp_path / f"c{i:05d}.fna"
        p.write_bytes(b">contig%d\n" % i + c.tobytes() + b"\n")
        paths.append(str(p))
    n = len(paths)
    pre = galah_amd.FinchPreclusterer(0.9, 1000, 21, ctx=ctx, io_threads=16)
    cl = galah_amd.HipAniClusterer(95.0, 0.15, small_genomes=True, ctx=ctx, io_threads=16)
    got = galah_amd.cluster(paths, pre, cl)
    assert sorted(x for c in got for x in c) == list(range(n))
    with ThreadPoolExecutor(os.cpu_count()) as ex:
        osk = list(ex.map(lambda b: oracle.sketch_bytes(b.tobytes() + b"N", 21, 1000, 0), contigs))
        ask = list(ex.map(lambda b: oracle.AniSketch.from_bytes(b.tobytes() + b"N", 15, 30, 20000), contigs))
    hashes = np.full((n, 1000), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    for i, o in enumerate(osk):
        hashes[i, : len(o)] = o
        lens[i] = len(o)
    opairs = oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=os.cpu_count())
    assert pre.last_pairs.tobytes() == opairs.tobytes()
    want = oracle.cluster(n, oracle.Cache.from_pairs(opairs), 95.0, lambda a, b: oracle.ani_pair(ask[a], ask[b], 0.15)[0])
    assert got == want
    assert 1000 <= len(got) <= 2200   # ~1 000 families, the 4 %-mutated members partly on their own


def test_config2_1000_full_length_genomes_sketch_stage_sampled(ctx):
    """The sketch stage at configs[1] scale under pytest (bench.py checks the same inside its timed run): 1 000 x 5 Mb
    through the fused pass; 24 sampled sketch rows and the seed counts of 8 genomes against the oracle, every row
    strictly ascending and full."""
    seed, members, length, rate = 42, 10, 5_000_000, 0.0253
    g = ctx.genomes_synthetic(seed, 100, members, length, rate)
    sk, idx = ctx.sketch_and_index(g, 21, 1000, 0)
    hashes, lens = sk.to_host()
    assert hashes.shape == (1000, 1000) and np.all(lens == 1000)
    assert np.all(hashes[:, 1:] > hashes[:, :-1])
    _, _, cnt = idx.meta()
    rows = list(range(0, 1000, 43))[:24]
    with ThreadPoolExecutor(os.cpu_count()) as ex:
        streams = list(ex.map(lambda r: oracle.synth_genome(seed, r // members, r % members, length, rate), rows))
        want = list(ex.map(lambda b: oracle.sketch_bytes(b, 21, 1000, 0), streams))
        seeds = list(ex.map(lambda b: oracle.AniSketch.from_bytes(b).nseeds, streams[:8]))
    for r, w in zip(rows, want):
        assert np.array_equal(hashes[r], w), r
    assert [int(cnt[r]) for r in rows[:8]] == seeds
    for h in (sk, idx, g):
        h.free()


def test_bench_line_contract():
    """The driver's contract for `bench.py`: exactly ONE line on stdout, a JSON object with the agreed keys, the roofline
    and CPU-baseline objects of the hot-path tier -- on a small workload (the default one is the driver's to run), once as
    a plain invocation and once self-launched with two ranks on the one GPU (host transport)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--species", "6", "--members", "4", "--length", "300000", "--steps", "2", "--warmup", "1", "--no-extras"]
    for extra, env in (([], {}), (["--gpus", "2"], {"GHIP_BENCH_BACKEND": "gloo"})):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + small + extra, capture_output=True, text=True,
                           timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout[:2000]
        d = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, k
        assert d["metric"].startswith("genome-pairs/sec") and d["unit"] == "genome-pairs/s" and d["higher_is_better"] is True
        assert d["n_gpus"] == (2 if extra else 1) and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"   # (--species: per-GPU work fixed)
        assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
        rf = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in rf, k
        assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
        if not extra:
            for cb in (d["cpu_baseline"], d["configs1_1k"]["cpu_baseline"]):   # the headline's bounded sample = the whole configs[1]-shaped run
                for k in ("value", "unit", "cores", "kind", "sample"):
                    assert k in cb, k
                assert cb["kind"] in ("port", "reference") and cb["value"] > 0
            assert "parity_checked" in d["configs1_1k"]["cpu_baseline"] and "frac_of_lds_roof" in d["configs1_1k"]["kernels"]["pair_intersect_tile"]
        else:
            assert len(d["per_rank"]) == 2 and d["config"]["transport"] == "host-callback"


def test_ingest_of_a_large_file_in_every_form(ctx, tmp_path, opts):
    """A 30 Mb genome (above the 24 MB an ASCII staging slot holds, inside what a packed one does) next to a small one:
    the resident stream is the same whichever way it travelled -- packed through a slot, as it is through a slot or a
    blocking copy, or by the two-phase form."""
    rng = np.random.default_rng(3)
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=30_000_000)]
    big[1_000_000:1_000_300] = ord("N")
    small = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=50_000)]
    paths = []
    for name, seq in (("big.fna", big), ("small.fna", small)):
        p = tmp_path / name
        body = seq.reshape(-1, 100)
        lines = np.concatenate([body, np.full((body.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
        p.write_bytes(b">contig1 x\n" + lines[: len(lines) // 2] + b">contig2\n" + lines[len(lines) // 2:])
        paths.append(str(p))
    want = [galah_amd.fasta_stream(p)[0].tobytes() for p in paths]
    for form in ("packed", "ascii", "pageable", "two-phase"):
        opts(ingest_form=form)
        g = ctx.genomes_from_files(paths, 4)
        for i in range(2):
            assert g.to_host(i).tobytes() == want[i], (form, i)
        g.free()
