"""End to end against the oracle AT SCALE, under pytest: 10 000 synthetic genomes of 200 kb -- sketch ->
pairs -> lazy ANI on the device -> clusters -- compared with the oracle's sketches, the oracle's pair loop, the oracle's
ANI of every precluster pair and the oracle's run of the reference's greedy clusterer, in genome order and in a quality
order.  (Until round 4 the 10 000- and 50 000-genome end-to-end runs were checked only inside bench.py.)"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import galah_amd
import oracle
from conftest import never_run_on_hardware

SEED, MEMBERS, RATE = 42, 10, 0.0253


_FRONT = {}   # (n_species, length, min_ani, min_af) -> sketches, pairs, ANI table: the order only changes the clustering


def oracle_end_to_end(n_species, length, order=None, threads=None, min_ani=0.9, thr=95.0, min_af=0.15):
    """The whole path on the CPU oracle: (hashes, lens, pairs, {(i, j): ANI} of every precluster pair, clusters).  With
    `order` (order[x] = the genome that comes x-th) the clusterer sees the genomes in that order and the clusters hold
    positions -- what ghip_cluster_index / ghip_cluster_ranks return."""
    threads = threads or min(32, os.cpu_count() or 1)
    n = n_species * MEMBERS
    key = (n_species, length, min_ani, min_af)
    if key not in _FRONT:
        _FRONT.clear()   # (one workload's tables at a time: the 10 000-genome one is 80 MB of sketches)
        _FRONT[key] = _oracle_front(n_species, length, threads, min_ani, min_af)
    hashes, lens, pairs, look = _FRONT[key]
    return hashes, lens, pairs, look, _oracle_clusters(n, pairs, look, order, thr)


def _oracle_front(n_species, length, threads, min_ani, min_af):
    n = n_species * MEMBERS
    with ThreadPoolExecutor(threads) as ex:
        # (ctypes releases the GIL inside the oracle's C functions)
        sk = list(ex.map(lambda g: oracle.sketch_bytes(oracle.synth_genome(SEED, g // MEMBERS, g % MEMBERS, length, RATE), 21, 1000, 0), range(n)))
    lens = np.array([len(s) for s in sk], dtype=np.uint32)
    hashes = np.full((n, 1000), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    for g, s in enumerate(sk):
        hashes[g, : len(s)] = s
    pairs = oracle.distances_from_sketches(hashes, lens, np.float32(min_ani), threads=threads)
    # ANI of every precluster pair: batches by the first genome's species (a batch builds the ANI sketches it needs once)
    cut = np.searchsorted(pairs["i"] // MEMBERS, np.arange(n_species + 1))

    def batch(sp):
        rows = pairs[cut[sp]:cut[sp + 1]]
        sks = {}
        out = {}
        for p in rows:
            for g in (int(p["i"]), int(p["j"])):
                if g not in sks:
                    sks[g] = oracle.AniSketch.from_bytes(oracle.synth_genome(SEED, g // MEMBERS, g % MEMBERS, length, RATE))
            out[(int(p["i"]), int(p["j"]))] = oracle.ani_pair(sks[int(p["i"])], sks[int(p["j"])], min_af)[0]
        return out

    look = {}
    with ThreadPoolExecutor(threads) as ex:
        for part in ex.map(batch, range(n_species)):
            look.update(part)
    assert len(look) == len(pairs)
    return hashes, lens, pairs, look


def _oracle_clusters(n, pairs, look, order, thr):
    if order is None:
        clusters = oracle.cluster(n, oracle.Cache.from_pairs(pairs), thr, lambda a, b: look[(min(a, b), max(a, b))])
    else:
        order = np.asarray(order, dtype=np.int64)
        rank_of = np.empty(n, np.int64)
        rank_of[order] = np.arange(n)
        re = pairs.copy()
        a, b = rank_of[pairs["i"]], rank_of[pairs["j"]]
        re["i"], re["j"] = np.minimum(a, b), np.maximum(a, b)
        re = re[np.lexsort((re["j"], re["i"]))]
        clusters = oracle.cluster(n, oracle.Cache.from_pairs(re), thr,
                                  lambda x, y: look[(min(int(order[x]), int(order[y])), max(int(order[x]), int(order[y])))])
    return clusters


def test_oracle_harness_is_self_consistent():
    """(no GPU) the batched harness above == the oracle driven pair by pair, on 6 species, both orders."""
    n_species, length = 6, 60_000
    n = n_species * MEMBERS
    hashes, lens, pairs, look, clusters = oracle_end_to_end(n_species, length, threads=4)
    streams = [oracle.synth_genome(SEED, g // MEMBERS, g % MEMBERS, length, RATE) for g in range(n)]
    sks = [oracle.AniSketch.from_bytes(s) for s in streams]
    assert all(np.array_equal(hashes[g, : lens[g]], oracle.sketch_bytes(streams[g], 21, 1000, 0)) for g in range(n))
    assert clusters == oracle.cluster(n, oracle.Cache.from_pairs(pairs), 95.0, lambda a, b: oracle.ani_pair(sks[a], sks[b], 0.15)[0])
    order = np.random.default_rng(3).permutation(n)
    _, _, _, _, oc = oracle_end_to_end(n_species, length, order=order, threads=4)
    # the same through a physically re-ordered genome list
    psk = [sks[int(g)] for g in order]
    ph, pl = hashes[order], lens[order]
    pp = oracle.distances_from_sketches(ph, pl, np.float32(0.9))
    assert oc == oracle.cluster(n, oracle.Cache.from_pairs(pp), 95.0, lambda a, b: oracle.ani_pair(psk[a], psk[b], 0.15)[0])
    assert sorted(x for c in oc for x in c) == list(range(n)) and len(pairs) >= n_species * 40


@pytest.mark.gpu
@never_run_on_hardware
@pytest.mark.parametrize("quality_order", [False, True])
def test_10k_genomes_end_to_end_against_the_oracle(ctx, quality_order):
    """10 000 x 200 kb (1 000 species x 10, ~95 % ANI: the north-star's shape at a CPU-affordable length): every sketch, the
    whole pair list, the clusters and the number of ANI pairs asked -- device == oracle; and a sample of the device's ANI
    values against the oracle's table."""
    from galah_amd.distributed import DereplicationJob
    n_species, length = 1000, 200_000
    n = n_species * MEMBERS
    order = None
    if quality_order:   # CheckM2-style qualities -> Parks2020_reduced order (src/cluster_argument_parsing.rs:1078-1092)
        rng = np.random.default_rng(50)
        order = galah_amd.quality_order_parks2020_reduced(rng.uniform(70, 100, n).astype(np.float32) / np.float32(100),
                                                          rng.uniform(0, 5, n).astype(np.float32) / np.float32(100),
                                                          rng.integers(1, 400, n), rng.integers(0, 20000, n))
    job = DereplicationJob(ctx, 0, 1, n_genomes=n, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15, lazy_ani=True)
    if order is not None:
        job.set_order(order)
    job.load_synthetic(SEED, MEMBERS, length, RATE)
    res = job.step()
    got_h, got_l = job.sketches_to_host()
    hashes, lens, pairs, look, clusters = oracle_end_to_end(n_species, length, order=order)
    assert np.array_equal(got_l, lens) and np.array_equal(got_h, hashes)
    assert res["pairs"].tobytes() == pairs.tobytes() and len(pairs) >= 40_000
    assert res["clusters"].tolist() == clusters
    assert 1_000 < len(clusters) < 9_000 and res["ani_pairs_asked"] < len(pairs)
    # the device's ANI values (the lazy rounds keep theirs inside the library): a sample through ghip_ani_pairs
    sk, idx = ctx.sketch_and_index(job.genomes, 21, 1000, 0)
    sample = pairs[:: max(1, len(pairs) // 600)]
    pi = np.stack([sample["i"], sample["j"]], axis=1).astype(np.uint32)
    ani = ctx.ani_pairs(idx, pi, 0.15)
    assert all(np.float32(look[(int(a), int(b))]) == v for (a, b), v in zip(pi, ani))
    sk.free(); idx.free()
