"""world_size > 1 on CPU: the library's exchange logic (galah_amd/csrc/comm.cpp) over its host-callback transport fed by
torch.distributed's gloo backend -- no GPU, no compute.  What runs here is what the GPU ranks run above the two
primitives: the variable-length gathers, the (i, j) merge of candidate shares, the sharding rules."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from galah_amd import PAIR_DTYPE  # noqa: E402
from galah_amd.distributed import Comm, shard_range, tile_pairs_of_rank  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _share(rank, world, n=200, seed=3):
    """Deterministic candidate list; rank's share = pairs with (i + j) % world == rank (the join form's deal)."""
    rng = np.random.default_rng(seed)
    rows = sorted({(int(a), int(b)) for a, b in rng.integers(0, n, size=(900, 2)) if a < b})
    allp = np.zeros(len(rows), dtype=PAIR_DTYPE)
    for x, (i, j) in enumerate(rows):
        allp[x] = (i, j, (i * 7 + j) % 1000, 1000 + (i + j) % 900, np.float32(0.9 + ((i * 31 + j) % 100) / 1000.0))
    mine = allp[(allp["i"] + allp["j"]) % world == rank]
    return allp, mine


def _graph(n, seed):
    """A random precluster graph with families (the same on every rank) and a deterministic ANI per edge."""
    rng = np.random.default_rng(seed)
    fam = rng.integers(0, n // 6, size=n)
    rows = sorted({(int(a), int(b)) for a in range(n) for b in range(a + 1, n) if fam[a] == fam[b]} |
                  {(int(min(a, b)), int(max(a, b))) for a, b in rng.integers(0, n, size=(60, 2)) if a != b})
    pairs = np.zeros(len(rows), dtype=PAIR_DTYPE)
    for x, (i, j) in enumerate(rows):
        pairs[x] = (i, j, 500, 1500, np.float32(0.9 + ((i * 31 + j * 17) % 100) / 1000.0))
    table = (np.float32(93.0) + ((pairs["i"] * 7 + pairs["j"] * 13) % 50).astype(np.float32) / np.float32(10)).astype(np.float32)
    return n, pairs, table


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = Comm.from_torch_gloo(None, rank, world)   # no device context: host payloads only
    assert (comm.rank, comm.world, comm.transport) == (rank, world, "host-callback")
    out = {"rank": rank}
    # fixed-size host all-gather
    got = comm.allgather_host(np.arange(5, dtype=np.int64) + 100 * rank)
    out["fixed"] = got.tolist()
    # candidate shares -> whole list in (i, j) order; rank 1 also contributes nothing in a second round (ragged)
    allp, mine = _share(rank, world)
    merged = comm.allgather_pairs(mine)
    out["merged_ok"] = merged.tobytes() == allp.tobytes()
    merged2 = comm.allgather_pairs(mine if rank != 1 else mine[:0])
    want2 = allp[(allp["i"] + allp["j"]) % world != 1]
    out["ragged_ok"] = merged2.tobytes() == want2.tobytes()
    out["empty_ok"] = len(comm.allgather_pairs(mine[:0])) == 0
    # a device collective on a communicator without a device context is an error, not a crash
    try:
        comm.allgather_device(0, 0, 16)
        out["nodev"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["nodev"] = type(e).__name__
    try:
        comm.exchange_device(0, [0] * (world + 1), 0, [0] * (world + 1))
        out["nodev_exchange"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["nodev_exchange"] = type(e).__name__
    # phase boundary (ghip_comm_agree): all fine -> nobody raises; the last rank reports a failure -> EVERY rank raises
    # together, that rank with its own code, the others with GHIP_EPEER naming it; the communicator stays usable
    comm.agree(0)
    try:
        comm.agree(3 if rank == world - 1 else 0)
        out["agree"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["agree"] = str(e)
    out["after_agree"] = comm.allgather_host(np.int64([rank])).ravel().tolist()
    # the clusterer's lazy ANI rounds dealt over the ranks with the HOST's batched ANI answering (ghip_cluster_lazy_comm): the
    # dealing, the per-round gather behind a status word and the filing in request order are what the GPU ranks run
    # (ghip_cluster_index_comm shares the code); here the "ANI" is a table every rank can compute
    n, pairs, table = _graph(200, 11)
    answered = []

    def ani_of(edges):
        answered.append(len(edges))
        assert all(int(pairs["i"][e]) // ((n + world - 1) // world) == rank for e in edges)   # only the edges this rank owns
        return table[edges]

    for order in (None, np.random.default_rng(4).permutation(n)):
        clusters, st = comm.cluster_lazy(n, pairs, np.float32(95.0), ani_of, order)
        key = "lazy" if order is None else "lazy_order"
        out[key] = {"clusters": clusters.tolist(), "asked": st["asked"], "rounds": st["rounds"], "asked_here": st["asked_here"]}
    # a callback that fails on ONE rank: that rank raises its own exception, the others GHIP_EPEER -- nobody hangs
    def bad(edges):
        if rank == world - 1:
            raise ValueError("calculate_ani failed here")
        return table[edges]
    try:
        comm.cluster_lazy(n, pairs, np.float32(95.0), bad)
        out["lazy_fail"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["lazy_fail"] = f"{type(e).__name__}: {e}"
    out["after_fail"] = comm.allgather_host(np.int64([rank])).ravel().tolist()
    # settings that decide the collectives: ONE rank holds another lazy_flush_below -- its rounds would ask for other
    # request lists, i.e. other gather sizes.  Every rank gets GHIP_EINVAL at the head of the rounds; with the setting
    # restored the same call runs.  (A host-payload communicator has no context: the process-wide options are its settings.)
    import galah_amd
    saved = galah_amd.get_options()
    if rank == world - 1:
        galah_amd.set_options(None, lazy_flush_below=saved["lazy_flush_below"] + 7)
    try:
        comm.cluster_lazy(n, pairs, np.float32(95.0), lambda e: table[e])
        out["settings"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["settings"] = str(e)
    try:
        comm.agree(0)
        out["settings_agree"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["settings_agree"] = str(e)
    galah_amd.set_options(None, **saved)
    comm.agree(0)
    out["after_settings"] = comm.cluster_lazy(n, pairs, np.float32(95.0), lambda e: table[e])[0].tolist()
    # ONE rank passes a pair list that names a genome out of range: it gets its own GHIP_EINVAL, its peers GHIP_EPEER -- nobody
    # is left waiting in the first round's gather
    bad_pairs = pairs.copy()
    if rank == 0:
        bad_pairs["j"][3] = n + 5
    try:
        comm.cluster_lazy(n, bad_pairs, np.float32(95.0), lambda e: table[e])
        out["bad_args"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["bad_args"] = str(e)
    out["after_bad_args"] = comm.allgather_host(np.int64([rank])).ravel().tolist()
    q.put(out)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_logic_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=300) for _ in range(world)), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r, o in enumerate(outs):
        assert o["fixed"] == [[100 * x + y for y in range(5)] for x in range(world)]
        assert o["merged_ok"] and o["ragged_ok"] and o["empty_ok"]
        assert o["nodev"] == "GalahHipError" and o["nodev_exchange"] == "GalahHipError"
        assert o["agree"].startswith("GHIP_EHIP" if r == world - 1 else "GHIP_EPEER"), o["agree"]
        assert r == world - 1 or f"rank {world - 1} failed" in o["agree"]
        assert o["after_agree"] == list(range(world))
    # the lazy rounds over the ranks == one rank's ghip_cluster_lazy == the oracle's clusterer with every ANI known
    import galah_amd
    import oracle
    n, pairs, table = _graph(200, 11)
    want, asked = galah_amd.cluster_pairs_lazy(n, pairs, np.float32(95.0), lambda e: table[e])
    look = {(int(p["i"]), int(p["j"])): float(v) for p, v in zip(pairs, table)}
    assert want == oracle.cluster(n, oracle.Cache.from_pairs(pairs), 95.0, lambda a, b: look[(min(a, b), max(a, b))])
    order = np.random.default_rng(4).permutation(n)
    rank_of = np.empty(n, np.int64)
    rank_of[order] = np.arange(n)
    re = pairs.copy()
    a, b = rank_of[pairs["i"]], rank_of[pairs["j"]]
    re["i"], re["j"] = np.minimum(a, b), np.maximum(a, b)
    perm = np.lexsort((re["j"], re["i"]))
    look_o = {(int(p["i"]), int(p["j"])): float(v) for p, v in zip(re[perm], table[perm])}
    want_o = oracle.cluster(n, oracle.Cache.from_pairs(re[perm]), 95.0, lambda x, y: look_o[(min(x, y), max(x, y))])
    for o in outs:
        assert o["lazy"]["clusters"] == want and o["lazy"]["asked"] == asked and o["lazy"]["rounds"] >= 1
        assert o["lazy_order"]["clusters"] == want_o
    assert sum(o["lazy"]["asked_here"] for o in outs) == asked and sum(1 for o in outs if o["lazy"]["asked_here"]) >= 2
    assert all(o["lazy"]["clusters"] == outs[0]["lazy"]["clusters"] for o in outs)
    for r, o in enumerate(outs):
        assert (o["lazy_fail"].startswith("ValueError") if r == world - 1 else "GHIP_EPEER" in o["lazy_fail"]), o["lazy_fail"]
        assert o["after_fail"] == list(range(world))
        assert o["settings"].startswith("GHIP_EINVAL") and "different ghip_options" in o["settings"], o["settings"]
        assert o["settings_agree"].startswith("GHIP_EINVAL"), o["settings_agree"]
        assert o["after_settings"] == want
        assert o["bad_args"].startswith("GHIP_EINVAL" if r == 0 else "GHIP_EPEER"), o["bad_args"]
        assert o["after_bad_args"] == list(range(world))


def test_single_rank_communicator_needs_no_transport():
    comm = Comm.single(None)
    assert (comm.rank, comm.world, comm.transport) == (0, 1, "self")
    allp, _ = _share(0, 1)
    assert comm.allgather_pairs(allp).tobytes() == allp.tobytes()
    assert comm.allgather_host(np.float32([1.5, 2.5])).tolist() == [[1.5, 2.5]]
    comm.agree(0)
    # the lazy rounds with the host's ANI on a one-rank communicator == ghip_cluster_lazy, for a few random graphs and orders
    import galah_amd
    for seed in range(6):
        n, pairs, table = _graph(60 + 37 * seed, seed)
        want, asked = galah_amd.cluster_pairs_lazy(n, pairs, np.float32(95.0), lambda e: table[e])
        got, st = comm.cluster_lazy(n, pairs, np.float32(95.0), lambda e: table[e])
        assert got == want and st["asked"] == asked == st["asked_here"]
        order = np.random.default_rng(seed).permutation(n)
        got_o, _ = comm.cluster_lazy(n, pairs, np.float32(95.0), lambda e: table[e], order)
        assert sorted(x for c in got_o for x in c) == list(range(n)) and len(got_o) >= 1
    with pytest.raises(ValueError):
        comm.cluster_lazy(n, pairs, np.float32(95.0), lambda e: (_ for _ in ()).throw(ValueError("boom")))


def test_shard_ranges_cover_and_tiles_partition():
    for n in (1, 7, 8, 9, 100, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count, block = shard_range(n, r, world)
                assert count <= block
                seen += list(range(first, first + count))
            assert seen == list(range(n))
            tiles = [t for r in range(world) for t in tile_pairs_of_rank(n, 8, r, world)]
            nt = (n + 7) // 8
            assert sorted(tiles) == [(i, j) for i in range(nt) for j in range(i, nt)]
