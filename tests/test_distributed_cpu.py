"""world_size > 1 exchange logic on CPU: gloo backend, the oracle-backed FakeEngine in place of
the HIP engine.  Checks that the sharded job returns exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
from galah_amd.distributed import DereplicationJob, shard_range, tile_pairs_of_rank  # noqa: E402

SEED, MEMBERS, LENGTH, RATE, N = 5, 3, 60_000, 0.0253, 11  # 11 genomes: ragged last shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, replicate=False):
    from fake_engine import FakeEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    engine = FakeEngine()
    engine.replicate = replicate   # True: the pair stage hands every rank the whole list (what the join form does)
    job = DereplicationJob(None, rank, world, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0),
                           min_af=0.15, engine=engine)
    job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
    res = job.step()
    compared = torch.tensor([job.last_pairs_compared], dtype=torch.int64)
    dist.all_reduce(compared)
    if rank == 0:
        q.put({"clusters": res["clusters"], "pairs": res["pairs"].tobytes(), "ani": res["pair_ani"].tobytes(),
               "compared": int(compared.item())})
    dist.barrier()
    dist.destroy_process_group()


def _single():
    from fake_engine import FakeEngine
    job = DereplicationJob(None, 0, 1, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0),
                           min_af=0.15, engine=FakeEngine())
    job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
    return job.step()


@pytest.mark.parametrize("world,replicate", [(2, False), (3, False), (2, True)])
def test_sharded_job_equals_single_process(world, replicate):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, replicate)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _single()
    assert got["compared"] == N * (N - 1) // 2
    assert got["pairs"] == want["pairs"].tobytes()
    assert got["ani"] == want["pair_ani"].tobytes()
    assert got["clusters"] == want["clusters"]
    # and the single-process FakeEngine agrees with the plain oracle end to end
    streams = [oracle.synth_genome(SEED, g // MEMBERS, g % MEMBERS, LENGTH, RATE) for g in range(N)]
    sks = [oracle.AniSketch.from_bytes(s) for s in streams]
    for p, a in zip(want["pairs"], want["pair_ani"]):
        assert np.float32(oracle.ani_pair(sks[p["i"]], sks[p["j"]], 0.15)[0]) == a


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
