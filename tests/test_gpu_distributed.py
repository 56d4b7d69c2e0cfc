"""The real HipEngine under world_size 2: two processes sharing the one GPU of the test box,
gloo backend (RCCL refuses two ranks on one device).  Exercises every C-ABI call of the
sharded path -- synthetic_range, copy_into, wrap_device, precluster_shard, ANI export/wrap --
and checks the result against the single-process HIP run and the oracle."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SEED, MEMBERS, LENGTH, RATE, N = 5, 3, 200_000, 0.0253, 19


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, N=N, LENGTH=LENGTH):
    import torch

    import galah_amd
    from galah_amd.distributed import DereplicationJob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = galah_amd.Context(0)
    job = DereplicationJob(ctx, rank, world, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15)
    job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
    for _ in range(2):  # twice: the memory pool and the wrapped handles must survive re-use
        res = job.step()
    if rank == 0:
        hashes, lens = job.sketches_to_host()
        q.put({"clusters": res["clusters"], "pairs": res["pairs"].tobytes(), "ani": res["pair_ani"].tobytes(),
               "hashes": hashes.tobytes(), "lens": lens.tobytes()})
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_three_ranks_join_form_equals_single_rank(ctx):
    """N = 2100 short genomes on three ranks (ragged shards): the pair stage switches to the join form, which every
    rank runs in full and keeps whole (ghip_precluster_ranks: no exchange of candidate lists); families straddle the shard boundaries, so the ANI index
    slices are exchanged too."""
    import oracle
    from galah_amd.distributed import DereplicationJob
    n, length = 2100, 30_000
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 3, port, q, n, length)) for r in range(3)]
    for p in procs:
        p.start()
    got = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    job = DereplicationJob(ctx, 0, 1, n_genomes=n, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15)
    job.load_synthetic(SEED, MEMBERS, length, RATE)
    want = job.step()
    hashes, lens = job.sketches_to_host()
    assert got["hashes"] == hashes.tobytes() and got["lens"] == lens.tobytes()
    assert got["pairs"] == want["pairs"].tobytes()
    assert got["ani"] == want["pair_ani"].tobytes()
    assert got["clusters"] == want["clusters"]
    assert want["pairs"].tobytes() == oracle.distances_from_sketches(hashes, lens, np.float32(0.9), threads=32).tobytes()


def test_two_ranks_one_gpu_equals_single_rank(ctx):
    import oracle
    from galah_amd.distributed import DereplicationJob
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    job = DereplicationJob(ctx, 0, 1, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15)
    job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
    want = job.step()
    hashes, lens = job.sketches_to_host()
    assert got["hashes"] == hashes.tobytes() and got["lens"] == lens.tobytes()
    assert got["pairs"] == want["pairs"].tobytes()
    assert got["ani"] == want["pair_ani"].tobytes()
    assert got["clusters"] == want["clusters"]
    # oracle end to end
    streams = [oracle.synth_genome(SEED, g // MEMBERS, g % MEMBERS, LENGTH, RATE) for g in range(N)]
    opairs = oracle.distances_from_sketches(hashes, lens, np.float32(0.9))
    assert want["pairs"].tobytes() == opairs.tobytes()
    sks = [oracle.AniSketch.from_bytes(s) for s in streams]
    oc = oracle.cluster(N, oracle.Cache.from_pairs(opairs), 95.0, lambda a, b: oracle.ani_pair(sks[a], sks[b], 0.15)[0])
    assert want["clusters"] == oc


def _nccl_single_rank(port, q):
    """The collectives of galah_amd.distributed.Exchange issued on a ONE-rank RCCL group: RCCL refuses two ranks on
    one device, so this is as far as the `nccl` backend can be exercised on a one-GPU box -- dtypes, device
    tensors, object collectives and the bench's timing reduction all go through the real RCCL entry points."""
    import torch

    import galah_amd
    from galah_amd.distributed import DereplicationJob, Exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ex = Exchange(0, 1, force=True)      # issue the collectives on the one-rank group instead of short-circuiting
    assert not ex.stage_on_host
    h = torch.arange(6 * 8, dtype=torch.int64, device="cuda").reshape(6, 8) - 3
    assert torch.equal(ex.all_gather_blocks(h, 5), h[:5])                  # sketch rows (i64 view of u64 hashes)
    lens = torch.arange(6, dtype=torch.int32, device="cuda")
    assert torch.equal(ex.all_gather_blocks(lens, 6), lens)
    codes = torch.arange(50, dtype=torch.int16, device="cuda")             # ANI index slices travel as raw bytes
    assert torch.equal(ex.all_gather_flat(codes, [50]), codes)
    assert ex.all_gather_flat(codes[:0], [0]).numel() == 0
    ani = np.linspace(90, 100, 37).astype(np.float32)                      # ANI results: host array, one collective
    assert ex.all_gather_host_array(ani, [37]).tobytes() == ani.tobytes()
    pairs = np.zeros(3, dtype=galah_amd.PAIR_DTYPE); pairs["i"] = [1, 2, 3]
    assert ex.all_gather_object(pairs)[0].tobytes() == pairs.tobytes()     # the share of a dense pair stage
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    # and a whole job on the one-rank group (world 1 short-circuits the exchange, the group is live underneath)
    ctx = galah_amd.Context(0)
    job = DereplicationJob(ctx, 0, 1, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15)
    job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
    res = job.step()
    q.put({"ok": float(t.item()) == 1.25, "clusters": len(res["clusters"])})
    dist.destroy_process_group()
    ctx.close()


def test_rccl_entry_points_on_one_rank():
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_nccl_single_rank, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and got["ok"] and got["clusters"] > 0
