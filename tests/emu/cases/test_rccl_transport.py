"""EMULATION ONLY (collected when GALAH_TEST_EMU=1; tests/test_emu.py runs this file in a child process of the CPU suite).

The RCCL transport of galah_amd/csrc/comm.cpp with world > 1 -- which no one-GPU box can run (RCCL refuses two ranks on one
device) -- on the CPU: the library's own sources over the wave64 emulator (tests/emu), the ranks as THREADS of this process,
and tests/emu/fake_rccl standing in for librccl (a rendezvous that checks what real RCCL would hang on: counts that differ
between the ranks of an all-gather, a send that meets a receive of another size).  What runs is the library's code around the
collectives: buffer sizes and offsets of ncclAllGather / grouped ncclSend+ncclRecv, the status words, rccl_wait's poll of
ncclCommGetAsyncError and its deadline, ncclCommAbort and the dead-communicator state."""
import os
import threading
import time

import numpy as np
import pytest

import galah_amd
from galah_amd.distributed import Comm, DereplicationJob

SEED, MEMBERS, LENGTH, RATE, N = 5, 3, 120_000, 0.0253, 19


def _ranks(world, body, timeout_s=600):
    """body(rank, ctx, comm) on one thread per rank; returns the list of results (an exception becomes the result)."""
    uid = Comm.rccl_unique_id()
    out = [None] * world

    def run(r):
        ctx = galah_amd.Context(0)
        try:
            comm = Comm.from_rccl_id(ctx, r, world, uid)
            assert (comm.rank, comm.world, comm.transport) == (r, world, "rccl")
            try:
                out[r] = body(r, ctx, comm)
            finally:
                comm.close()
        except BaseException as e:  # noqa: BLE001
            out[r] = e
        finally:
            ctx.close()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout_s)
        assert not t.is_alive(), "a rank hangs"
    return out


def _single(lazy, order_seed=None):
    ctx = galah_amd.Context(0)
    job = DereplicationJob(ctx, 0, 1, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15, lazy_ani=lazy)
    if order_seed is not None:
        job.set_order(np.random.default_rng(order_seed).permutation(N))
    job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
    res = job.step()
    hashes, lens = job.sketches_to_host()
    out = {"pairs": res["pairs"].tobytes(), "clusters": res["clusters"].tolist() if lazy else res["clusters"],
           "ani": None if lazy else res["pair_ani"].tobytes(), "hashes": hashes.tobytes(), "lens": lens.tobytes()}
    job = None
    ctx.close()
    return out


def test_primitives_over_the_rccl_branches():
    world = 3

    def body(r, ctx, comm):
        got = comm.allgather_host(np.arange(5, dtype=np.int64) + 100 * r).tolist()        # a small payload: the pre-allocated path
        big = comm.allgather_host(np.full(3000, r, dtype=np.int64))                       # 24 kB: the pool-buffer path behind a status word
        comm.agree(0)
        try:
            comm.agree(3 if r == 1 else 0)
            agreed = "no error"
        except galah_amd.GalahHipError as e:
            agreed = str(e)
        return got, big[:, 0].tolist() + [int(big.sum())], agreed, comm.allgather_host(np.int64([r])).ravel().tolist()

    for r, o in enumerate(_ranks(world, body)):
        assert not isinstance(o, BaseException), o
        got, big, agreed, after = o
        assert got == [[100 * x + y for y in range(5)] for x in range(world)]
        assert big == [0, 1, 2, 3000 * 3]
        assert agreed.startswith("GHIP_EHIP" if r == 1 else "GHIP_EPEER"), agreed
        assert after == [0, 1, 2]


@pytest.mark.parametrize("world,lazy,order_seed,options", [(2, False, None, {}), (3, False, None, {"pair_form": "join"}),
                                                           (3, True, None, {}), (4, True, 17, {}), (3, True, None, {"join_ranks": "records", "pair_form": "join"})])
def test_whole_pass_over_rccl_equals_one_rank(world, lazy, order_seed, options):
    """sketch + seeds per rank, ncclAllGather of the sketch matrix, the pair stage (hash-sharded join with its two exchanges when
    forced; dense shares otherwise), the candidate gather, the ANI-index slices over grouped ncclSend / ncclRecv, then the eager
    shares or the lazy rounds: byte-identical to one rank."""
    want = _single(lazy, order_seed)

    def body(r, ctx, comm):
        ctx.set_options(**options)
        job = DereplicationJob(ctx, r, world, n_genomes=N, min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af=0.15, comm=comm, lazy_ani=lazy)
        if order_seed is not None:
            job.set_order(np.random.default_rng(order_seed).permutation(N))
        job.load_synthetic(SEED, MEMBERS, LENGTH, RATE)
        for _ in range(2):
            res = job.step()
        hashes, lens = job.sketches_to_host()
        out = {"hashes": hashes.tobytes(), "lens": lens.tobytes()}
        if lazy or r == 0:
            out.update(pairs=res["pairs"].tobytes(), clusters=res["clusters"].tolist() if lazy else res["clusters"],
                       ani=None if lazy else res["pair_ani"].tobytes())
        job = None
        return out

    outs = _ranks(world, body)
    for r, o in enumerate(outs):
        assert not isinstance(o, BaseException), (r, o)
        assert o["hashes"] == want["hashes"] and o["lens"] == want["lens"]
        if lazy or r == 0:
            assert o["pairs"] == want["pairs"] and o["clusters"] == want["clusters"] and o["ani"] == want["ani"], r


def test_a_peer_that_never_enters_the_collective_costs_a_deadline_not_a_hang():
    """Rank 2 stays out of an all-gather (it has died, say).  Ranks 0 and 1 wait comm_timeout_ms, abort their communicator and
    return GHIP_EPEER; every later collective on it fails at once.  (Without the deadline: hipStreamSynchronize for ever.)"""
    world = 3

    def body(r, ctx, comm):
        comm.allgather_host(np.int64([r]))          # the communicator works
        if r == 2:
            time.sleep(1.5)
            return "stayed out"
        ctx.set_options(comm_timeout_ms=300)
        t0 = time.time()
        try:
            comm.allgather_host(np.int64([r]))
            first = "no error"
        except galah_amd.GalahHipError as e:
            first = str(e)
        dt = time.time() - t0
        t0 = time.time()
        try:
            comm.allgather_host(np.int64([r]))
            second = "no error"
        except galah_amd.GalahHipError as e:
            second = str(e)
        return first, dt, second, time.time() - t0

    outs = _ranks(world, body)
    assert outs[2] == "stayed out"
    for o in outs[:2]:
        assert not isinstance(o, BaseException), o
        first, dt, second, dt2 = o
        assert first.startswith("GHIP_EPEER") and "comm_timeout_ms" in first and 0.25 < dt < 1.4, (first, dt)
        assert "aborted" in second and dt2 < 0.1, (second, dt2)


def test_no_deadline_by_default_and_a_late_healthy_peer_is_waited_for_at_a_phase_boundary():
    """the deadline counts from a rank's own entry, so it must not end a job whose peer is merely late.  The default
    is no deadline; with one set, the status-word exchange of a phase boundary (ghip_comm_agree: where uneven shards show)
    gets ten times the value, a data collective behind it the value itself."""
    world = 2
    assert galah_amd.get_options()["comm_timeout_ms"] == 0

    def body(r, ctx, comm):
        assert ctx.options()["comm_timeout_ms"] == 0
        ctx.set_options(comm_timeout_ms=150)
        comm.allgather_host(np.int64([r]))
        if r == 1:
            time.sleep(0.6)                    # four deadlines late for the boundary: healthy, just slow
        t0 = time.time()
        comm.agree(0)
        dt = time.time() - t0
        if r == 1:
            time.sleep(0.6)                    # ... and as late for a data collective: that one is not waited for
        try:
            comm.allgather_host(np.int64([r]))
            second = "no error"
        except galah_amd.GalahHipError as e:
            second = str(e)
        return dt, second

    outs = _ranks(world, body)
    for o in outs:
        assert not isinstance(o, BaseException), o
    assert outs[0][0] > 0.4, outs            # rank 0 waited for its peer at the boundary, past comm_timeout_ms
    assert outs[0][1].startswith("GHIP_EPEER") and "comm_timeout_ms)" in outs[0][1], outs[0]


def test_rccl_reports_the_remote_failure_before_the_deadline():
    """With RCCL itself noticing the dead peer (ncclCommGetAsyncError -> ncclRemoteError), the wait ends long before a 20 s
    deadline."""
    world = 2
    os.environ["FAKE_RCCL_ABORT_SEEN_BY_PEERS"] = "1"
    try:
        def body(r, ctx, comm):
            comm.allgather_host(np.int64([r]))
            if r == 1:
                ctx.set_options(comm_timeout_ms=50)   # rank 1 gives up on a collective rank 0 is late for, and aborts
                try:
                    comm.allgather_host(np.int64([r]))
                    return "no error"
                except galah_amd.GalahHipError as e:
                    return str(e)
            time.sleep(0.6)
            ctx.set_options(comm_timeout_ms=20000)
            t0 = time.time()
            try:
                comm.allgather_host(np.int64([r]))
                return "no error", 0.0
            except galah_amd.GalahHipError as e:
                return str(e), time.time() - t0

        outs = _ranks(world, body)
        assert isinstance(outs[1], str) and outs[1].startswith("GHIP_EPEER"), outs[1]
        msg, dt = outs[0]
        assert msg.startswith("GHIP_EPEER") and "RCCL reports" in msg and dt < 2.0, (msg, dt)
    finally:
        del os.environ["FAKE_RCCL_ABORT_SEEN_BY_PEERS"]
