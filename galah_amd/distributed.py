"""Multi-GPU orchestration of one dereplication job -- a thin caller of the C ABI's exchange entry points
(include/galah_hip.h "multi-GPU exchange", galah_amd/csrc/comm.cpp); nothing here moves or computes data.

  rank r owns genomes [r*B, min((r+1)*B, N)),  B = ceil(N / world)           (contiguous blocks)
  1. sketch the local genomes (MinHash + ANI seed index, one fused pass)      (no communication)
  2. ALL-GATHER the packed sketch matrix          -> u64[N][s] on every rank (N*s*8 bytes, once)
  3. pair work of the gathered matrix dealt over the ranks; the shares are gathered and merged by (i, j)
  4. a pair's ANI is computed where its first genome lives; only the index slices a rank needs but does not own move
  5. ANI values gathered (consecutive runs of the sorted list); greedy clustering on rank 0's host (src/clusterer.rs)

Steps 1-5 are ONE call, ghip_distances_and_ani_ranks, on a communicator made here:
  "rccl"  one process per GPU, ncclAllGather over xGMI issued by the library itself (the 128-byte unique id travels
          through torch.distributed's CPU side);
  "gloo"  the library's host-callback transport fed by torch.distributed all_gather on CPU tensors (ranks sharing one
          GPU in tests; the functional fallback);
  "local" one process driving several contexts from threads (peer copies) -- see cluster_files_multi.
There is no CPU compute path in this package.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import PAIR_DTYPE, GalahHipError, check


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(first, count, block) of rank's contiguous genome block (ghip_shard_range)."""
    f, c, b = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    _lib.lib().ghip_shard_range(n, rank, world, C.byref(f), C.byref(c), C.byref(b))
    return int(f.value), int(c.value), int(b.value)


def tile_pairs_of_rank(n: int, pt: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Host restatement of the dense pair kernels' block-cyclic tile deal (pairs.hip): upper-triangle tile
    pairs in row-major order, tile t goes to rank t % world."""
    nt = (n + pt - 1) // pt
    out, t = [], 0
    for ti in range(nt):
        for tj in range(ti, nt):
            if t % world == rank:
                out.append((ti, tj))
            t += 1
    return out


class Comm:
    """One rank's communicator (ghip_comm).  ctx may be None for a host-payload-only callback communicator."""

    def __init__(self, handle, ctx, keep=None):
        self._h = handle
        self.ctx = ctx
        self._keep = keep   # the ctypes callback object must outlive the communicator

    # ---- constructors
    @classmethod
    def single(cls, ctx) -> "Comm":
        h = C.c_void_p()
        check(_lib.lib().ghip_comm_init_callback(ctx._h if ctx is not None else None, 0, 1, None, None, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_torch_gloo(cls, ctx, rank: int, world: int, group=None) -> "Comm":
        """Host-callback transport: the all-gather of host bytes is torch.distributed's, on CPU tensors."""
        import torch
        import torch.distributed as dist
        failure = []

        def _allgather(_user, send, nbytes, recv):
            try:
                s = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8)
                r = torch.frombuffer((C.c_uint8 * (nbytes * world)).from_address(recv), dtype=torch.uint8)
                dist.all_gather_into_tensor(r, s, group=group)
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not escape through ctypes
                failure.append(e)
                return 1

        cb = _lib.ALLGATHER_FN(_allgather)
        h = C.c_void_p()
        check(_lib.lib().ghip_comm_init_callback(ctx._h if ctx is not None else None, rank, world, C.cast(cb, C.c_void_p), None, C.byref(h)))
        c = cls(h, ctx, keep=(cb, failure))
        return c

    @classmethod
    def from_rccl_id(cls, ctx, rank: int, world: int, uid: bytes) -> "Comm":
        """RCCL transport for a host that ships the 128-byte unique id itself (rank 0: Comm.rccl_unique_id()): the library calls
        ncclCommInitRank; blocks until every rank has joined."""
        h = C.c_void_p()
        check(_lib.lib().ghip_comm_init_rank(ctx._h, rank, world, (C.c_uint8 * 128).from_buffer_copy(uid), C.byref(h)), ctx._h)
        return cls(h, ctx)

    @staticmethod
    def rccl_unique_id() -> bytes:
        uid = (C.c_uint8 * 128)()
        check(_lib.lib().ghip_comm_unique_id(uid))
        return bytes(uid)

    @classmethod
    def from_torch_rccl(cls, ctx, rank: int, world: int, timeout_s: float = 180.0) -> "Comm":
        """RCCL transport: rank 0 makes the unique id, torch.distributed (any backend) ships its 128 bytes, the
        library calls ncclCommInitRank itself.  The init runs under a watchdog; raises on failure or timeout."""
        import torch.distributed as dist
        L = _lib.lib()
        uid = (C.c_uint8 * 128)()
        box = [None]
        if rank == 0:
            rc = L.ghip_comm_unique_id(uid)
            box[0] = bytes(uid) if rc == 0 else None
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise GalahHipError(5, "rank 0 could not create an RCCL unique id: " + (L.ghip_last_error(None) or b"").decode())
        uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        result = {}

        def _init():
            result["rc"] = L.ghip_comm_init_rank(ctx._h, rank, world, uid, C.byref(h))
            if result["rc"] == 0:
                # a first, tiny collective under the same watchdog: a communicator that initialises but whose
                # all-gather fails or hangs (no usable link between two ranks) is found out here, where the caller
                # can still fall back to the host transport on every rank
                probe = np.array([rank * 7 + 1], dtype=np.int64)
                got = np.empty((world, 1), dtype=np.int64)
                result["probe_rc"] = L.ghip_comm_allgather_host(h, probe.ctypes.data, probe.nbytes, got.ctypes.data)
                result["probe_ok"] = result["probe_rc"] == 0 and got[:, 0].tolist() == [r * 7 + 1 for r in range(world)]

        t = threading.Thread(target=_init, daemon=True)
        t.start()
        t.join(timeout_s)
        if t.is_alive():
            raise TimeoutError(f"RCCL communicator set-up (ncclCommInitRank + a first all-gather) did not finish within {timeout_s:.0f} s")
        check(result["rc"], ctx._h)
        if not result.get("probe_ok"):
            raise GalahHipError(5, "the first RCCL all-gather failed: " + L.ghip_comm_last_error(h).decode())
        return cls(h, ctx)

    # ---- queries
    @property
    def rank(self) -> int:
        return int(_lib.lib().ghip_comm_rank(self._h))

    @property
    def world(self) -> int:
        return int(_lib.lib().ghip_comm_world(self._h))

    @property
    def transport(self) -> str:
        return _lib.lib().ghip_comm_transport(self._h).decode()

    def _check(self, rc: int):
        if rc != 0:
            if self._keep and self._keep[1]:
                raise self._keep[1][0]
            raise GalahHipError(rc, _lib.lib().ghip_comm_last_error(self._h).decode())

    # ---- collectives
    def allgather_host(self, arr: np.ndarray) -> np.ndarray:
        """Same-shaped host arrays -> stacked [world, ...]."""
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        self._check(_lib.lib().ghip_comm_allgather_host(self._h, a.ctypes.data, a.nbytes, out.ctypes.data))
        return out

    def allgather_device(self, d_send: int, d_recv: int, nbytes: int):
        self._check(_lib.lib().ghip_comm_allgather_device(self._h, C.c_void_p(d_send), C.c_void_p(d_recv), nbytes))

    def exchange_device(self, d_send: int, send_off, d_recv: int, recv_off):
        """All-to-all-v of device bytes (ghip_comm_exchange_device): byte offsets, world + 1 each."""
        so = np.ascontiguousarray(send_off, dtype=np.uint64)
        ro = np.ascontiguousarray(recv_off, dtype=np.uint64)
        assert so.size == self.world + 1 and ro.size == self.world + 1
        self._check(_lib.lib().ghip_comm_exchange_device(self._h, C.c_void_p(d_send), so.ctypes.data_as(C.c_void_p), C.c_void_p(d_recv),
                                                         ro.ctypes.data_as(C.c_void_p)))

    def allgather_pairs(self, pairs: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        out, n = C.c_void_p(), C.c_size_t(0)
        self._check(_lib.lib().ghip_allgather_pairs(self._h, p.ctypes.data, p.shape[0], C.byref(out), C.byref(n)))
        return _take(out, n.value, PAIR_DTYPE)

    def allgather_sketches(self, local, n_total: int):
        from .engine import Sketches
        h = C.c_void_p()
        self._check(_lib.lib().ghip_allgather_sketches(self._h, local._h, n_total, C.byref(h)))
        return Sketches(self.ctx, h)

    def exchange_ani_index(self, local, n_total: int, pairs: np.ndarray):
        """-> (index serving this rank's pairs, position of every global genome in it (uint32, 0xffffffff = absent))."""
        from .engine import AniIndex
        p = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        ids = np.empty(max(n_total, 1), dtype=np.uint32)
        h = C.c_void_p()
        self._check(_lib.lib().ghip_exchange_ani_index(self._h, local._h, n_total, p.ctypes.data, p.shape[0], C.byref(h), ids.ctypes.data))
        if h.value == local._h.value:
            return local, ids[:n_total]
        return AniIndex(self.ctx, h), ids[:n_total]

    def distances_and_ani(self, genomes, n_total: int, kmer: int, sketch_size: int, min_ani, ani_k: int, ani_c: int,
                          ani_chunk: int, min_af: float, want_sketches: bool = False):
        """ghip_distances_and_ani_ranks -> (pairs, ani, stage milliseconds, gathered Sketches or None)."""
        from .engine import Sketches
        pp, pa, n = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        sk = C.c_void_p()
        tm = _lib.RankTimes()
        self._check(_lib.lib().ghip_distances_and_ani_ranks(self._h, genomes._h, n_total, kmer, sketch_size, 0, np.float32(min_ani),
                                                            ani_k, ani_c, ani_chunk, np.float32(min_af), C.byref(pp), C.byref(pa),
                                                            C.byref(n), C.byref(sk) if want_sketches else None, C.byref(tm)))
        pairs = _take(pp, n.value, PAIR_DTYPE)
        ani = _take(pa, n.value, np.dtype(np.float32))
        times = {k[:-3]: float(getattr(tm, k)) for k, _ in _lib.RankTimes._fields_ if k.endswith("_ms")}
        times["_pairs_compared"] = int(tm.pairs_compared)
        return pairs, ani, times, (Sketches(self.ctx, sk) if want_sketches else None)

    def cluster_ranks(self, genomes, n_total: int, kmer: int, sketch_size: int, min_ani, ani_k: int, ani_c: int, ani_chunk: int,
                      min_af: float, ani_threshold, order=None, want_pairs: bool = True, want_sketches: bool = False):
        """ghip_cluster_ranks: the whole pass with the LAZY ANI rounds of the native clusterer dealt over the ranks -- the
        algorithm one rank runs, for every world size.  Every rank gets the same result.
        -> (clusters, pairs or None, stage milliseconds + counters, gathered Sketches or None)."""
        from .engine import ClusterList, Sketches
        L = _lib.lib()
        members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        pp, npairs, sk = C.c_void_p(), C.c_size_t(0), C.c_void_p()
        tm = _lib.ClusterTimes()
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
            assert order.shape == (n_total,)
        self._check(L.ghip_cluster_ranks(self._h, genomes._h, n_total, kmer, sketch_size, 0, np.float32(min_ani), ani_k, ani_c, ani_chunk,
                                         np.float32(min_af), order.ctypes.data if order is not None else None, np.float32(ani_threshold),
                                         C.byref(members), C.byref(offsets), C.byref(nc), C.byref(pp) if want_pairs else None,
                                         C.byref(npairs), C.byref(sk) if want_sketches else None, C.byref(tm)))
        try:
            off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
            mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()
        finally:
            L.ghip_free(members)
            L.ghip_free(offsets)
        pairs = _take(pp, npairs.value, PAIR_DTYPE) if want_pairs else None
        times = {k: (float(getattr(tm, k)) if k.endswith("_ms") else int(getattr(tm, k))) for k, _ in _lib.ClusterTimes._fields_}
        times["n_pairs"] = int(npairs.value)
        return ClusterList(mem, off), pairs, times, (Sketches(self.ctx, sk) if want_sketches else None)

    def cluster_index(self, idx, local_ids, n_total: int, pairs: np.ndarray, ani_threshold, min_af: float = 0.15, order=None):
        """ghip_cluster_index_comm (every rank calls it with the same pair list): -> (clusters, stats)."""
        from .engine import ClusterList
        L = _lib.lib()
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        ids = None if local_ids is None else np.ascontiguousarray(local_ids, dtype=np.uint32)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
        members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        stats = np.zeros(5, dtype=np.uint64)
        self._check(L.ghip_cluster_index_comm(self._h, idx._h if idx is not None else None, ids.ctypes.data if ids is not None else None, n_total,
                                              pairs.ctypes.data, pairs.shape[0], order.ctypes.data if order is not None else None,
                                              np.float32(ani_threshold), np.float32(min_af), C.byref(members), C.byref(offsets), C.byref(nc),
                                              stats.ctypes.data))
        try:
            off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
            mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()
        finally:
            L.ghip_free(members)
            L.ghip_free(offsets)
        return ClusterList(mem, off), {"asked": int(stats[0]), "rounds": int(stats[1]), "ani_ms": float(stats[2]) * 1e-6,
                                       "total_ms": float(stats[3]) * 1e-6, "asked_here": int(stats[4])}

    def cluster_lazy(self, n_total: int, pairs: np.ndarray, ani_threshold, ani_of_edges, order=None):
        """ghip_cluster_lazy_comm: the lazy rounds answered by the host's own batched ANI on every rank -- ani_of_edges(edge
        indices into `pairs`) -> percent values, called with the edges THIS rank answers.  Every rank calls it with the same
        pair list.  Needs no device.  -> (clusters, stats)."""
        from .engine import ClusterList
        L = _lib.lib()
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
        failure = []

        def _cb(_user, edges, n, out):
            try:
                idx = np.ctypeslib.as_array(edges, shape=(n,)).copy()
                vals = np.ascontiguousarray(ani_of_edges(idx), dtype=np.float32)
                assert vals.shape == (n,)
                C.memmove(out, vals.ctypes.data, 4 * n)
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not escape through ctypes
                failure.append(e)
                return 1

        cb = _lib.ANI_BATCH_CALLBACK(_cb)
        members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        stats = np.zeros(5, dtype=np.uint64)
        rc = L.ghip_cluster_lazy_comm(self._h, n_total, pairs.ctypes.data, pairs.shape[0], order.ctypes.data if order is not None else None,
                                      np.float32(ani_threshold), cb, None, C.byref(members), C.byref(offsets), C.byref(nc), stats.ctypes.data)
        if rc != 0 and failure:
            raise failure[0]
        self._check(rc)
        try:
            off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
            mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()
        finally:
            L.ghip_free(members)
            L.ghip_free(offsets)
        return ClusterList(mem, off), {"asked": int(stats[0]), "rounds": int(stats[1]), "ani_ms": float(stats[2]) * 1e-6,
                                       "total_ms": float(stats[3]) * 1e-6, "asked_here": int(stats[4])}

    def agree(self, status: int = 0):
        """ghip_comm_agree: raises on every rank when any rank passes a non-zero status."""
        self._check(_lib.lib().ghip_comm_agree(self._h, int(status)))

    def close(self):
        if self._h:
            _lib.lib().ghip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _take(p, n: int, dtype: np.dtype) -> np.ndarray:
    try:
        if n == 0 or not p.value:
            return np.empty(0, dtype=dtype)
        buf = (C.c_char * (n * dtype.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).copy()
    finally:
        _lib.lib().ghip_free(p)


def local_comms(contexts: Sequence) -> List[Comm]:
    """One process, several contexts (one thread each): peer-copy transport."""
    world = len(contexts)
    arr = (C.c_void_p * world)(*[c._h for c in contexts])
    out = (C.c_void_p * world)()
    check(_lib.lib().ghip_comm_init_local(arr, world, out))
    return [Comm(C.c_void_p(out[r]), contexts[r]) for r in range(world)]


def cluster_files_multi(contexts: Sequence, paths: Sequence[str], min_ani: float = 0.9, ani_threshold: float = 95.0,
                        min_aligned_fraction: float = 0.15, kmer: int = 21, sketch_size: int = 1000, ani_c: int = 125,
                        io_threads: int = 8) -> List[List[int]]:
    """Files in -> clusters out on len(contexts) GPUs driven by THIS process (ghip_cluster_files_multi)."""
    L = _lib.lib()
    world = len(contexts)
    arr = (C.c_void_p * world)(*[c._h for c in contexts])
    cp = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    check(L.ghip_cluster_files_multi(arr, world, cp, len(paths), kmer, sketch_size, np.float32(min_ani), np.float32(ani_threshold),
                                     np.float32(min_aligned_fraction), ani_c, io_threads, C.byref(members), C.byref(offsets),
                                     C.byref(nc)), contexts[0]._h)
    try:
        off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
        mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()
    finally:
        L.ghip_free(members)
        L.ghip_free(offsets)
    mem_l, off_l = mem.tolist(), off.tolist()
    return [mem_l[off_l[c]:off_l[c + 1]] for c in range(nc.value)]


def make_comm(ctx, rank: int, world: int, backend: str = "rccl") -> Comm:
    """The communicator a torch.distributed-launched rank uses.  backend "rccl": the library's own RCCL communicator,
    with the host-callback transport over torch.distributed as the fallback if RCCL cannot be initialised on EVERY
    rank; "gloo": the host-callback transport outright."""
    if world == 1:
        return Comm.single(ctx)
    import torch
    import torch.distributed as dist
    if backend == "rccl":
        comm, ok = None, 1
        try:
            comm = Comm.from_torch_rccl(ctx, rank, world)
        except Exception as e:  # noqa: BLE001
            import sys
            print(f"[galah_amd] rank {rank}: RCCL communicator failed ({e!r}); falling back to the host transport", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # CPU tensor: needs a gloo-capable process group
        if int(flag.item()) == 1:
            return comm
        comm = None  # a half-initialised RCCL communicator is abandoned, not destroyed (its peers may be stuck)
    return Comm.from_torch_gloo(ctx, rank, world)


class DereplicationJob:
    """One rank of a dereplication job over genomes resident in HBM."""

    def __init__(self, ctx, rank: int, world: int, n_genomes: int, kmer: int = 21, sketch_size: int = 1000,
                 min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af: float = 0.15, comm: Optional[Comm] = None,
                 backend: str = "rccl", ani_k: int = 15, ani_c: int = 125, ani_chunk: int = 20000, lazy_ani: bool = False):
        self.ctx, self.rank, self.world, self.n = ctx, rank, world, n_genomes
        self.kmer, self.s = kmer, sketch_size
        self.ani_k, self.ani_c, self.ani_chunk = ani_k, ani_c, ani_chunk
        self.min_ani, self.ani_threshold, self.min_af = np.float32(min_ani), np.float32(ani_threshold), float(min_af)
        self.comm = comm if comm is not None else make_comm(ctx, rank, world, backend)
        self.first, self.count, self.block = shard_range(n_genomes, rank, world)
        # lazy_ani: the clusterer's ANI is asked for lazily, in batches, only for the pairs the greedy rules look at -- what the
        # reference does one `skani dist` at a time (src/clusterer.rs:194-204, 377-405) -- by the native clusterer, for EVERY
        # world size (ghip_cluster_ranks: a round's requests are dealt to the ranks).  Otherwise every precluster pair's ANI is
        # computed in parallel shares (ghip_distances_and_ani_ranks) and rank 0 clusters from the values.
        self.lazy_ani = lazy_ani
        self.last_pairs_asked = 0
        self.genomes = None
        self._stage: Dict[str, float] = {}
        self._steps = 0
        self._full = None
        self.last_pairs_compared = 0
        self.order = None

    def set_order(self, order):
        """Genome order of the clustering stage (one rank): order[x] = the genome that comes x-th -- galah sorts its input by
        quality before clustering (src/cluster_argument_parsing.rs:863-1157), so the best genome of a cluster becomes its
        representative.  Sketches, pairs and ANI are computed where the genomes lie; the pair list is re-indexed and re-sorted
        before the greedy clusterer, whose clusters then hold positions in `order`."""
        order = np.asarray(order, dtype=np.int64)
        assert self.lazy_ani and sorted(order.tolist()) == list(range(self.n))
        self.order = order.astype(np.uint32)

    def load_synthetic(self, seed: int, members: int, length: int, sub_rate: float):
        self.genomes = self.ctx.genomes_synthetic_range(seed, members, self.first, self.count, length, sub_rate)

    def load_files(self, paths, io_threads: int = 8):
        assert len(paths) == self.n
        self.genomes = self.ctx.genomes_from_files(list(paths[self.first: self.first + self.count]), io_threads)

    @property
    def local_bases(self) -> int:
        return self.genomes.total_bases

    def reset_stage_timers(self):
        self._stage, self._steps = {}, 0

    def stage_ms(self) -> Dict[str, float]:
        return {k: v / max(self._steps, 1) for k, v in self._stage.items()}

    def sketches_to_host(self):
        """(u64[N][s], u32[N]) of the gathered matrix of the last step."""
        return self._full.to_host()

    def _step_lazy(self) -> Dict:
        # clusterer::cluster in native code (ghip_cluster_ranks): sketch + seed, (gather), pair stage, (gather, index slices),
        # then the lazy rounds answered by the resident ANI index on the rank that owns each pair; with a quality order the
        # clusterer sees the genomes in that order and the clusters hold positions in it.  Every rank returns the same clusters.
        clusters, pairs, tm, self._full = self.comm.cluster_ranks(self.genomes, self.n, self.kmer, self.s, self.min_ani, self.ani_k, self.ani_c,
                                                                 self.ani_chunk, self.min_af, self.ani_threshold, self.order,
                                                                 want_pairs=True, want_sketches=True)
        self.last_pairs_compared = tm["pairs_compared"]
        self.last_pairs_asked = tm["ani_pairs_asked"]
        for k, v in (("sketch", tm["sketch_ms"]), ("allgather_sketches", tm["allgather_sketches_ms"]), ("pairs", tm["pairs_ms"]),
                     ("allgather_pairs", tm["allgather_pairs_ms"]), ("exchange_ani_index", tm["exchange_ani_index_ms"]),
                     ("ani_pairs", tm["ani_rounds_ms"]), ("host_cluster", tm["cluster_host_ms"])):
            if self.world > 1 or k in ("sketch", "pairs", "ani_pairs", "host_cluster"):
                self._stage[k] = self._stage.get(k, 0.0) + v
        self._steps += 1
        return {"n_pairs": len(pairs), "n_clusters": len(clusters), "clusters": clusters, "pairs": pairs, "pair_ani": None,
                "ani_pairs_asked": tm["ani_pairs_asked"], "ani_pairs_here": tm["ani_pairs_here"], "lazy_rounds": tm["lazy_rounds"]}

    def step(self) -> Dict:
        if self._full is not None:
            self._full.free()
            self._full = None
        if self.lazy_ani:
            return self._step_lazy()
        pairs, ani, times, self._full = self.comm.distances_and_ani(self.genomes, self.n, self.kmer, self.s, self.min_ani, self.ani_k,
                                                                    self.ani_c, self.ani_chunk, self.min_af, want_sketches=True)
        self.last_pairs_compared = times.pop("_pairs_compared")
        for k, v in times.items():
            self._stage[k] = self._stage.get(k, 0.0) + v
        result = {"n_pairs": 0, "n_clusters": 0, "clusters": None, "pairs": None, "pair_ani": None}
        if self.rank == 0:
            from .engine import cluster_pairs
            t0 = time.perf_counter()
            clusters = cluster_pairs(self.n, pairs, self.ani_threshold, ani, False)
            self._stage["host_cluster"] = self._stage.get("host_cluster", 0.0) + (time.perf_counter() - t0) * 1e3
            result = {"n_pairs": len(pairs), "n_clusters": len(clusters), "clusters": clusters, "pairs": pairs, "pair_ani": ani}
        self._steps += 1
        return result
