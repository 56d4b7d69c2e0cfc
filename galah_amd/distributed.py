"""Multi-GPU orchestration of one dereplication job: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) for the two real exchange steps of the path.

  rank r owns genomes [r*B, min((r+1)*B, N)),  B = ceil(N / world)           (contiguous blocks)
  1. sketch the local genomes                     -> u64[B][s] + u32[B]      (no communication)
  2. ALL-GATHER the packed sketch matrix          -> u64[N][s] on every rank (N*s*8 bytes, once)
  3. pair tiles t with t % world == rank          -> local candidate list    (no communication)
  4. all-gather the (small) candidate lists; a pair is computed on the rank that owns its first
     genome, so only the ANI seed indexes of genomes needed by a rank that does not own them are
     exchanged (ALL-GATHER of the packed slices; ~0.33 MB per needed genome)
  5. ANI of the local share; gather the values on rank 0; greedy clustering on the host
                                                                             (src/clusterer.rs)

With world == 1 the same code runs with the exchange steps skipped.  The compute engine is
pluggable so the exchange logic can be exercised on CPU (gloo) in tests; the product engine is
HipEngine (libgalah_hip.so) -- there is no CPU compute path in this package.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from ._lib import PAIR_DTYPE


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(first, count, block) of rank's contiguous genome block."""
    block = (n + world - 1) // world
    first = min(rank * block, n)
    return first, min(block, n - first), block


def tile_pairs_of_rank(n: int, pt: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Host restatement of the kernel's block-cyclic tile deal (pairs.hip): upper-triangle tile
    pairs in row-major order, tile t goes to rank t % world."""
    nt = (n + pt - 1) // pt
    out, t = [], 0
    for ti in range(nt):
        for tj in range(ti, nt):
            if t % world == rank:
                out.append((ti, tj))
            t += 1
    return out


class Exchange:
    """torch.distributed plumbing; tensors live wherever the engine puts them (HBM or host)."""

    def __init__(self, rank: int, world: int, force: bool = False):
        self.force = force
        self.rank, self.world = rank, world
        self.stage_on_host = False
        # force=True (tests): issue the collectives even on a one-rank group instead of short-circuiting them
        if world > 1 or force:
            import torch.distributed as dist
            self.dist = dist
            # gloo (CPU tests, or two test ranks sharing one GPU) moves device tensors through the host;
            # nccl (= RCCL over xGMI, the production path) gathers straight out of HBM.
            self.stage_on_host = dist.get_backend() == "gloo"

    def all_gather_host_array(self, arr: np.ndarray, sizes: List[int]) -> np.ndarray:
        """Host arrays of per-rank length sizes[r] (known to every rank) -> their concatenation, one collective."""
        import torch
        if self.world == 1 and not self.force:
            return arr
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if not self.stage_on_host:  # nccl moves device tensors only
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        return self.all_gather_flat(t, sizes).cpu().numpy()

    def _gather(self, out, inp):
        if self.stage_on_host and inp.is_cuda:
            o = out.cpu()
            self.dist.all_gather_into_tensor(o, inp.cpu().contiguous())
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp.contiguous())

    def all_gather_blocks(self, local, n_total: int):
        """local: [block, ...] tensor (same block on every rank) -> [n_total, ...]."""
        import torch
        if self.world == 1 and not self.force:
            return local[:n_total]
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self._gather(out, local)
        return out[:n_total]

    def all_gather_flat(self, local, sizes: List[int]):
        """1-D tensors of per-rank length sizes[r] -> their concatenation on every rank.
        Moved as raw bytes (u8), so any element type works with any backend."""
        import torch
        if self.world == 1 and not self.force:
            return local
        if max(sizes) == 0:  # nothing to exchange (e.g. no candidate pair spans two ranks)
            return local[:0]
        esz = local.element_size()
        raw = local.contiguous().view(torch.uint8)
        m = max(sizes) * esz
        m = (m + 15) // 16 * 16
        padded = torch.zeros(m, dtype=torch.uint8, device=local.device)
        padded[: raw.shape[0]] = raw
        out = torch.empty(self.world * m, dtype=torch.uint8, device=local.device)
        self._gather(out, padded)
        if all(sz * esz == m for sz in sizes):
            return out.view(local.dtype)
        return torch.cat([out[r * m: r * m + sizes[r] * esz] for r in range(self.world)]).view(local.dtype)

    def all_gather_object(self, obj):
        if self.world == 1 and not self.force:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class HipEngine:
    """The product engine: every method is a thin call into libgalah_hip.so."""

    def __init__(self, ctx, kmer: int, sketch_size: int, ani_k: int = 15, ani_c: int = 125, ani_chunk: int = 20000):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device)
        self.kmer, self.s = kmer, sketch_size
        self.ani_k, self.ani_c, self.ani_chunk = ani_k, ani_c, ani_chunk
        self.genomes = None
        self._keep = []

    # ---- inputs
    def load_synthetic(self, seed, members, first, count, length, sub_rate):
        self.genomes = self.ctx.genomes_synthetic_range(seed, members, first, count, length, sub_rate)

    def load_files(self, paths, io_threads=8):
        self.genomes = self.ctx.genomes_from_files(paths, io_threads)

    @property
    def local_bases(self) -> int:
        return self.genomes.total_bases

    # ---- stage 1/2
    def sketch_local(self, block: int):
        """-> (int64[block][s], int32[block]) device tensors, rows past the local count padded."""
        t = self.torch
        # one pass over the bases yields the MinHash sketches and the ANI seed index
        sk, self._fused_index = self.ctx.sketch_and_index(self.genomes, self.kmer, self.s, 0, self.ani_k, self.ani_c,
                                                          self.ani_chunk)
        hashes = t.full((block, self.s), -1, dtype=t.int64, device=self.device)  # -1 == 2^64-1
        lens = t.zeros(block, dtype=t.int32, device=self.device)
        t.cuda.current_stream().synchronize()
        self.ctx.sketches_copy_into(sk, hashes.data_ptr(), lens.data_ptr())
        self.ctx.synchronize()
        sk.free()
        return hashes, lens

    def precluster(self, hashes, lens, n: int, min_ani, rank: int, world: int):
        """-> (pairs sorted by (i, j), replicated).  replicated: every rank holds the whole list (the join form ran on
        the full matrix everywhere), nothing to exchange; otherwise `pairs` is this rank's share."""
        self.torch.cuda.current_stream().synchronize()
        sk = self.ctx.sketches_wrap_device(hashes.data_ptr(), lens.data_ptr(), n, self.s, self.kmer)
        pairs, replicated = self.ctx.precluster_ranks(sk, min_ani, rank, world)
        self.last_pairs_compared = self.ctx.last_pairs_compared
        sk.free()
        return pairs, replicated

    def sketches_to_host(self, hashes, lens):
        return hashes.cpu().numpy().view(np.uint64), lens.cpu().numpy().view(np.uint32)

    # ---- stage 4
    def ani_build_local(self):
        """-> (meta dict of host arrays, dict of flat device tensors)."""
        t = self.torch
        idx = getattr(self, "_fused_index", None)
        self._fused_index = None
        if idx is None:
            idx = self.ctx.ani_index_build(self.genomes, self.ani_k, self.ani_c, self.ani_chunk)
        glen, cap, cnt = idx.meta()
        lay = idx.layout()
        self._local_index = idx
        return idx, {"glen": glen, "cap": cap, "cnt": cnt}, lay

    def ani_export(self, idx, lay):
        t = self.torch
        arrs = {
            "seed_code": t.empty(int(lay.n_seed_slots), dtype=t.int32, device=self.device),
            "seed_chunk": t.empty(int(lay.n_seed_slots), dtype=t.int16, device=self.device),
            "bin_start": t.empty(int(lay.n_bin_slots), dtype=t.int32, device=self.device),
            "chunk_total": t.empty(int(lay.n_chunk_slots), dtype=t.int32, device=self.device),
        }
        t.cuda.current_stream().synchronize()
        self.ctx.memcpy_d2d(arrs["seed_code"].data_ptr(), lay.d_seed_code, int(lay.n_seed_slots) * 4)
        self.ctx.memcpy_d2d(arrs["seed_chunk"].data_ptr(), lay.d_seed_chunk, int(lay.n_seed_slots) * 2)
        self.ctx.memcpy_d2d(arrs["bin_start"].data_ptr(), lay.d_bin_start, int(lay.n_bin_slots) * 4)
        self.ctx.memcpy_d2d(arrs["chunk_total"].data_ptr(), lay.d_chunk_total, int(lay.n_chunk_slots) * 4)
        self.ctx.synchronize()
        return arrs

    def ani_wrap(self, meta, arrs):
        self.torch.cuda.current_stream().synchronize()
        self._keep = [arrs]  # the wrapped index borrows these tensors
        return self.ctx.ani_index_wrap_device(self.ani_k, self.ani_c, self.ani_chunk, meta["glen"], meta["cap"],
                                              meta["cnt"], arrs["seed_code"].data_ptr(),
                                              arrs["seed_chunk"].data_ptr(), arrs["bin_start"].data_ptr(),
                                              arrs["chunk_total"].data_ptr())

    def ani_pairs(self, idx, pairs: np.ndarray, min_af: float) -> np.ndarray:
        if len(pairs) == 0:
            return np.zeros(0, dtype=np.float32)
        pi = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32)
        return self.ctx.ani_pairs(idx, pi, min_af)

    def cluster(self, n, pairs, pair_ani, ani_threshold):
        from .engine import cluster_pairs
        return cluster_pairs(n, pairs, ani_threshold, pair_ani, False)


class DereplicationJob:
    def __init__(self, ctx, rank: int, world: int, n_genomes: int, kmer: int = 21, sketch_size: int = 1000,
                 min_ani=np.float32(0.9), ani_threshold=np.float32(95.0), min_af: float = 0.15, engine=None):
        self.rank, self.world, self.n = rank, world, n_genomes
        self.min_ani, self.ani_threshold, self.min_af = np.float32(min_ani), np.float32(ani_threshold), float(min_af)
        self.engine = engine if engine is not None else HipEngine(ctx, kmer, sketch_size)
        self.ex = Exchange(rank, world)
        self.first, self.count, self.block = shard_range(n_genomes, rank, world)
        self._stage = {}
        self._steps = 0
        self._full = None

    def load_synthetic(self, seed: int, members: int, length: int, sub_rate: float):
        self.engine.load_synthetic(seed, members, self.first, self.count, length, sub_rate)

    def load_files(self, paths, io_threads: int = 8):
        assert len(paths) == self.n
        self.engine.load_files(list(paths[self.first: self.first + self.count]), io_threads)

    @property
    def local_bases(self) -> int:
        return self.engine.local_bases

    @property
    def last_pairs_compared(self) -> int:
        return getattr(self.engine, "last_pairs_compared", 0)

    def _tick(self, name: str, t0: float) -> float:
        t1 = time.perf_counter()
        self._stage[name] = self._stage.get(name, 0.0) + (t1 - t0)
        return t1

    def reset_stage_timers(self):
        self._stage, self._steps = {}, 0

    def stage_ms(self) -> Dict[str, float]:
        return {k: v / max(self._steps, 1) * 1e3 for k, v in self._stage.items()}

    def sketches_to_host(self):
        return self.engine.sketches_to_host(*self._full)

    def _pack_genomes(self, arrs, meta, ids):
        """Flat-array slices of the local genomes `ids` (ascending), packed back to back."""
        import torch
        cap = meta["cap"].astype(np.int64)
        nch = (meta["glen"].astype(np.int64) + self._chunk() - 1) // self._chunk()
        s0 = np.concatenate([[0], np.cumsum(cap)])
        c0 = np.concatenate([[0], np.cumsum(nch)])
        nb = 16385  # 2^14 + 1 bin offsets per genome (GHIP_ANI_BIN_COUNT + 1)

        def cat(t, spans):
            return torch.cat([t[a:b] for a, b in spans]) if spans else t[:0]

        packed = {
            "seed_code": cat(arrs["seed_code"], [(s0[g], s0[g + 1]) for g in ids]),
            "seed_chunk": cat(arrs["seed_chunk"], [(s0[g], s0[g + 1]) for g in ids]),
            "bin_start": cat(arrs["bin_start"], [(g * nb, (g + 1) * nb) for g in ids]),
            "chunk_total": cat(arrs["chunk_total"], [(c0[g], c0[g + 1]) for g in ids]),
        }
        pmeta = {k: meta[k][ids] for k in ("glen", "cap", "cnt")}
        pmeta["sizes"] = {k: int(v.shape[0]) for k, v in packed.items()}
        return packed, pmeta

    def _chunk(self) -> int:
        return getattr(self.engine, "ani_chunk", 20000)

    def step(self) -> Dict:
        e, ex = self.engine, self.ex
        t = time.perf_counter()
        hashes_l, lens_l = e.sketch_local(self.block)
        t = self._tick("sketch", t)
        hashes = ex.all_gather_blocks(hashes_l, self.n)
        lens = ex.all_gather_blocks(lens_l, self.n)
        self._full = (hashes, lens)
        t = self._tick("allgather_sketches", t)
        pairs, replicated = e.precluster(hashes, lens, self.n, self.min_ani, self.rank, self.world)
        t = self._tick("pairs", t)
        idx_l, meta_l, lay = e.ani_build_local()
        t = self._tick("ani_index", t)
        result = {"n_pairs": 0, "n_clusters": 0, "clusters": None, "pairs": None, "pair_ani": None}
        if self.world == 1:
            idx, allp = idx_l, pairs
            alla = e.ani_pairs(idx, allp, self.min_af)
            t = self._tick("ani_pairs", t)
        else:
            if replicated:
                # the join form ran on the full matrix on every rank: each already holds the whole sorted list
                # (the branch is a function of the sketches alone, so all ranks take it together)
                allp = pairs
            else:
                # every rank learns the whole (small) candidate list
                parts = ex.all_gather_object(pairs)
                allp = np.concatenate(parts) if parts else np.zeros(0, PAIR_DTYPE)
                # (i, j) order through one u64 key: numpy's structured-field sort is ~5x slower at 10^4..10^5 pairs
                key = (allp["i"].astype(np.uint64) << np.uint64(32)) | allp["j"].astype(np.uint64)
                allp = allp[np.argsort(key, kind="stable")]
            t = self._tick("allgather_pairs", t)
            # A pair is computed where its first genome lives; only the genomes a rank needs but does
            # not own are exchanged (instead of all-gathering the whole index, ~0.33 MB per genome).
            owner_i = allp["i"] // self.block
            owner_j = allp["j"] // self.block
            needed = np.unique(allp["j"][owner_i != owner_j]).astype(np.int64)   # same on every rank
            mine_mask = owner_i == self.rank
            mine = allp[mine_mask]
            if len(needed) == 0:
                # no candidate pair spans two ranks (families do not straddle block boundaries): the local
                # index serves as it is -- no export, no collective (`needed` is identical on every rank)
                idx = idx_l
            else:
                arrs_l = e.ani_export(idx_l, lay)
                send_ids = needed[(needed >= self.first) & (needed < self.first + self.count)] - self.first
                packed, pmeta = self._pack_genomes(arrs_l, meta_l, send_ids)
                metas = ex.all_gather_object(pmeta)
                recv = {k: ex.all_gather_flat(packed[k], [m["sizes"][k] for m in metas]) for k in packed}
                rmeta = {k: np.concatenate([m[k] for m in metas]) for k in ("glen", "cap", "cnt")}
                # combined index = local genomes, then the received ones (in `needed` order: owners ascending)
                cmeta = {k: np.concatenate([meta_l[k], rmeta[k]]) for k in ("glen", "cap", "cnt")}
                import torch
                carrs = {k: torch.cat([arrs_l[k], recv[k]]) for k in arrs_l}
                idx = e.ani_wrap(cmeta, carrs)
            t = self._tick("exchange_ani_index", t)
            remap = np.full(self.n, -1, dtype=np.int64)
            remap[self.first: self.first + self.count] = np.arange(self.count)
            remap[needed] = np.where(remap[needed] >= 0, remap[needed], self.count + np.arange(len(needed)))
            local_pairs = np.zeros(len(mine), dtype=PAIR_DTYPE)
            local_pairs["i"], local_pairs["j"] = remap[mine["i"]], remap[mine["j"]]
            ani_mine = e.ani_pairs(idx, local_pairs, self.min_af)
            t = self._tick("ani_pairs", t)
            # allp is sorted by i and a pair is computed where genome i lives, so the ranks' results are consecutive
            # runs of the list in rank order, of lengths every rank can count: one collective, nothing to scatter
            alla = ex.all_gather_host_array(ani_mine.astype(np.float32, copy=False),
                                            np.bincount(owner_i, minlength=self.world).tolist())
            if len(allp) == 0:
                alla = np.zeros(0, dtype=np.float32)
            t = self._tick("gather_ani", t)
        if self.rank == 0:
            clusters = e.cluster(self.n, allp, alla, self.ani_threshold)
            t = self._tick("host_cluster", t)
            result = {"n_pairs": len(allp), "n_clusters": len(clusters), "clusters": clusters, "pairs": allp,
                      "pair_ani": alla}
        if hasattr(idx, "free"):
            idx.free()
        if self.world > 1 and idx is not idx_l and hasattr(idx_l, "free"):
            idx_l.free()
        self._steps += 1
        return result
