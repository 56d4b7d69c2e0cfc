"""Thin object layer over the C ABI: one Context per (process, GPU).

Plumbing only -- every call goes straight to libgalah_hip.so; nothing here computes.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import PAIR_DTYPE, GalahHipError, check

KERNELS = ("sketch_kmers", "sketch_select", "pair_table_build", "pair_intersect_tile", "pair_join", "ani_seeds", "ani_bin",
           "ani_pairs", "synth_genomes")


def fasta_stream(path: str) -> Tuple[np.ndarray, Tuple[int, int, int]]:
    """Host-only: (device-format stream of one FASTA file, (contigs, ambiguous bases, N50))."""
    p, n = C.c_void_p(), C.c_size_t(0)
    stats = (C.c_uint64 * 3)()
    check(_lib.lib().ghip_fasta_stream(path.encode(), C.byref(p), C.byref(n), stats))
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[: n.value].copy()
    _lib.lib().ghip_free(p)
    return out, (int(stats[0]), int(stats[1]), int(stats[2]))


def device_count() -> int:
    return int(_lib.lib().ghip_device_count())


class _Handle:
    _free_name = ""

    def __init__(self, ctx: "Context", handle):
        self.ctx = ctx
        self._h = handle

    def free(self):
        if self._h:
            getattr(_lib.lib(), self._free_name)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Genomes(_Handle):
    _free_name = "ghip_genomes_free"

    def __len__(self):
        return int(_lib.lib().ghip_genomes_count(self._h))

    @property
    def total_bases(self) -> int:
        return int(_lib.lib().ghip_genomes_total_bases(self._h))

    def length(self, idx: int) -> int:
        return int(_lib.lib().ghip_genomes_length(self._h, idx))

    def stats(self, idx: int) -> Tuple[int, int, int]:
        """(num_contigs, num_ambiguous_bases, n50) -- GenomeAssemblyStats of src/genome_stats.rs."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(_lib.lib().ghip_genomes_stats(self._h, idx, C.byref(a), C.byref(b), C.byref(c)), self.ctx._h)
        return int(a.value), int(b.value), int(c.value)

    def to_host(self, idx: int) -> np.ndarray:
        out = np.empty(self.length(idx), dtype=np.uint8)
        check(_lib.lib().ghip_genomes_to_host(self.ctx._h, self._h, idx, out.ctypes.data), self.ctx._h)
        return out


class Sketches(_Handle):
    _free_name = "ghip_sketches_free"

    def __len__(self):
        return int(_lib.lib().ghip_sketches_count(self._h))

    @property
    def size(self) -> int:
        return int(_lib.lib().ghip_sketches_size(self._h))

    @property
    def kmer(self) -> int:
        return int(_lib.lib().ghip_sketches_kmer(self._h))

    @property
    def device_hashes(self) -> int:
        return int(_lib.lib().ghip_sketches_device_hashes(self._h) or 0)

    @property
    def device_lens(self) -> int:
        return int(_lib.lib().ghip_sketches_device_lens(self._h) or 0)

    def save(self, path: str, names: Optional[Sequence[str]] = None, seed: int = 0):
        """Persist the matrix ("GHIPSK02": k, s, hash seed, row lengths, hashes, genome names, checksum)."""
        if names is None:
            check(_lib.lib().ghip_sketches_save(self.ctx._h, self._h, path.encode()), self.ctx._h)
            return
        assert len(names) == len(self), "one name per sketch row"
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        check(_lib.lib().ghip_sketches_save_named(self.ctx._h, self._h, arr, seed, path.encode()), self.ctx._h)

    def to_host(self) -> Tuple[np.ndarray, np.ndarray]:
        n, s = len(self), self.size
        hashes = np.empty((n, s), dtype=np.uint64)
        lens = np.empty(n, dtype=np.uint32)
        check(_lib.lib().ghip_sketches_to_host(self.ctx._h, self._h, hashes.ctypes.data, lens.ctypes.data), self.ctx._h)
        return hashes, lens


class AniIndex(_Handle):
    _free_name = "ghip_ani_index_free"

    def layout(self) -> "_lib.AniLayout":
        lay = _lib.AniLayout()
        check(_lib.lib().ghip_ani_index_layout(self._h, C.byref(lay)), self.ctx._h)
        return lay

    def meta(self):
        """(genome_len u64[n], seed_cap u64[n], seed_count u32[n]) host arrays."""
        n = self.layout().n
        glen, cap = (np.zeros(n, dtype=np.uint64) for _ in range(2))
        cnt = np.zeros(n, dtype=np.uint32)
        check(_lib.lib().ghip_ani_index_meta(self._h, glen.ctypes.data, cap.ctypes.data, cnt.ctypes.data), self.ctx._h)
        return glen, cap, cnt


class Context:
    def __init__(self, device: int = 0):
        L = _lib.lib()
        h = C.c_void_p()
        rc = L.ghip_init(device, C.byref(h))
        if rc != 0:
            msg = L.ghip_last_error(None)
            raise GalahHipError(rc, msg.decode() if msg else "")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            _lib.lib().ghip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- options (ghip_options: what the GHIP_* environment variables only seed the defaults of)
    def options(self) -> Dict[str, int]:
        return _lib.get_options(self._h)

    def set_options(self, **fields) -> Dict[str, int]:
        """Change the named ghip_options fields of THIS context; returns their previous values."""
        return _lib.set_options(self._h, **fields)

    @contextlib.contextmanager
    def with_options(self, **fields):
        old = self.set_options(**fields)
        try:
            yield self
        finally:
            self.set_options(**old)

    # ---- plumbing
    def set_stream(self, hip_stream: Optional[int]):
        check(_lib.lib().ghip_set_stream(self._h, C.c_void_p(hip_stream or 0)), self._h)

    def synchronize(self):
        check(_lib.lib().ghip_synchronize(self._h), self._h)

    def memcpy_d2d(self, dst: int, src: int, nbytes: int):
        check(_lib.lib().ghip_memcpy_d2d(self._h, C.c_void_p(dst), C.c_void_p(src), nbytes), self._h)

    def profile(self, enable: bool = True):
        check(_lib.lib().ghip_profile_enable(self._h, 1 if enable else 0), self._h)

    def profile_reset(self):
        check(_lib.lib().ghip_profile_reset(self._h), self._h)

    def kernel_stats(self) -> Dict[str, Tuple[int, float]]:
        out = {}
        for k in KERNELS:
            n, ms = C.c_uint64(0), C.c_double(0)
            check(_lib.lib().ghip_kernel_stats(self._h, k.encode(), C.byref(n), C.byref(ms)), self._h)
            out[k] = (int(n.value), float(ms.value))
        return out

    def ingest_counters(self) -> Dict[str, int]:
        """ghip_ingest_counters: gzip files inflated on the device / left to the host's inflate, device microseconds of that path."""
        out = (C.c_uint64 * 4)()
        check(_lib.lib().ghip_ingest_counters(self._h, out), self._h)
        return {"gz_device_files": int(out[0]), "gz_host_files": int(out[1]), "gz_device_us": int(out[2]), "two_phase_repeats": int(out[3])}

    def hash_floor_ms(self, wave_positions: int) -> float:
        """Duration of the MurmurHash3 filter instructions alone for `wave_positions` wave-level evaluations."""
        ms = C.c_double(0)
        check(_lib.lib().ghip_selftest_hash_floor(self._h, wave_positions, C.byref(ms)), self._h)
        return float(ms.value)

    # ---- ingest
    def genomes_from_files(self, paths: Sequence[str], io_threads: int = 1) -> Genomes:
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p()
        check(_lib.lib().ghip_genomes_from_files(self._h, arr, len(paths), io_threads, C.byref(h)), self._h)
        return Genomes(self, h)

    def genomes_from_host(self, streams: Sequence[bytes | np.ndarray]) -> Genomes:
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        offsets = np.zeros(len(arrs) + 1, dtype=np.uint64)
        for i, a in enumerate(arrs):
            offsets[i + 1] = offsets[i] + a.size
        flat = np.concatenate(arrs) if arrs else np.empty(0, dtype=np.uint8)
        flat = np.ascontiguousarray(flat)
        h = C.c_void_p()
        check(_lib.lib().ghip_genomes_from_host(self._h, flat.ctypes.data, offsets.ctypes.data, len(arrs), C.byref(h)), self._h)
        return Genomes(self, h)

    def genomes_synthetic_range(self, seed: int, members: int, first: int, count: int, length: int,
                                sub_rate: float) -> Genomes:
        h = C.c_void_p()
        check(_lib.lib().ghip_genomes_synthetic_range(self._h, seed, members, first, count, length, sub_rate,
                                                      C.byref(h)), self._h)
        return Genomes(self, h)

    def genomes_synthetic(self, seed: int, n_species: int, members: int, length: int, sub_rate: float) -> Genomes:
        h = C.c_void_p()
        check(_lib.lib().ghip_genomes_synthetic(self._h, seed, n_species, members, length, sub_rate, C.byref(h)), self._h)
        return Genomes(self, h)

    # ---- sketching
    def sketch_genomes(self, g: Genomes, k: int = 21, s: int = 1000, seed: int = 0) -> Sketches:
        h = C.c_void_p()
        check(_lib.lib().ghip_sketch_genomes(self._h, g._h, k, s, seed, C.byref(h)), self._h)
        return Sketches(self, h)

    def sketch_files(self, paths: Sequence[str], k: int = 21, s: int = 1000, seed: int = 0, io_threads: int = 1) -> Sketches:
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = C.c_void_p()
        check(_lib.lib().ghip_sketch_files(self._h, arr, len(paths), k, s, seed, io_threads, C.byref(h)), self._h)
        return Sketches(self, h)

    def sketches_from_host(self, hashes: np.ndarray, lens: np.ndarray, k: int = 21) -> Sketches:
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n, s = hashes.shape
        h = C.c_void_p()
        check(_lib.lib().ghip_sketches_from_host(self._h, hashes.ctypes.data, lens.ctypes.data, n, s, k, C.byref(h)), self._h)
        return Sketches(self, h)

    def sketches_load(self, path: str) -> Sketches:
        h = C.c_void_p()
        check(_lib.lib().ghip_sketches_load(self._h, path.encode(), C.byref(h)), self._h)
        return Sketches(self, h)

    def sketches_load_named(self, path: str) -> Tuple[Sketches, List[str], int]:
        """-> (matrix, genome names in row order, hash seed) of a persisted matrix; a damaged file raises."""
        h, names, nb, seed = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
        check(_lib.lib().ghip_sketches_load_named(self._h, path.encode(), C.byref(h), C.byref(names), C.byref(nb), C.byref(seed)), self._h)
        try:
            blob = C.string_at(names, nb.value) if names.value else b""
        finally:
            _lib.lib().ghip_free(names)
        sk = Sketches(self, h)
        parts = blob.split(b"\0")[: len(sk)]
        return sk, [p.decode() for p in parts], int(seed.value)

    def sketches_concat(self, a: Sketches, b: Sketches) -> Sketches:
        h = C.c_void_p()
        check(_lib.lib().ghip_sketches_concat(self._h, a._h, b._h, C.byref(h)), self._h)
        return Sketches(self, h)

    def precluster_from(self, sk: Sketches, row_lo: int, min_ani: float) -> np.ndarray:
        """The (new x all) rectangle of an incremental run: pairs (i, j), i < j, with j >= row_lo (ghip_precluster_from)."""
        p, n = C.c_void_p(), C.c_size_t(0)
        check(_lib.lib().ghip_precluster_from(self._h, sk._h, row_lo, np.float32(min_ani), C.byref(p), C.byref(n)), self._h)
        return self._take_pairs(p, n)

    def sketches_wrap_device(self, d_hashes: int, d_lens: int, n: int, s: int, k: int = 21) -> Sketches:
        h = C.c_void_p()
        check(_lib.lib().ghip_sketches_wrap_device(self._h, C.c_void_p(d_hashes), C.c_void_p(d_lens), n, s, k, C.byref(h)), self._h)
        return Sketches(self, h)

    def sketches_copy_into(self, sk: Sketches, d_hashes: int, d_lens: int):
        check(_lib.lib().ghip_sketches_copy_into(self._h, sk._h, C.c_void_p(d_hashes), C.c_void_p(d_lens)), self._h)

    # ---- precluster
    @staticmethod
    def _take_pairs(p, n) -> np.ndarray:
        try:
            if n.value == 0:
                return np.empty(0, dtype=PAIR_DTYPE)
            buf = (C.c_char * (n.value * PAIR_DTYPE.itemsize)).from_address(p.value)
            return np.frombuffer(buf, dtype=PAIR_DTYPE).copy()
        finally:
            _lib.lib().ghip_free(p)

    def precluster(self, sk: Sketches, min_ani: float, rank: int = 0, world: int = 1) -> np.ndarray:
        p = C.c_void_p()
        n = C.c_size_t(0)
        check(_lib.lib().ghip_precluster_shard(self._h, sk._h, np.float32(min_ani), rank, world, C.byref(p), C.byref(n)), self._h)
        return self._take_pairs(p, n)

    def precluster_ranks(self, sk: Sketches, min_ani: float, rank: int, world: int) -> Tuple[np.ndarray, bool]:
        """-> (pairs, replicated): the whole list on every rank when the join form ran, else this rank's share."""
        p = C.c_void_p()
        n = C.c_size_t(0)
        rep = C.c_int(0)
        check(_lib.lib().ghip_precluster_ranks(self._h, sk._h, np.float32(min_ani), rank, world, C.byref(p), C.byref(n),
                                               C.byref(rep)), self._h)
        return self._take_pairs(p, n), bool(rep.value)

    @property
    def last_pairs_compared(self) -> int:
        return int(_lib.lib().ghip_last_pairs_compared(self._h))

    # ---- ANI
    def ani_index_build(self, g: Genomes, k: int = 15, c: int = 125, chunk: int = 20000) -> AniIndex:
        h = C.c_void_p()
        check(_lib.lib().ghip_ani_index_build(self._h, g._h, k, c, chunk, C.byref(h)), self._h)
        return AniIndex(self, h)

    def sketch_and_index(self, g: Genomes, k: int = 21, s: int = 1000, seed: int = 0, ani_k: int = 15,
                         ani_c: int = 125, ani_chunk: int = 20000) -> Tuple[Sketches, AniIndex]:
        hs, hi = C.c_void_p(), C.c_void_p()
        check(_lib.lib().ghip_sketch_and_index(self._h, g._h, k, s, seed, ani_k, ani_c, ani_chunk, C.byref(hs),
                                               C.byref(hi)), self._h)
        return Sketches(self, hs), AniIndex(self, hi)

    def sketch_and_index_files(self, paths: Sequence[str], k: int = 21, s: int = 1000, seed: int = 0, ani_k: int = 15,
                               ani_c: int = 125, ani_chunk: int = 20000, io_threads: int = 1, batch_bytes: int = 0,
                               want_index: bool = True):
        """Files in -> (Sketches, AniIndex or None, stats u64[n][3] = contigs, ambiguous bases, N50): one read of every
        file, at most `batch_bytes` of bases in HBM at a time (0 = the library default)."""
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        hs, hi = C.c_void_p(), C.c_void_p()
        stats = np.zeros((len(paths), 3), dtype=np.uint64)
        check(_lib.lib().ghip_sketch_and_index_files(self._h, arr, len(paths), k, s, seed, ani_k, ani_c, ani_chunk, io_threads,
                                                     batch_bytes, C.byref(hs), C.byref(hi) if want_index else None,
                                                     stats.ctypes.data), self._h)
        return Sketches(self, hs), (AniIndex(self, hi) if want_index else None), stats

    def ani_index_wrap_device(self, k: int, c: int, chunk: int, genome_len, seed_cap, seed_count,
                              d_seed_code: int, d_seed_loc: int, d_bin_start: int, d_chunk_total: int) -> AniIndex:
        glen = np.ascontiguousarray(genome_len, dtype=np.uint64)
        cap = np.ascontiguousarray(seed_cap, dtype=np.uint64)
        cnt = np.ascontiguousarray(seed_count, dtype=np.uint32)
        h = C.c_void_p()
        check(_lib.lib().ghip_ani_index_wrap_device(self._h, len(glen), k, c, chunk, glen.ctypes.data, cap.ctypes.data,
                                                    cnt.ctypes.data, C.c_void_p(d_seed_code),
                                                    C.c_void_p(d_seed_loc), C.c_void_p(d_bin_start),
                                                    C.c_void_p(d_chunk_total), C.byref(h)), self._h)
        return AniIndex(self, h)

    def ani_pairs(self, idx: AniIndex, pairs: np.ndarray, min_af: float = 0.15, want_af: bool = False):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        n = pairs.shape[0]
        out = np.zeros(n, dtype=np.float32)
        af = np.zeros((n, 2), dtype=np.float32) if want_af else None
        check(_lib.lib().ghip_ani_pairs(self._h, idx._h, pairs.ctypes.data, n, np.float32(min_af), out.ctypes.data,
                                        af.ctypes.data if want_af else None), self._h)
        return (out, af) if want_af else out

    def cluster_index(self, idx: AniIndex, n_genomes: int, pairs: np.ndarray, ani_threshold: float, min_af: float = 0.15,
                      order: Optional[np.ndarray] = None) -> Tuple["ClusterList", Dict[str, float]]:
        """ghip_cluster_index: the greedy clusterer with the resident ANI index answering its lazy rounds, whole in native
        code.  `order` (optional) = the quality order, order[x] = genome that comes x-th; clusters then hold positions x.
        -> (clusters, {"asked", "rounds", "ani_ms", "total_ms"})."""
        L = _lib.lib()
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
            assert order.shape == (n_genomes,)
        members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        stats = np.zeros(4, dtype=np.uint64)
        check(L.ghip_cluster_index(self._h, idx._h if idx is not None else None, n_genomes, pairs.ctypes.data, pairs.shape[0],
                                   order.ctypes.data if order is not None else None, np.float32(ani_threshold), np.float32(min_af),
                                   C.byref(members), C.byref(offsets), C.byref(nc), stats.ctypes.data), self._h)
        try:
            off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
            mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()
        finally:
            L.ghip_free(members)
            L.ghip_free(offsets)
        return (ClusterList(mem, off),
                {"asked": int(stats[0]), "rounds": int(stats[1]), "ani_ms": float(stats[2]) * 1e-6, "total_ms": float(stats[3]) * 1e-6})

    def ani_pairs_detail(self, idx: AniIndex, pairs: np.ndarray) -> np.ndarray:
        """u64[n][6] = M, T of the lower-median chunk, aligned chunks, aligned bases of q, of r, c_pair (ghip_ani_pairs_detail)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros((pairs.shape[0], 6), dtype=np.uint64)
        check(_lib.lib().ghip_ani_pairs_detail(self._h, idx._h, pairs.ctypes.data, pairs.shape[0], out.ctypes.data), self._h)
        return out


class ClusterList(Sequence):
    """Clusters as the C ABI returns them -- members[offsets[c] : offsets[c + 1]], representative first -- behaving like the
    list of lists clusterer::cluster returns (indexing, iteration, len, == with a list of lists); the Python lists are only
    built when somebody looks (a step that clusters 10 000 genomes spent a third of its host time building 2 000 lists)."""

    def __init__(self, members: np.ndarray, offsets: np.ndarray):
        self.members, self.offsets = members, offsets
        self._lists = None

    def tolist(self) -> List[List[int]]:
        if self._lists is None:
            mem, off = self.members.tolist(), self.offsets.tolist()
            self._lists = [mem[off[c]:off[c + 1]] for c in range(len(off) - 1)]
        return self._lists

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, c):
        if isinstance(c, slice) or self._lists is not None:
            return self.tolist()[c]
        if c < 0:
            c += len(self)
        if not 0 <= c < len(self):
            raise IndexError(c)
        return self.members[int(self.offsets[c]):int(self.offsets[c + 1])].tolist()

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        if isinstance(other, ClusterList):
            return np.array_equal(self.offsets, other.offsets) and np.array_equal(self.members[: int(self.offsets[-1])], other.members[: int(other.offsets[-1])])
        return self.tolist() == other

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return repr(self.tolist())


def cluster_pairs(n_genomes: int, pairs: np.ndarray, ani_threshold: float, pair_ani: Optional[np.ndarray] = None,
                  skip_clusterer: bool = False, ani_callback=None) -> List[List[int]]:
    """ghip_cluster: clusterer::cluster from the precluster cache onwards (host, no GPU needed)."""
    L = _lib.lib()
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    pa = None
    if pair_ani is not None:
        pa = np.ascontiguousarray(pair_ani, dtype=np.float32)
        assert pa.shape[0] == pairs.shape[0]

    failure = []

    def _cb(_user, a, b, out):
        # an exception must not vanish inside ctypes (it would read as calculate_ani == None and the clustering would
        # go on): stash it, tell ghip_cluster to stop (< 0), re-raise below -- the reference would panic here
        try:
            r = ani_callback(int(a), int(b))
        except BaseException as e:  # noqa: BLE001
            failure.append(e)
            return -1
        if r is None:
            return 0
        out[0] = np.float32(r)
        return 1

    cb = _lib.ANI_CALLBACK(_cb) if ani_callback is not None else C.cast(None, _lib.ANI_CALLBACK)
    members, offsets, nc = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    rc = L.ghip_cluster(n_genomes, pairs.ctypes.data, pairs.shape[0], pa.ctypes.data if pa is not None else None,
                        1 if skip_clusterer else 0, np.float32(ani_threshold), cb, None,
                        C.byref(members), C.byref(offsets), C.byref(nc))
    if failure:
        raise failure[0]
    if rc != 0:
        raise GalahHipError(rc, "ghip_cluster failed (no representative with a known ANI, or bad input)")
    try:
        off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
        total = int(off[-1])
        mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(total, 1),)).copy()
    finally:
        L.ghip_free(members)
        L.ghip_free(offsets)
    mem_l, off_l = mem.tolist(), off.tolist()   # plain ints: slicing numpy scalars cluster by cluster is ~10x slower
    return [mem_l[off_l[c]:off_l[c + 1]] for c in range(nc.value)]


def cluster_pairs_lazy(n_genomes: int, pairs: np.ndarray, ani_threshold: float, ani_of_edges) -> Tuple[List[List[int]], int]:
    """ghip_cluster_lazy: the greedy clusterer with the ANI asked for in batches, only for the precluster pairs that touch
    a representative.  ani_of_edges(edge_indices: uint32 array) -> float32 array (NaN = None), one call per round.
    -> (clusters, number of pairs asked)."""
    L = _lib.lib()
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    failure = []

    def _cb(_user, edges, n, out):
        try:
            idx = np.ctypeslib.as_array(edges, shape=(n,))
            vals = np.ascontiguousarray(ani_of_edges(idx), dtype=np.float32)
            assert vals.shape == (n,)
            C.memmove(out, vals.ctypes.data, 4 * n)
            return 0
        except BaseException as e:  # noqa: BLE001 -- must not vanish inside ctypes
            failure.append(e)
            return 1

    cb = _lib.ANI_BATCH_CALLBACK(_cb)
    members, offsets, nc, asked = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
    rc = L.ghip_cluster_lazy(n_genomes, pairs.ctypes.data, pairs.shape[0], np.float32(ani_threshold), cb, None,
                             C.byref(members), C.byref(offsets), C.byref(nc), C.byref(asked))
    if failure:
        raise failure[0]
    if rc != 0:
        raise GalahHipError(rc, "ghip_cluster_lazy failed (no representative with a known ANI, or bad input)")
    try:
        off = np.ctypeslib.as_array(C.cast(offsets, C.POINTER(C.c_uint64)), shape=(nc.value + 1,)).copy()
        mem = np.ctypeslib.as_array(C.cast(members, C.POINTER(C.c_uint32)), shape=(max(int(off[-1]), 1),)).copy()
    finally:
        L.ghip_free(members)
        L.ghip_free(offsets)
    mem_l, off_l = mem.tolist(), off.tolist()
    return [mem_l[off_l[c]:off_l[c + 1]] for c in range(nc.value)], int(asked.value)
