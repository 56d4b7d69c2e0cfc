"""ctypes wrapper around the CPU oracle (oracle/libgalah_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), and
bench.py's cpu_baseline leg.  The product package (galah_amd) must never import this.

The oracle restates galah's finch precluster path (src/finch.rs:48-97) and host clusterer
(src/clusterer.rs) on the CPU; see galah_oracle.h for the provenance of every function.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgalah_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


class GoPair(C.Structure):
    _fields_ = [("i", C.c_uint32), ("j", C.c_uint32), ("common", C.c_uint32),
                ("total", C.c_uint32), ("ani", C.c_float)]


PAIR_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("common", "<u4"), ("total", "<u4"), ("ani", "<f4")])

_ANI_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_float))

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u64p, u32p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    L.go_murmur3_x64_128_h1.restype = C.c_uint64
    L.go_murmur3_x64_128_h1.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    L.go_murmur3_x64_128.restype = None
    L.go_murmur3_x64_128.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, u64p]
    L.go_normalize.restype = C.c_size_t
    L.go_normalize.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    L.go_sketch_file.restype = C.c_int
    L.go_sketch_file.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, u32p]
    L.go_sketch_files.restype = C.c_int
    L.go_sketch_files.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_int]
    L.go_sketch_bytes.restype = C.c_uint32
    L.go_sketch_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    for name in ("go_raw_distance", "go_raw_distance_closed_form"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, u64p, u64p]
    L.go_mash_ani.restype = C.c_double
    L.go_mash_ani.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.go_distances.restype = C.c_size_t
    L.go_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_float,
                               C.c_void_p, C.c_size_t, C.c_int]
    L.go_cache_new.restype = C.c_void_p
    L.go_cache_free.argtypes = [C.c_void_p]
    L.go_cache_insert.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_float]
    L.go_cache_get.restype = C.c_int
    L.go_cache_get.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]
    L.go_cache_contains.restype = C.c_int
    L.go_cache_contains.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    L.go_cache_len.restype = C.c_size_t
    L.go_cache_len.argtypes = [C.c_void_p]
    L.go_cache_entry.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.go_cache_transform_ids.restype = C.c_void_p
    L.go_cache_transform_ids.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t]
    L.go_cluster.restype = C.c_size_t
    L.go_cluster.argtypes = [C.c_size_t, C.c_void_p, C.c_int, C.c_float, _ANI_FN, C.c_void_p,
                             C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.go_genome_stats.restype = C.c_int
    L.go_genome_stats.argtypes = [C.c_char_p, u64p, u64p, u64p]
    L.go_splitmix64.restype = C.c_uint64
    L.go_splitmix64.argtypes = [C.c_uint64]
    L.go_synth_genome.restype = None
    L.go_synth_genome.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, C.c_void_p]
    L.go_ani_sketch_bytes.restype = C.c_void_p
    L.go_ani_sketch_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]
    L.go_ani_sketch_file.restype = C.c_int
    L.go_ani_sketch_file.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.go_ani_sketch_free.argtypes = [C.c_void_p]
    L.go_ani_sketch_nseeds.restype = C.c_size_t
    L.go_ani_sketch_nseeds.argtypes = [C.c_void_p]
    L.go_ani_sketch_seeds.restype = u64p
    L.go_ani_sketch_seeds.argtypes = [C.c_void_p]
    L.go_ani_sketch_chunks.restype = u32p
    L.go_ani_sketch_chunks.argtypes = [C.c_void_p]
    L.go_ani_sketch_positions.restype = u32p
    L.go_ani_sketch_positions.argtypes = [C.c_void_p]
    L.go_ani_sketch_strands.restype = C.POINTER(C.c_uint8)
    L.go_ani_sketch_strands.argtypes = [C.c_void_p]
    L.go_ani_sketch_length.restype = C.c_uint64
    L.go_ani_sketch_length.argtypes = [C.c_void_p]
    L.go_ani_definition_version.restype = C.c_uint32
    L.go_ani_definition_version.argtypes = []
    L.go_ani_density.restype = C.c_uint32
    L.go_ani_density.argtypes = [C.c_uint64, C.c_uint32]
    L.go_ani_sketch_density.restype = C.c_uint32
    L.go_ani_sketch_density.argtypes = [C.c_void_p]
    L.go_ani_pair_detail.restype = C.c_float
    L.go_ani_pair_detail.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.go_ani_pair_mode.restype = C.c_float
    L.go_ani_pair_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.go_ani_pair_pool_below.restype = C.c_float
    L.go_ani_pair_pool_below.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.go_ani_pair.restype = C.c_float
    L.go_ani_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    _lib = L
    return L


# ---------------------------------------------------------------- hashing / sketching
def murmur3_x64_128(key: bytes, seed: int = 0) -> Tuple[int, int]:
    out = (C.c_uint64 * 2)()
    lib().go_murmur3_x64_128(key, len(key), seed, out)
    return int(out[0]), int(out[1])


def normalize(seq: bytes) -> bytes:
    buf = C.create_string_buffer(len(seq) + 1)
    n = lib().go_normalize(seq, len(seq), buf)
    return buf.raw[:n]


def sketch_file(path: str, k: int = 21, s: int = 1000, seed: int = 0) -> np.ndarray:
    out = np.empty(s, dtype=np.uint64)
    n = C.c_uint32(0)
    rc = lib().go_sketch_file(path.encode(), k, s, seed, out.ctypes.data, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"oracle: failed to sketch {path} (rc={rc})")
    return out[: n.value].copy()


def sketch_files(paths: Sequence[str], k: int = 21, s: int = 1000, seed: int = 0,
                 threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """finch::sketch_files (src/finch.rs:69) -> packed u64[n][s] (pad = 2^64-1) + u32 len[n]."""
    n = len(paths)
    arr = (C.c_char_p * n)(*[p.encode() for p in paths])
    out = np.empty((n, s), dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    rc = lib().go_sketch_files(arr, n, k, s, seed, out.ctypes.data, lens.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError(f"oracle: failed to sketch genomes (rc={rc})")
    return out, lens


def sketch_bytes(norm: bytes | np.ndarray, k: int = 21, s: int = 1000, seed: int = 0) -> np.ndarray:
    a = np.frombuffer(norm, dtype=np.uint8) if isinstance(norm, (bytes, bytearray)) else np.ascontiguousarray(norm, dtype=np.uint8)
    out = np.empty(s, dtype=np.uint64)
    n = lib().go_sketch_bytes(a.ctypes.data, a.size, k, s, seed, out.ctypes.data)
    return out[:n].copy()


# ---------------------------------------------------------------- distance
def raw_distance(a: np.ndarray, b: np.ndarray, closed_form: bool = False) -> Tuple[int, int]:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    c, t = C.c_uint64(0), C.c_uint64(0)
    f = lib().go_raw_distance_closed_form if closed_form else lib().go_raw_distance
    f(a.ctypes.data, a.size, b.ctypes.data, b.size, C.byref(c), C.byref(t))
    return int(c.value), int(t.value)


def mash_ani(common: int, total: int, k: int = 21) -> float:
    return float(lib().go_mash_ani(common, total, k))


def distances_from_sketches(sk: np.ndarray, lens: np.ndarray, min_ani: float, k: int = 21,
                            threads: int = 1) -> np.ndarray:
    """Pair loop of finch::distances (src/finch.rs:74-96); structured array sorted by (i, j)."""
    sk = np.ascontiguousarray(sk, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n, s = sk.shape
    cap = 1 << 16
    while True:
        out = np.empty(cap, dtype=PAIR_DTYPE)
        m = lib().go_distances(sk.ctypes.data, lens.ctypes.data, n, s, k, np.float32(min_ani),
                               out.ctypes.data, cap, threads)
        if m <= cap:
            return out[:m].copy()
        cap = int(m)


def distances_rows(sk: np.ndarray, lens: np.ndarray, min_ani: float, k: int, row_lo: int, row_hi: int) -> Tuple[np.ndarray, int]:
    """Rows [row_lo, row_hi) of the outer index of finch::distances' pair loop (src/finch.rs:74-96), serial -> (hits, pairs compared)."""
    sk = np.ascontiguousarray(sk, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n, s = sk.shape
    L = lib()
    L.go_distances_rows.restype = C.c_size_t
    L.go_distances_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_float, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_uint64)]
    cap = 1 << 16
    while True:
        out = np.empty(cap, dtype=PAIR_DTYPE)
        looked = C.c_uint64(0)
        m = L.go_distances_rows(sk.ctypes.data, lens.ctypes.data, n, s, k, np.float32(min_ani), row_lo, row_hi, out.ctypes.data, cap, C.byref(looked))
        if m <= cap:
            return out[:m].copy(), int(looked.value)
        cap = int(m)


def distances(paths: Sequence[str], min_ani: float, num_kmers: int = 1000, kmer_length: int = 21,
              threads: int = 1) -> np.ndarray:
    """finch::distances(genome_fasta_paths, min_ani, num_kmers, kmer_length) (src/finch.rs:48-97)."""
    sk, lens = sketch_files(paths, kmer_length, num_kmers, 0, threads)
    return distances_from_sketches(sk, lens, min_ani, kmer_length, 1)


# ---------------------------------------------------------------- cache + clusterer
class Cache:
    """SortedPairGenomeDistanceCache (src/sorted_pair_genome_distance_cache.rs)."""

    def __init__(self, handle=None):
        self._h = handle if handle is not None else lib().go_cache_new()

    def __del__(self):
        try:
            lib().go_cache_free(self._h)
        except Exception:
            pass

    def insert(self, pair, value: Optional[float]):
        lib().go_cache_insert(self._h, pair[0], pair[1], 0 if value is None else 1,
                              np.float32(0.0 if value is None else value))

    def get(self, pair):
        """None = key absent; ("None",) semantics are expressed as (True, None)."""
        v = C.c_float(0)
        st = lib().go_cache_get(self._h, pair[0], pair[1], C.byref(v))
        if st == 0:
            return None
        return (True, None) if st == 1 else (True, np.float32(v.value))

    def contains_key(self, pair) -> bool:
        return bool(lib().go_cache_contains(self._h, pair[0], pair[1]))

    def __len__(self):
        return lib().go_cache_len(self._h)

    def items(self) -> List[Tuple[Tuple[int, int], Optional[np.float32]]]:
        out = []
        a, b, has, v = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_float()
        for idx in range(len(self)):
            lib().go_cache_entry(self._h, idx, C.byref(a), C.byref(b), C.byref(has), C.byref(v))
            out.append(((a.value, b.value), np.float32(v.value) if has.value else None))
        return out

    def transform_ids(self, ids: Sequence[int]) -> "Cache":
        arr = (C.c_size_t * len(ids))(*ids)
        return Cache(lib().go_cache_transform_ids(self._h, arr, len(ids)))

    @staticmethod
    def from_pairs(pairs: np.ndarray) -> "Cache":
        c = Cache()
        for p in pairs:
            c.insert((int(p["i"]), int(p["j"])), float(p["ani"]))
        return c


def cluster(n: int, precluster_cache: Cache, ani_threshold: float,
            calculate_ani: Optional[Callable[[int, int], Optional[float]]] = None,
            skip_clusterer: bool = False) -> List[List[int]]:
    """clusterer::cluster (src/clusterer.rs:14-152) from the precluster cache onwards.
    calculate_ani(a, b) mirrors ClusterDistanceFinder::calculate_ani on genome indices."""

    def _cb(_ctx, a, b, out):
        r = calculate_ani(a, b) if calculate_ani is not None else None
        if r is None:
            return 0
        out[0] = np.float32(r)
        return 1

    cb = _ANI_FN(_cb)
    members = (C.c_size_t * max(n, 1))()
    offsets = (C.c_size_t * (n + 1))()
    nc = lib().go_cluster(n, precluster_cache._h, 1 if skip_clusterer else 0, np.float32(ani_threshold),
                          cb, None, members, offsets)
    return [[int(members[x]) for x in range(offsets[c], offsets[c + 1])] for c in range(nc)]


def genome_stats(path: str) -> Tuple[int, int, int]:
    """calculate_genome_stats (src/genome_stats.rs:11-51) -> (num_contigs, num_ambiguous_bases, n50)."""
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = lib().go_genome_stats(path.encode(), C.byref(a), C.byref(b), C.byref(c))
    if rc != 0:
        raise RuntimeError(f"oracle: failed to read {path} (rc={rc})")
    return int(a.value), int(b.value), int(c.value)


# ---------------------------------------------------------------- synthetic genomes
def synth_genome(seed: int, species: int, member: int, length: int, sub_rate: float) -> np.ndarray:
    out = np.empty(length, dtype=np.uint8)
    lib().go_synth_genome(seed, species, member, length, sub_rate, out.ctypes.data)
    return out


# ---------------------------------------------------------------- ANI (parity unpinned)
def ani_definition_version() -> int:
    """GO_ANI_DEFINITION_VERSION: changes with every change of what the estimator returns (oracle/galah_oracle.h)."""
    return int(lib().go_ani_definition_version())


class AniSketch:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            lib().go_ani_sketch_free(self._h)
        except Exception:
            pass

    @staticmethod
    def from_bytes(g: bytes | np.ndarray, k: int = 15, c: int = 125, chunk: int = 20000) -> "AniSketch":
        a = np.frombuffer(g, dtype=np.uint8) if isinstance(g, (bytes, bytearray)) else np.ascontiguousarray(g, dtype=np.uint8)
        return AniSketch(lib().go_ani_sketch_bytes(a.ctypes.data, a.size, k, c, chunk))

    @staticmethod
    def from_file(path: str, k: int = 15, c: int = 125, chunk: int = 20000) -> "AniSketch":
        h = C.c_void_p()
        rc = lib().go_ani_sketch_file(path.encode(), k, c, chunk, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"oracle: failed to ANI-sketch {path} (rc={rc})")
        return AniSketch(h)

    @property
    def nseeds(self) -> int:
        return lib().go_ani_sketch_nseeds(self._h)

    @property
    def length(self) -> int:
        return lib().go_ani_sketch_length(self._h)

    @property
    def density(self) -> int:
        """The FracMinHash density c_g this genome was seeded at (ani_density of its length and the base c)."""
        return lib().go_ani_sketch_density(self._h)

    def seeds(self) -> np.ndarray:
        n = self.nseeds
        return np.ctypeslib.as_array(lib().go_ani_sketch_seeds(self._h), shape=(n,)).copy() if n else np.empty(0, np.uint64)

    def chunks(self) -> np.ndarray:
        n = self.nseeds
        return np.ctypeslib.as_array(lib().go_ani_sketch_chunks(self._h), shape=(n,)).copy() if n else np.empty(0, np.uint32)


    def positions(self) -> np.ndarray:
        n = self.nseeds
        return np.ctypeslib.as_array(lib().go_ani_sketch_positions(self._h), shape=(n,)).copy() if n else np.empty(0, np.uint32)

    def strands(self) -> np.ndarray:
        n = self.nseeds
        return np.ctypeslib.as_array(lib().go_ani_sketch_strands(self._h), shape=(n,)).copy() if n else np.empty(0, np.uint8)

    def locs(self, chunk: int = 20000) -> np.ndarray:
        """The device's packed seed location: chunk << 16 | strand << 15 | offset within the chunk."""
        p = self.positions().astype(np.uint64)
        return ((p // chunk) << np.uint64(16) | (self.strands().astype(np.uint64) << np.uint64(15)) | (p % chunk)).astype(np.uint32)


def ani_density(length: int, c: int = 125) -> int:
    return lib().go_ani_density(length, c)


def ani_pair_detail(q: AniSketch, r: AniSketch, min_af: float = 0.15):
    """(ANI percent, AF_q, AF_r, [M, T, aligned chunks, aligned bases of q, aligned bases of r, c_pair])."""
    afq, afr = C.c_float(0), C.c_float(0)
    d = (C.c_uint64 * 6)()
    ani = lib().go_ani_pair_detail(q._h, r._h, np.float32(min_af), C.byref(afq), C.byref(afr), d)
    return float(ani), float(afq.value), float(afr.value), [int(x) for x in d]


CHAIN, CHAIN_SPAN, AGG_WMEDIAN, AGG_POOLED, BAND_SPAN, BAND_SUB = 0, 1, 2, 4, 8, 16   # BAND_SUB | D << 8


def ani_pair_mode(q: AniSketch, r: AniSketch, flags: int, min_af: float = 0.15):
    """EXPERIMENTAL estimator variants (measurement only, scripts/ani_chain_vs_band.py): same return as ani_pair_detail."""
    afq, afr = C.c_float(0), C.c_float(0)
    d = (C.c_uint64 * 6)()
    ani = lib().go_ani_pair_mode(q._h, r._h, np.float32(min_af), flags, C.byref(afq), C.byref(afr), d)
    return float(ani), float(afq.value), float(afr.value), [int(x) for x in d]


def ani_pair_pool_below(q: AniSketch, r: AniSketch, pool_below: int, min_af: float = 0.15):
    """MEASUREMENT ONLY (scripts/ani_few_chunks.py): ani_pair_detail with another pooling limit."""
    afq, afr = C.c_float(0), C.c_float(0)
    d = (C.c_uint64 * 6)()
    ani = lib().go_ani_pair_pool_below(q._h, r._h, np.float32(min_af), pool_below, C.byref(afq), C.byref(afr), d)
    return float(ani), float(afq.value), float(afr.value), [int(x) for x in d]


def ani_pair(q: AniSketch, r: AniSketch, min_af: float = 0.15) -> Tuple[float, float, float]:
    afq, afr = C.c_float(0), C.c_float(0)
    ani = lib().go_ani_pair(q._h, r._h, np.float32(min_af), C.byref(afq), C.byref(afr))
    return float(ani), float(afq.value), float(afr.value)
