"""ctypes binding of the C-ABI drop-in library libgalah_hip.so (include/galah_hip.h).

Fails loudly when the HIP extension is missing: there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GHIP_LIB_OVERRIDE") or os.path.join(_HERE, "libgalah_hip.so")  # override: timing experiments only

PAIR_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("common", "<u4"), ("total", "<u4"), ("ani", "<f4")])

GHIP_OK = 0
ERROR_NAMES = {1: "GHIP_EINVAL", 2: "GHIP_EIO", 3: "GHIP_EHIP", 4: "GHIP_ENOMEM", 5: "GHIP_EUNSUPPORTED", 6: "GHIP_ECALLBACK", 7: "GHIP_EPEER"}

ANI_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float))
ANI_BATCH_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_float))


class GalahHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


# every symbol include/galah_hip.h declares: (restype, argtypes)
_vp, _sz, _u32, _u64, _f32, _int = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_float, C.c_int
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "ghip_abi_version": (_int, []),
    "ghip_ani_definition_version": (C.c_uint32, []),
    "ghip_get_options": (_int, [_vp, _vp]),
    "ghip_set_options": (_int, [_vp, _vp]),
    "ghip_device_count": (_int, []),
    "ghip_init": (_int, [_int, _pp]),
    "ghip_destroy": (None, [_vp]),
    "ghip_last_error": (C.c_char_p, [_vp]),
    "ghip_set_stream": (_int, [_vp, _vp]),
    "ghip_synchronize": (_int, [_vp]),
    "ghip_memcpy_d2d": (_int, [_vp, _vp, _vp, _sz]),
    "ghip_profile_enable": (_int, [_vp, _int]),
    "ghip_profile_reset": (_int, [_vp]),
    "ghip_kernel_stats": (_int, [_vp, C.c_char_p, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "ghip_ingest_counters": (_int, [_vp, C.POINTER(_u64)]),
    "ghip_selftest_hash_floor": (_int, [_vp, _u64, C.POINTER(C.c_double)]),
    "ghip_genomes_from_files": (_int, [_vp, C.POINTER(C.c_char_p), _sz, _int, _pp]),
    "ghip_genomes_from_host": (_int, [_vp, _vp, _vp, _sz, _pp]),
    "ghip_genomes_synthetic": (_int, [_vp, _u64, _u32, _u32, _u64, C.c_double, _pp]),
    "ghip_genomes_synthetic_range": (_int, [_vp, _u64, _u32, _u64, _u64, _u64, C.c_double, _pp]),
    "ghip_genomes_count": (_sz, [_vp]),
    "ghip_genomes_total_bases": (_u64, [_vp]),
    "ghip_genomes_length": (_u64, [_vp, _sz]),
    "ghip_genomes_to_host": (_int, [_vp, _vp, _sz, _vp]),
    "ghip_genomes_stats": (_int, [_vp, _sz, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "ghip_genomes_free": (None, [_vp]),
    "ghip_sketch_genomes": (_int, [_vp, _vp, _u32, _u32, _u64, _pp]),
    "ghip_sketch_files": (_int, [_vp, C.POINTER(C.c_char_p), _sz, _u32, _u32, _u64, _int, _pp]),
    "ghip_sketches_from_host": (_int, [_vp, _vp, _vp, _sz, _u32, _u32, _pp]),
    "ghip_sketches_wrap_device": (_int, [_vp, _vp, _vp, _sz, _u32, _u32, _pp]),
    "ghip_sketches_to_host": (_int, [_vp, _vp, _vp, _vp]),
    "ghip_sketches_copy_into": (_int, [_vp, _vp, _vp, _vp]),
    "ghip_sketches_save": (_int, [_vp, _vp, C.c_char_p]),
    "ghip_sketches_load": (_int, [_vp, C.c_char_p, _pp]),
    "ghip_sketches_save_named": (_int, [_vp, _vp, _vp, _u64, C.c_char_p]),
    "ghip_sketches_load_named": (_int, [_vp, C.c_char_p, _pp, _pp, C.POINTER(_sz), C.POINTER(_u64)]),
    "ghip_sketches_concat": (_int, [_vp, _vp, _vp, _pp]),
    "ghip_precluster_from": (_int, [_vp, _vp, _sz, _f32, _pp, C.POINTER(_sz)]),
    "ghip_sketches_count": (_sz, [_vp]),
    "ghip_sketches_size": (_u32, [_vp]),
    "ghip_sketches_kmer": (_u32, [_vp]),
    "ghip_sketches_device_hashes": (_vp, [_vp]),
    "ghip_sketches_device_lens": (_vp, [_vp]),
    "ghip_sketches_free": (None, [_vp]),
    "ghip_precluster": (_int, [_vp, _vp, _f32, _pp, C.POINTER(_sz)]),
    "ghip_precluster_shard": (_int, [_vp, _vp, _f32, _u32, _u32, _pp, C.POINTER(_sz)]),
    "ghip_precluster_ranks": (_int, [_vp, _vp, _f32, _u32, _u32, _pp, C.POINTER(_sz), C.POINTER(C.c_int)]),
    "ghip_last_pairs_compared": (_u64, [_vp]),
    "ghip_ani_index_build": (_int, [_vp, _vp, _u32, _u32, _u32, _pp]),
    "ghip_sketch_and_index": (_int, [_vp, _vp, _u32, _u32, _u64, _u32, _u32, _u32, _pp, _pp]),
    "ghip_fasta_stream": (_int, [C.c_char_p, _pp, C.POINTER(_sz), _vp]),
    "ghip_sketch_and_index_files": (_int, [_vp, _vp, _sz, _u32, _u32, _u64, _u32, _u32, _u32, _int, _u64, _pp, _vp, _vp]),
    "ghip_ani_pairs": (_int, [_vp, _vp, _vp, _sz, _f32, _vp, _vp]),
    "ghip_ani_pairs_detail": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "ghip_ani_index_free": (None, [_vp]),
    "ghip_ani_index_layout": (_int, [_vp, _vp]),
    "ghip_ani_index_meta": (_int, [_vp, _vp, _vp, _vp]),
    "ghip_ani_index_wrap_device": (_int, [_vp, _sz, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pp]),
    "ghip_comm_unique_id": (_int, [_vp]),
    "ghip_comm_init_rank": (_int, [_vp, _u32, _u32, _vp, _pp]),
    "ghip_comm_init_local": (_int, [_pp, _u32, _pp]),
    "ghip_comm_init_callback": (_int, [_vp, _u32, _u32, _vp, _vp, _pp]),
    "ghip_comm_destroy": (None, [_vp]),
    "ghip_comm_rank": (_u32, [_vp]),
    "ghip_comm_world": (_u32, [_vp]),
    "ghip_comm_transport": (C.c_char_p, [_vp]),
    "ghip_comm_last_error": (C.c_char_p, [_vp]),
    "ghip_comm_allgather_device": (_int, [_vp, _vp, _vp, _sz]),
    "ghip_comm_exchange_device": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "ghip_comm_allgather_host": (_int, [_vp, _vp, _sz, _vp]),
    "ghip_shard_range": (None, [_sz, _u32, _u32, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "ghip_allgather_sketches": (_int, [_vp, _vp, _sz, _pp]),
    "ghip_precluster_comm": (_int, [_vp, _vp, _f32, _pp, C.POINTER(_sz), C.POINTER(C.c_int)]),
    "ghip_allgather_pairs": (_int, [_vp, _vp, _sz, _pp, C.POINTER(_sz)]),
    "ghip_exchange_ani_index": (_int, [_vp, _vp, _sz, _vp, _sz, _pp, _vp]),
    "ghip_distances_and_ani_ranks": (_int, [_vp, _vp, _sz, _u32, _u32, _u64, _f32, _u32, _u32, _u32, _f32, _pp, _pp,
                                            C.POINTER(_sz), _pp, _vp]),
    "ghip_cluster_ranks": (_int, [_vp, _vp, _sz, _u32, _u32, _u64, _f32, _u32, _u32, _u32, _f32, _vp, _f32, _pp, _pp, C.POINTER(_sz), _pp,
                                  C.POINTER(_sz), _pp, _vp]),
    "ghip_cluster_index_comm": (_int, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _f32, _f32, _pp, _pp, C.POINTER(_sz), _vp]),
    "ghip_cluster_lazy_comm": (_int, [_vp, _sz, _vp, _sz, _vp, _f32, ANI_BATCH_CALLBACK, _vp, _pp, _pp, C.POINTER(_sz), _vp]),
    "ghip_comm_agree": (_int, [_vp, _int]),
    "ghip_comm_context": (_vp, [_vp]),
    "ghip_cluster_files_multi": (_int, [_pp, _u32, C.POINTER(C.c_char_p), _sz, _u32, _u32, _f32, _f32, _f32, _u32, _int, _pp, _pp,
                                        C.POINTER(_sz)]),
    "ghip_cluster": (_int, [_sz, _vp, _sz, _vp, _int, _f32, ANI_CALLBACK, _vp, _pp, _pp, C.POINTER(_sz)]),
    "ghip_cluster_index": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _f32, _f32, _pp, _pp, C.POINTER(_sz), _vp]),
    "ghip_cluster_lazy": (_int, [_sz, _vp, _sz, _f32, ANI_BATCH_CALLBACK, _vp, _pp, _pp, C.POINTER(_sz), C.POINTER(_u64)]),
    "ghip_free": (None, [_vp]),
}



ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class RankTimes(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("sketch_ms", "allgather_sketches_ms", "pairs_ms", "allgather_pairs_ms",
                                          "exchange_ani_index_ms", "ani_pairs_ms", "gather_ani_ms")] + [("pairs_compared", C.c_uint64)]


class ClusterTimes(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("sketch_ms", "allgather_sketches_ms", "pairs_ms", "allgather_pairs_ms", "exchange_ani_index_ms",
                                          "ani_rounds_ms", "cluster_host_ms")] + \
               [(k, C.c_uint64) for k in ("pairs_compared", "ani_pairs_asked", "ani_pairs_here", "lazy_rounds")]


# ghip_options (include/galah_hip.h): every switch of the library, per context
OPTION_FIELDS = ("struct_size", "pair_form", "join_ranks", "ingest_form", "ingest_groups", "io_threads_plain", "io_threads_gz", "copy_streams",
                 "use_libdeflate", "pipeline_pieces", "overlap_binning", "lazy_flush_below", "cluster_threads", "ani_force_general",
                 "ani_tall_below", "debug", "pair_debug", "fault_stage", "fault_rank", "join_fused", "probe_arranged",
                 "comm_timeout_ms", "gz_device")
PAIR_FORMS = {"auto": 0, "join": 1, "probe": 2, "merge": 3}
JOIN_RANKS = {"hash": 0, "records": 1, "replicate": 2}
INGEST_FORMS = {"packed": 0, "ascii": 1, "pageable": 2, "two-phase": 3}
FAULT_STAGES = {"none": 0, "sketch": 1, "pairs_stage1": 2, "pairs_stage2": 3, "index_pack": 4, "ani_round": 5, "gz_small_batches": 6}
_OPTION_ENUMS = {"pair_form": PAIR_FORMS, "join_ranks": JOIN_RANKS, "ingest_form": INGEST_FORMS, "fault_stage": FAULT_STAGES}


class Options(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in OPTION_FIELDS]


def get_options(ctx_handle=None) -> dict:
    """The options of a context (None: the process-wide defaults, seeded once from the GHIP_* environment)."""
    o = Options()
    check(lib().ghip_get_options(ctx_handle, C.byref(o)), ctx_handle)
    return {k: int(getattr(o, k)) for k in OPTION_FIELDS if k != "struct_size"}


def set_options(ctx_handle=None, **fields) -> dict:
    """ghip_set_options: change the named fields (enumerations by name or number); returns the previous values of those
    fields, so that `old = set_options(h, pair_form="join"); ...; set_options(h, **old)` restores them."""
    o = Options()
    check(lib().ghip_get_options(ctx_handle, C.byref(o)), ctx_handle)
    old = {}
    for k, v in fields.items():
        if k not in OPTION_FIELDS or k == "struct_size":
            raise KeyError(f"ghip_options has no field {k!r}")
        old[k] = int(getattr(o, k))
        setattr(o, k, _OPTION_ENUMS[k][v] if isinstance(v, str) else int(v))
    o.struct_size = C.sizeof(Options)
    check(lib().ghip_set_options(ctx_handle, C.byref(o)), ctx_handle)
    return old


class AniLayout(C.Structure):
    _fields_ = [("n", C.c_size_t), ("n_seed_slots", C.c_uint64), ("n_bin_slots", C.c_uint64),
                ("n_chunk_slots", C.c_uint64), ("d_seed_code", C.c_void_p), ("d_seed_loc", C.c_void_p),
                ("d_bin_start", C.c_void_p), ("d_chunk_total", C.c_void_p)]


_lib = None


def lib() -> C.CDLL:
    """Load libgalah_hip.so; raise (never fall back) when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C galah_amd/csrc). "
            "galah_amd has no CPU fallback.")
    # Load order matters in this image: the PyTorch-ROCm wheel bundles its own HIP/HSA runtime.  If
    # /opt/rocm's libamdhip64 is mapped first (by this library) and torch is imported later, torch
    # ends up with a mixed runtime and reports "No HIP GPUs are available".  Importing torch first
    # makes both share one runtime, so the plumbing (torch tensors, torch.distributed/RCCL) and
    # this library see the same device state.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError if the library does not export it
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc: int, ctx_handle=None):
    if rc != GHIP_OK:
        msg = lib().ghip_last_error(ctx_handle)
        raise GalahHipError(rc, msg.decode() if msg else "")
