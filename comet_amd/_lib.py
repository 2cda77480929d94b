"""ctypes binding of libcomet_hip.so (the C ABI declared in include/comet_gpu.h).

The product path has no CPU fallback: if the shared library (built by ``__graft_entry__.build()`` /
``make -C comet_amd/csrc``) is missing, or no gfx950 device is visible when a context is created,
this module raises — it never routes to the oracle or to any host implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libcomet_hip.so"
HEADER_PATH = _HERE.parent / "include" / "comet_gpu.h"

# status codes (include/comet_gpu.h: comet_status)
OK, ERR_INVALID_ARG, ERR_DIM_MISMATCH, ERR_ZERO_VECTOR, ERR_NOT_TRAINED, ERR_NOT_FOUND = 0, 1, 2, 3, 4, 5
ERR_ALREADY_DELETED, ERR_TRAIN_DATA, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_UNKNOWN_METRIC = 6, 7, 8, 9, 10, 11
ERR_FORMAT, ERR_IO = 12, 13

L2, L2SQ, COSINE = 0, 1, 2
KIND_FLAT, KIND_IVF, KIND_PQ, KIND_IVFPQ, KIND_HNSW, KIND_BM25 = range(6)


class CometError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class ZeroVectorError(CometError):
    """comet.ErrZeroVector (distance.go:12)"""


class SearchParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("threshold", C.c_float), ("nprobes", C.c_int32), ("ef_search", C.c_int32),
                ("filter_ids", C.POINTER(C.c_uint32)), ("n_filter", C.c_int32), ("mode", C.c_int32)]


WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)   # comet_write_cb
READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)    # comet_read_cb

_lib = None


def declared_symbols() -> list[str]:
    """Every function name include/comet_gpu.h declares (used by the CPU-side ABI test)."""
    text = HEADER_PATH.read_text()
    return sorted(set(re.findall(r"COMET_API\s+[\w\s\*]+?\b(comet_\w+)\s*\(", text)))


def load() -> C.CDLL:
    """Load libcomet_hip.so. Torch (if importable) is imported first so that both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C comet_amd/csrc` (there is no CPU fallback)")
    if os.environ.get("COMET_NO_TORCH_PRELOAD") != "1":
        try:  # same libamdhip64.so.7 SONAME: whichever loads first serves both
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    p, i32, i64, u64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
    pp = C.POINTER(C.c_void_p)
    sig = {
        "comet_last_error": (C.c_char_p, []),
        "comet_version": (C.c_char_p, []),
        "comet_device_count": (i32, [C.POINTER(i32)]),
        "comet_ctx_create": (i32, [i32, pp]),
        "comet_ctx_destroy": (i32, [p]),
        "comet_ctx_sync": (i32, [p]),
        "comet_ctx_stream": (p, [p]),
        "comet_ctx_fence": (i32, [p]),
        "comet_dev_alloc": (i32, [p, sz, pp]),
        "comet_dev_free": (i32, [p, p]),
        "comet_memcpy_h2d": (i32, [p, p, p, sz]),
        "comet_memcpy_d2h": (i32, [p, p, p, sz]),
        "comet_synth_fill_dev": (i32, [p, u64, u64, u64, p]),
        "comet_synth_mixture_dev": (i32, [p, u64, i32, f32, i32, f32, u64, u64, i32, p]),
        "comet_profile_enable": (i32, [p, i32]),
        "comet_profile_reset": (i32, [p]),
        "comet_profile_only": (i32, [p, C.c_char_p]),
        "comet_ctx_set_lanes": (i32, [p, i32]),
        "comet_profile_get": (i32, [p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
        "comet_profile_dump": (i32, [p, C.c_char_p, sz]),
        "comet_distance": (i32, [p, i32, p, p, i32, p]),
        "comet_distance_batch": (i32, [p, i32, p, i32, p, i32, p]),
        "comet_preprocess": (i32, [p, i32, p, i32, p]),
        "comet_norm_batch": (i32, [p, p, i64, i32, p]),
        "comet_normalize_batch": (i32, [p, p, i64, i32, p]),
        "comet_scale_batch": (i32, [p, p, i64, i32, f32, p]),
        "comet_kmeans": (i32, [p, p, i64, i32, i32, i32, i32, p, p, C.POINTER(i32)]),
        "comet_nearest_centroid": (i32, [p, p, i64, i32, p, i32, i32, p]),
        "comet_flat_create": (i32, [p, i32, i32, pp]),
        "comet_ivf_create": (i32, [p, i32, i32, i32, pp]),
        "comet_pq_create": (i32, [p, i32, i32, i32, i32, pp]),
        "comet_ivfpq_create": (i32, [p, i32, i32, i32, i32, i32, pp]),
        "comet_hnsw_create": (i32, [p, i32, i32, i32, i32, i32, pp]),
        "comet_hnsw_add_with_levels": (i32, [p, p, p, p, i64, C.POINTER(i64)]),
        "comet_hnsw_set_level_seed": (i32, [p, u64]),
        "comet_hnsw_export_graph": (i32, [p, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), p, p, p, p, p, C.POINTER(C.c_uint32), C.POINTER(i32)]),
        "comet_hnsw_load_graph": (i32, [p, i64, p, p, p, p, p, C.c_uint32, i32]),
        "comet_index_destroy": (i32, [p]),
        "comet_index_kind": (i32, [p]),
        "comet_index_dim": (i32, [p]),
        "comet_index_metric": (i32, [p]),
        "comet_index_trained": (i32, [p]),
        "comet_index_size": (i64, [p]),
        "comet_index_default_nprobes": (i32, [p]),
        "comet_index_train": (i32, [p, p, i64]),
        "comet_index_train_dev": (i32, [p, p, i64]),
        "comet_index_add": (i32, [p, p, p, i64, C.POINTER(i64), p]),
        "comet_index_add_dev": (i32, [p, p, p, i64, C.POINTER(i64)]),
        "comet_index_remove": (i32, [p, C.c_uint32]),
        "comet_index_flush": (i32, [p]),
        "comet_index_search": (i32, [p, p, i32, C.POINTER(SearchParams), p, p, p, i32]),
        "comet_index_search_dev": (i32, [p, p, i32, C.POINTER(SearchParams), p, p, p, i32]),
        "comet_index_search_dev_async": (i32, [p, p, i32, C.POINTER(SearchParams), p, p, p, i32, C.POINTER(u64)]),
        "comet_index_search_wait": (i32, [p, u64]),
        "comet_merge_topk_dev": (i32, [p, p, p, p, i32, i32, i32, i32, p, p, p]),
        "comet_merge_topk_packed_dev": (i32, [p, p, i64, i32, i32, i32, i32, p, p, p]),
        "comet_segments_search_dev": (i32, [pp, i32, p, i32, C.POINTER(SearchParams), p, p, p, i32]),
        "comet_segments_search": (i32, [pp, i32, p, i32, C.POINTER(SearchParams), p, p, p, i32]),
        "comet_index_get_centroids": (i32, [p, p]),
        "comet_index_get_codebooks": (i32, [p, p]),
        "comet_index_list_size": (i32, [p, i32, C.POINTER(i64)]),
        "comet_index_list_read": (i32, [p, i32, p, p, p]),
        "comet_bm25_create": (i32, [p, pp]),
        "comet_bm25_destroy": (i32, [p]),
        "comet_bm25_add": (i32, [p, C.c_uint32, p, i32]),
        "comet_bm25_remove": (i32, [p, C.c_uint32]),
        "comet_bm25_flush": (i32, [p]),
        "comet_bm25_num_docs": (i64, [p]),
        "comet_bm25_avg_doc_len": (C.c_double, [p]),
        "comet_bm25_search": (i32, [p, p, p, i32, i32, p, i32, p, p, p, p, i32]),
        "comet_hybrid_rrf_search": (i32, [p, p, p, p, p, i32, i32, i32, i32, C.c_double, p, p, p]),
        "comet_index_export": (i32, [p, p, p, p]),
        "comet_comm_unique_id": (i32, [p]),
        "comet_comm_create": (i32, [p, p, i32, i32, pp]),
        "comet_comm_destroy": (i32, [p]),
        "comet_comm_rank": (i32, [p]),
        "comet_comm_world": (i32, [p]),
        "comet_comm_allreduce_f64": (i32, [p, C.POINTER(C.c_double), i32]),
        "comet_comm_barrier": (i32, [p]),
        "comet_comm_sync": (i32, [p]),
        "comet_index_fetch_vectors": (i32, [p, p, i32, p]),
        "comet_index_set_shard": (i32, [p, i32, i32]),
        "comet_index_get_list_owners": (i32, [p, p, i32]),
        "comet_index_set_list_owners": (i32, [p, p, i32]),
        "comet_index_search_sharded_async": (i32, [p, p, p, i32, C.POINTER(SearchParams), p, p, p, i32, C.POINTER(u64)]),
        "comet_index_search_sharded_wait": (i32, [p, p, u64, i32]),
        "comet_index_write_to": (i32, [p, WRITE_CB, p, C.POINTER(i64)]),
        "comet_index_read_from": (i32, [p, READ_CB, p, C.POINTER(i64)]),
        "comet_index_serialize": (i32, [p, p, sz, C.POINTER(sz)]),
        "comet_index_deserialize": (i32, [p, p, sz, C.POINTER(sz)]),
        "comet_index_get_stat": (i32, [p, C.c_char_p, C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _ = f32
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = load().comet_last_error().decode("utf-8", "replace")
    if rc == ERR_ZERO_VECTOR:
        raise ZeroVectorError(rc, msg)
    raise CometError(rc, msg)
