"""ctypes wrapper around oracle/libcomet_oracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's Go loops (oracle/comet_oracle.cpp). Only tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "libcomet_oracle.so"
L2, L2SQ, COSINE = 0, 1, 2
METRIC = {"l2": L2, "l2_squared": L2SQ, "cosine": COSINE}
ERR_ZERO_VECTOR, ERR_NOT_TRAINED, ERR_NOT_FOUND, ERR_ALREADY_DELETED, ERR_TRAIN_DATA = -1, -3, -4, -6, -7

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not SO.exists():
            subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
        L = C.CDLL(str(SO))
        vp, i32, u32, f32, u64, i64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_uint64, C.c_int64, C.c_double
        sig = {
            "orc_distance": (f32, [i32, vp, vp, i32]), "orc_distance_batch": (None, [i32, vp, i32, vp, i32, vp]),
            "orc_preprocess": (i32, [i32, vp, i32, vp]), "orc_norm": (f32, [vp, i32]), "orc_scale": (None, [vp, i32, f32, vp]),
            "orc_normalize": (None, [vp, i32, vp]), "orc_sanitize_k": (i32, [i32, i32]), "orc_autocut": (i32, [vp, i32, i32]),
            "orc_flat_new": (vp, [i32, i32]), "orc_flat_free": (None, [vp]), "orc_flat_add": (i32, [vp, u32, vp]),
            "orc_flat_remove": (i32, [vp, u32]), "orc_flat_flush": (None, [vp]), "orc_flat_size": (i32, [vp]),
            "orc_flat_vectors": (vp, [vp]),
            "orc_flat_search": (i32, [vp, vp, i32, f32, vp, i32, vp, vp, i32]),
            "orc_kmeans": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]), "orc_nearest_centroid": (i32, [vp, vp, i32, i32, i32]),
            "orc_ivf_new": (vp, [i32, i32, i32]), "orc_ivf_free": (None, [vp]), "orc_ivf_train": (i32, [vp, vp, i32]),
            "orc_ivf_add": (i32, [vp, u32, vp]), "orc_ivf_remove": (i32, [vp, u32]), "orc_ivf_centroids": (vp, [vp]),
            "orc_ivf_list_size": (i32, [vp, i32]), "orc_ivf_search": (i32, [vp, vp, i32, i32, f32, vp, i32, vp, vp, i32]),
            "orc_pq_new": (vp, [i32, i32, i32, i32]), "orc_pq_free": (None, [vp]), "orc_pq_train": (i32, [vp, vp, i32]),
            "orc_pq_add": (i32, [vp, u32, vp]), "orc_pq_remove": (i32, [vp, u32]), "orc_pq_codebooks": (vp, [vp]),
            "orc_pq_codes": (vp, [vp]), "orc_pq_size": (i32, [vp]),
            "orc_pq_search": (i32, [vp, vp, i32, f32, vp, i32, vp, vp, i32]),
            "orc_ivfpq_new": (vp, [i32, i32, i32, i32, i32]), "orc_ivfpq_free": (None, [vp]), "orc_ivfpq_train": (i32, [vp, vp, i32]),
            "orc_ivfpq_set_quantizers": (i32, [vp, vp, vp]), "orc_ivfpq_append_encoded": (i32, [vp, i32, i32, vp, vp]),
            "orc_ivfpq_add": (i32, [vp, u32, vp]), "orc_ivfpq_remove": (i32, [vp, u32]), "orc_ivfpq_centroids": (vp, [vp]),
            "orc_ivfpq_codebooks": (vp, [vp]), "orc_ivfpq_list_size": (i32, [vp, i32]), "orc_ivfpq_list_codes": (vp, [vp, i32]),
            "orc_ivfpq_list_ids": (vp, [vp, i32]),
            "orc_ivfpq_search": (i32, [vp, vp, i32, i32, f32, vp, i32, vp, vp, i32]),
            "orc_hnsw_new": (vp, [i32, i32, i32, i32, i32, u64]), "orc_hnsw_free": (None, [vp]),
            "orc_hnsw_add": (i32, [vp, u32, vp]), "orc_hnsw_add_with_level": (i32, [vp, u32, vp, i32]),
            "orc_hnsw_remove": (i32, [vp, u32]), "orc_hnsw_size": (i32, [vp]), "orc_hnsw_max_level": (i32, [vp]),
            "orc_hnsw_entry": (u32, [vp]), "orc_hnsw_stats": (None, [vp, vp, vp, i32]),
            "orc_hnsw_export": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
            "orc_hnsw_search": (i32, [vp, vp, i32, i32, f32, vp, i32, vp, vp, i32]),
            "orc_go_log": (dbl, [dbl]),
            "orc_bm25_new": (vp, []), "orc_bm25_free": (None, [vp]), "orc_bm25_add": (i32, [vp, u32, vp, i32]),
            "orc_bm25_remove": (i32, [vp, u32]), "orc_bm25_flush": (None, [vp]), "orc_bm25_num_docs": (u32, [vp]), "orc_bm25_avg_doc_len": (dbl, [vp]),
            "orc_bm25_search": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, vp, i32]),
            "orc_aggregate": (i32, [i32, vp, vp, i32, vp, vp]),
            "orc_rrf": (i32, [dbl, vp, vp, i32, vp, vp, i32, vp, vp]),
            "orc_synth_fill": (None, [u64, u64, u64, vp]), "orc_synth_mixture": (None, [u64, i32, f32, i32, f32, u64, u64, i32, vp]),
            "orc_ivf_flush": (None, [vp]), "orc_pq_flush": (None, [vp]), "orc_ivfpq_flush": (None, [vp]), "orc_hnsw_flush": (None, [vp]),
            "orc_flat_write": (i64, [vp, vp, i64]), "orc_flat_read": (i64, [vp, vp, i64]),
            "orc_ivf_write": (i64, [vp, vp, i64]), "orc_ivf_read": (i64, [vp, vp, i64]),
            "orc_pq_write": (i64, [vp, vp, i64]), "orc_pq_read": (i64, [vp, vp, i64]),
            "orc_ivfpq_write": (i64, [vp, vp, i64]), "orc_ivfpq_read": (i64, [vp, vp, i64]),
            "orc_hnsw_write": (i64, [vp, vp, i64]), "orc_hnsw_read": (i64, [vp, vp, i64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _ = i64
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def distance(metric, a, b) -> np.float32:
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_distance(METRIC[metric], _p(a), _p(b), a.shape[0]))


def distance_batch(metric, queries, target) -> np.ndarray:
    q, t = _f32(queries), _f32(target)
    out = np.empty(q.shape[0], np.float32)
    lib().orc_distance_batch(METRIC[metric], _p(q), q.shape[0], _p(t), t.shape[0], _p(out))
    return out


def preprocess(metric, x):
    x = _f32(x)
    out = np.empty_like(x)
    rc = lib().orc_preprocess(METRIC[metric], _p(x), x.shape[0], _p(out))
    return (None, rc) if rc else (out, 0)


def preprocess_rows(metric, X):
    X = _f32(X)
    out = np.empty_like(X)
    for i in range(X.shape[0]):
        rc = lib().orc_preprocess(METRIC[metric], _p(X[i]), X.shape[1], _p(out[i]))
        assert rc == 0
    return out


def norm(v):
    v = _f32(v)
    return np.float32(lib().orc_norm(_p(v), v.shape[0]))


def normalize(v):
    v = _f32(v)
    out = np.empty_like(v)
    lib().orc_normalize(_p(v), v.shape[0], _p(out))
    return out


def scale(v, s):
    v = _f32(v)
    out = np.empty_like(v)
    lib().orc_scale(_p(v), v.shape[0], C.c_float(s), _p(out))
    return out


def autocut(scores, cutoff) -> int:
    s = _f32(scores)
    return lib().orc_autocut(_p(s), s.shape[0], cutoff)


def kmeans(vectors, k, metric="l2_squared", max_iter=20):
    v = _f32(vectors)
    if v.ndim != 2 or v.shape[0] == 0:
        return None, None
    n, d = v.shape
    cent = np.zeros((max(1, min(max(k, 1), n)), d), np.float32)
    assign = np.zeros(n, np.int32)
    ke = lib().orc_kmeans(_p(v), n, d, k, METRIC[metric], max_iter, _p(cent), _p(assign))
    if ke == 0:
        return None, None
    return cent[:ke], assign


def nearest_centroid(v, centroids, metric):
    v, c = _f32(v), _f32(centroids)
    return lib().orc_nearest_centroid(_p(v), _p(c), c.shape[0], c.shape[1], METRIC[metric])


def synth(seed, offset, n):
    out = np.empty(n, np.float32)
    lib().orc_synth_fill(C.c_uint64(seed), C.c_uint64(offset), C.c_uint64(n), _p(out))
    return out


def synth_mixture(seed, n_centers, sigma, n_sub, sigma_noise, row_base, n_rows, dim):
    out = np.empty((n_rows, dim), np.float32)
    lib().orc_synth_mixture(C.c_uint64(seed), int(n_centers), C.c_float(sigma), int(n_sub), C.c_float(sigma_noise), C.c_uint64(row_base),
                            C.c_uint64(n_rows), int(dim), _p(out))
    return out


class _Index:
    _free = None
    _search = None
    _io = None     # "flat" / "ivf" / "pq" / "ivfpq" / "hnsw"

    def to_bytes(self) -> bytes:
        """WriteTo: flushes, then the reference's on-disk layout."""
        fn = getattr(lib(), f"orc_{self._io}_write")
        n = fn(self.h, None, 0)
        buf = (C.c_uint8 * max(1, n))()
        assert fn(self.h, buf, n) == n
        return bytes(buf[:n])

    def from_bytes(self, data: bytes) -> int:
        """ReadFrom: bytes consumed, or a negative error."""
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
        n = getattr(lib(), f"orc_{self._io}_read")(self.h, buf, len(data))
        if n >= 0 and hasattr(self, "n"):
            self.n = self._count()
        return n

    def flush(self):
        getattr(lib(), f"orc_{self._io}_flush")(self.h)

    def __del__(self):
        try:
            if self.h:
                getattr(lib(), self._free)(self.h)
                self.h = None
        except Exception:
            pass

    def _do_search(self, fn, q, k, extra, threshold, filter_ids, cap):
        q = _f32(q)
        cap = cap or max(1, self.capacity())
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.float32)
        flt = np.ascontiguousarray(list(filter_ids), dtype=np.uint32)
        args = [self.h, _p(q), int(k)] + list(extra) + [C.c_float(threshold), _p(flt) if flt.size else None, int(flt.size), _p(ids), _p(sc), cap]
        n = fn(*args)
        if n < 0:
            return n, None, None
        m = min(n, cap)
        return n, ids[:m], sc[:m]


class Flat(_Index):
    _free = "orc_flat_free"
    _io = "flat"

    def __init__(self, dim, metric):
        self.dim, self.metric = dim, metric
        self.h = lib().orc_flat_new(dim, METRIC[metric])

    def add(self, i, v):
        v = _f32(v)
        return lib().orc_flat_add(self.h, int(i), _p(v))

    def add_batch(self, ids, X):
        X = _f32(X)
        for i, v in zip(ids, X):
            rc = lib().orc_flat_add(self.h, int(i), _p(v))
            if rc:
                return rc
        return 0

    def remove(self, i):
        return lib().orc_flat_remove(self.h, int(i))

    def flush(self):
        lib().orc_flat_flush(self.h)

    def capacity(self):
        return lib().orc_flat_size(self.h)

    def search(self, q, k, threshold=0.0, filter_ids=(), cap=None):
        return self._do_search(lib().orc_flat_search, q, k, [], threshold, filter_ids, cap)


class IVF(_Index):
    _free = "orc_ivf_free"
    _io = "ivf"

    def _count(self):
        return sum(self.list_sizes())

    def __init__(self, dim, metric, nlist):
        self.dim, self.metric, self.nlist, self.n = dim, metric, nlist, 0
        self.h = lib().orc_ivf_new(dim, METRIC[metric], nlist)

    def train(self, X):
        X = _f32(X)
        return lib().orc_ivf_train(self.h, _p(X), X.shape[0])

    def add(self, i, v):
        v = _f32(v)
        rc = lib().orc_ivf_add(self.h, int(i), _p(v))
        self.n += rc == 0
        return rc

    def add_batch(self, ids, X):
        for i, v in zip(ids, _f32(X)):
            rc = self.add(i, v)
            if rc:
                return rc
        return 0

    def remove(self, i):
        return lib().orc_ivf_remove(self.h, int(i))

    def capacity(self):
        return self.n

    def centroids(self):
        ptr = lib().orc_ivf_centroids(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.nlist, self.dim)).copy()

    def list_sizes(self):
        return [lib().orc_ivf_list_size(self.h, l) for l in range(self.nlist)]

    def search(self, q, k, nprobes, threshold=0.0, filter_ids=(), cap=None):
        return self._do_search(lib().orc_ivf_search, q, k, [int(nprobes)], threshold, filter_ids, cap)


class PQ(_Index):
    _free = "orc_pq_free"
    _io = "pq"

    def __init__(self, dim, metric, M, nbits):
        self.dim, self.metric, self.M, self.nbits = dim, metric, M, nbits
        self.ksub, self.dsub = 1 << nbits, dim // M
        self.h = lib().orc_pq_new(dim, METRIC[metric], M, nbits)

    def train(self, X):
        X = _f32(X)
        return lib().orc_pq_train(self.h, _p(X), X.shape[0])

    def add(self, i, v):
        v = _f32(v)
        return lib().orc_pq_add(self.h, int(i), _p(v))

    def add_batch(self, ids, X):
        for i, v in zip(ids, _f32(X)):
            rc = self.add(i, v)
            if rc:
                return rc
        return 0

    def remove(self, i):
        return lib().orc_pq_remove(self.h, int(i))

    def capacity(self):
        return lib().orc_pq_size(self.h)

    def codebooks(self):
        ptr = lib().orc_pq_codebooks(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.M, self.ksub, self.dsub)).copy()

    def codes(self):
        n = self.capacity()
        ptr = lib().orc_pq_codes(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n, self.M)).copy()

    def search(self, q, k, threshold=0.0, filter_ids=(), cap=None):
        return self._do_search(lib().orc_pq_search, q, k, [], threshold, filter_ids, cap)


class IVFPQ(_Index):
    _free = "orc_ivfpq_free"
    _io = "ivfpq"

    def _count(self):
        return sum(self.list_size(l) for l in range(self.nlist))

    def __init__(self, dim, metric, nlist, M, nbits):
        self.dim, self.metric, self.nlist, self.M, self.nbits, self.n = dim, metric, nlist, M, nbits, 0
        self.ksub, self.dsub = 1 << nbits, dim // M
        self.h = lib().orc_ivfpq_new(dim, METRIC[metric], nlist, M, nbits)

    def train(self, X):
        X = _f32(X)
        return lib().orc_ivfpq_train(self.h, _p(X), X.shape[0])

    def set_quantizers(self, centroids, codebooks):
        c, cb = _f32(centroids), _f32(codebooks)
        return lib().orc_ivfpq_set_quantizers(self.h, _p(c), _p(cb))

    def append_encoded(self, lst, ids, codes):
        ids = np.ascontiguousarray(ids, np.uint32)
        codes = np.ascontiguousarray(codes, np.uint8)
        self.n += ids.shape[0]
        return lib().orc_ivfpq_append_encoded(self.h, int(lst), ids.shape[0], _p(ids), _p(codes))

    def add(self, i, v):
        v = _f32(v)
        rc = lib().orc_ivfpq_add(self.h, int(i), _p(v))
        self.n += rc == 0
        return rc

    def add_batch(self, ids, X):
        for i, v in zip(ids, _f32(X)):
            rc = self.add(i, v)
            if rc:
                return rc
        return 0

    def remove(self, i):
        return lib().orc_ivfpq_remove(self.h, int(i))

    def capacity(self):
        return self.n

    def centroids(self):
        ptr = lib().orc_ivfpq_centroids(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.nlist, self.dim)).copy()

    def codebooks(self):
        ptr = lib().orc_ivfpq_codebooks(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.M, self.ksub, self.dsub)).copy()

    def list_size(self, l):
        return lib().orc_ivfpq_list_size(self.h, l)

    def list_codes(self, l):
        n = self.list_size(l)
        if n == 0:
            return np.zeros((0, self.M), np.uint8)
        ptr = lib().orc_ivfpq_list_codes(self.h, l)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n, self.M)).copy()

    def list_ids(self, l):
        n = self.list_size(l)
        if n == 0:
            return np.zeros(0, np.uint32)
        ptr = lib().orc_ivfpq_list_ids(self.h, l)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n,)).copy()

    def search(self, q, k, nprobes, threshold=0.0, filter_ids=(), cap=None):
        return self._do_search(lib().orc_ivfpq_search, q, k, [int(nprobes)], threshold, filter_ids, cap)


class BM25:
    def __init__(self):
        self.h = lib().orc_bm25_new()

    def __del__(self):
        try:
            if self.h:
                lib().orc_bm25_free(self.h)
                self.h = None
        except Exception:
            pass

    def add(self, doc_id, tokens):
        t = np.ascontiguousarray(list(tokens), np.uint32)
        return lib().orc_bm25_add(self.h, int(doc_id), _p(t), int(t.size))

    def remove(self, doc_id):
        return lib().orc_bm25_remove(self.h, int(doc_id))

    def flush(self):
        lib().orc_bm25_flush(self.h)

    def num_docs(self):
        return lib().orc_bm25_num_docs(self.h)

    def avg_doc_len(self):
        return lib().orc_bm25_avg_doc_len(self.h)

    def search(self, qtokens, k, filter_ids=(), cap=4096):
        q = np.ascontiguousarray(list(qtokens), np.uint32)
        flt = np.ascontiguousarray(list(filter_ids), np.uint32)
        ids = np.zeros(cap, np.uint32); sc = np.zeros(cap, np.float32); sc64 = np.zeros(cap, np.float64)
        n = lib().orc_bm25_search(self.h, _p(q), int(q.size), int(k), _p(flt) if flt.size else None, int(flt.size), _p(ids), _p(sc), _p(sc64), cap)
        m = min(n, cap)
        return n, ids[:m], sc[:m], sc64[:m]


class HNSW(_Index):
    _free = "orc_hnsw_free"
    _io = "hnsw"

    def __init__(self, dim, metric, m=0, efc=0, efs=0, seed=12345):
        self.dim, self.metric = dim, metric
        self.h = lib().orc_hnsw_new(dim, METRIC[metric], m, efc, efs, C.c_uint64(seed))

    def add(self, i, v, level=-1):
        v = _f32(v)
        return lib().orc_hnsw_add_with_level(self.h, int(i), _p(v), int(level))

    def add_batch(self, ids, X):
        for i, v in zip(ids, _f32(X)):
            rc = self.add(i, v)
            if rc:
                return rc
        return 0

    def remove(self, i):
        return lib().orc_hnsw_remove(self.h, int(i))

    def capacity(self):
        return lib().orc_hnsw_size(self.h)

    def max_level(self):
        return lib().orc_hnsw_max_level(self.h)

    def entry(self):
        return lib().orc_hnsw_entry(self.h)

    def stats(self, reset=True):
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_hnsw_stats(self.h, C.byref(a), C.byref(b), 1 if reset else 0)
        return a.value, b.value

    def export(self):
        slots, ne = C.c_int64(), C.c_int64()
        n = lib().orc_hnsw_export(self.h, None, None, None, None, None, C.byref(slots), C.byref(ne))
        ids = np.zeros(n, np.uint32); levels = np.zeros(n, np.int32); vecs = np.zeros((n, self.dim), np.float32)
        eoff = np.zeros(slots.value + 1, np.int64); edges = np.zeros(max(1, ne.value), np.uint32)
        lib().orc_hnsw_export(self.h, _p(ids), _p(levels), _p(vecs), _p(eoff), _p(edges), None, None)
        return ids, levels, vecs, eoff, edges[:ne.value]

    def search(self, q, k, ef=0, threshold=0.0, filter_ids=(), cap=None):
        return self._do_search(lib().orc_hnsw_search, q, k, [int(ef)], threshold, filter_ids, cap)
