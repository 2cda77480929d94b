"""Host-side mirror of comet's index / search interfaces over the C ABI.

Names and behaviour follow the reference's Go API so that the parity tests read like the reference's
own tests:

    idx = FlatIndex(ctx, 128, "l2_squared")            # comet.NewFlatIndex(128, comet.L2Squared)
    idx.add(node_id, vector)                             # idx.Add(*NewVectorNodeWithID(id, v))
    results = idx.new_search().with_query(q).with_k(10).execute()   # NewSearch().WithQuery().WithK().Execute()

`VectorIndex` is index.go:32-63, `VectorSearch` index_search.go:141-279. What stays on the host here is
exactly what stays in Go above the boundary: multi-query aggregation
(aggregation.go:107-255), LimitResults / Autocut (limiter.go), argument validation. All distance
arithmetic, list scans and top-k selection run in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import CometError, SearchParams, ZeroVectorError, check  # noqa: F401

EUCLIDEAN, L2_SQUARED, COSINE = "l2", "l2_squared", "cosine"          # DistanceKind distance.go:19-39
_METRIC = {EUCLIDEAN: _lib.L2, L2_SQUARED: _lib.L2SQ, COSINE: _lib.COSINE}
SUM_AGGREGATION, MAX_AGGREGATION, MEAN_AGGREGATION = "sum", "max", "mean"  # aggregation.go


class UnknownDistanceKind(ValueError):
    """comet.ErrUnknownDistanceKind (distance.go:9)"""


def _metric_code(kind: str) -> int:
    if kind not in _METRIC:
        raise UnknownDistanceKind("unknown distance kind")
    return _METRIC[kind]


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """One GPU: HIP device + stream + scratch arena (comet_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.comet_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.comet_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.comet_ctx_sync(self.h))

    @property
    def stream(self) -> int:
        return int(self.lib.comet_ctx_stream(self.h) or 0)

    def fence(self) -> None:
        """after the caller's OWN kernels on `stream` (lane 0): asynchronous searches enqueued afterwards start behind them, whichever lane they run on"""
        check(self.lib.comet_ctx_fence(self.h))

    # raw device memory (for callers that keep queries / results resident in HBM)
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(self.lib.comet_dev_alloc(self.h, int(nbytes), C.byref(p)))
        return int(p.value)

    def free(self, ptr: int):
        check(self.lib.comet_dev_free(self.h, C.c_void_p(ptr)))

    def upload(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        check(self.lib.comet_memcpy_h2d(self.h, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(self.lib.comet_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def synth_fill(self, ptr: int, seed: int, offset: int, n: int):
        check(self.lib.comet_synth_fill_dev(self.h, C.c_uint64(seed), C.c_uint64(offset), C.c_uint64(n), C.c_void_p(ptr)))

    def synth_mixture(self, ptr: int, seed: int, n_centers: int, sigma: float, n_sub: int, sigma_noise: float, row_base: int, n_rows: int, dim: int):
        check(self.lib.comet_synth_mixture_dev(self.h, C.c_uint64(seed), int(n_centers), C.c_float(sigma), int(n_sub), C.c_float(sigma_noise),
                                               C.c_uint64(row_base), C.c_uint64(n_rows), int(dim), C.c_void_p(ptr)))

    # profiling
    def profile(self, on: bool = True):
        check(self.lib.comet_profile_enable(self.h, 1 if on else 0))

    def set_lanes(self, lanes: int) -> None:
        """1 .. 8 execution lanes (default 8; Flat / IVF use two of them, PQ / IVFPQ four, HNSW eight): the asynchronous searches of an index rotate through up to that many streams"""
        check(self.lib.comet_ctx_set_lanes(self.h, int(lanes)))

    def profile_only(self, name: str | None) -> None:
        """time only the launches of one profiling scope (None: all of them)"""
        check(self.lib.comet_profile_only(self.h, (name or "").encode()))

    def profile_reset(self):
        check(self.lib.comet_profile_reset(self.h))

    def profile_get(self, prefix: str) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_int64()
        check(self.lib.comet_profile_get(self.h, prefix.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_dump(self) -> dict[str, tuple[float, int]]:
        buf = C.create_string_buffer(1 << 16)
        check(self.lib.comet_profile_dump(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, n = line.rsplit(" ", 2)
            out[name] = (float(ms), int(n))
        return out

    # comet.Distance singletons (distance.go:50-81) evaluated by the device kernels
    def distance(self, kind: str, a, b) -> float:
        a, b = _f32(a), _f32(b)
        out = C.c_float()
        check(self.lib.comet_distance(self.h, _metric_code(kind), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                      int(a.shape[0]), C.byref(out)))
        return np.float32(out.value)

    def distance_batch(self, kind: str, queries, target) -> np.ndarray:
        q, t = _f32(queries), _f32(target)
        out = np.empty(q.shape[0], dtype=np.float32)
        check(self.lib.comet_distance_batch(self.h, _metric_code(kind), q.ctypes.data_as(C.c_void_p), int(q.shape[0]),
                                            t.ctypes.data_as(C.c_void_p), int(t.shape[0]), out.ctypes.data_as(C.c_void_p)))
        return out

    def preprocess(self, kind: str, x) -> np.ndarray:
        x = _f32(x)
        out = np.empty_like(x)
        check(self.lib.comet_preprocess(self.h, _metric_code(kind), x.ctypes.data_as(C.c_void_p), int(x.shape[0]),
                                        out.ctypes.data_as(C.c_void_p)))
        return out

    # Norm / Normalize / Scale (distance.go:312-428): one vector or a batch of rows
    def _rows(self, x):
        x = _f32(x)
        return (x[None, :], True) if x.ndim == 1 else (x, False)

    def norm(self, x):
        X, one = self._rows(x)
        out = np.empty(X.shape[0], np.float32)
        check(self.lib.comet_norm_batch(self.h, X.ctypes.data_as(C.c_void_p), X.shape[0], X.shape[1], out.ctypes.data_as(C.c_void_p)))
        return out[0] if one else out

    def normalize(self, x) -> np.ndarray:
        X, one = self._rows(x)
        out = np.empty_like(X)
        check(self.lib.comet_normalize_batch(self.h, X.ctypes.data_as(C.c_void_p), X.shape[0], X.shape[1], out.ctypes.data_as(C.c_void_p)))
        return out[0] if one else out

    def scale(self, x, scalar: float) -> np.ndarray:
        X, one = self._rows(x)
        out = np.empty_like(X)
        check(self.lib.comet_scale_batch(self.h, X.ctypes.data_as(C.c_void_p), X.shape[0], X.shape[1], C.c_float(float(scalar)), out.ctypes.data_as(C.c_void_p)))
        return out[0] if one else out

    def kmeans(self, vectors, k: int, kind: str = L2_SQUARED, max_iter: int = 20):
        """KMeans / KMeansSubspace (clustering.go:60,112). Returns (centroids, assignments) or (None, None)."""
        v = _f32(vectors)
        if v.ndim != 2 or v.shape[0] == 0 or k <= 0:
            return None, None
        n, d = v.shape
        ke = min(k, n)
        cent = np.empty((ke, d), dtype=np.float32)
        assign = np.empty(n, dtype=np.int32)
        kout = C.c_int32()
        check(self.lib.comet_kmeans(self.h, v.ctypes.data_as(C.c_void_p), n, d, k, _metric_code(kind), max_iter,
                                    cent.ctypes.data_as(C.c_void_p), assign.ctypes.data_as(C.c_void_p), C.byref(kout)))
        return cent[:kout.value], assign

    def nearest_centroid(self, vectors, centroids, kind: str) -> np.ndarray:
        v, cts = _f32(vectors), _f32(centroids)
        if v.ndim == 1:
            v = v[None, :]
        out = np.empty(v.shape[0], dtype=np.int32)
        check(self.lib.comet_nearest_centroid(self.h, v.ctypes.data_as(C.c_void_p), v.shape[0], v.shape[1],
                                              cts.ctypes.data_as(C.c_void_p), cts.shape[0], _metric_code(kind),
                                              out.ctypes.data_as(C.c_void_p)))
        return out


@dataclass
class VectorResult:
    """VectorResult{Node, Score} index_search.go:84-90 (the node is identified by its id)."""
    id: int
    score: np.float32


# ---- host-side post-processing that the reference also runs on the host ----------------------------
def sanitize_k(k: int, max_results: int) -> int:           # limiter.go:12-17
    return max_results if (k <= 0 or k > max_results) else k


def autocut(y: Sequence[float], cutoff: int) -> int:        # limiter.go:81-118
    y = np.asarray(y, dtype=np.float32)
    n = len(y)
    if n <= 1:
        return n
    step = np.float32(1.0) / (np.float32(n) - np.float32(1.0))
    with np.errstate(all="ignore"):
        diff = [(y[i] - y[0]) / (y[n - 1] - y[0]) - (np.float32(0.0) + np.float32(i) * step) for i in range(n)]
    extrema = 0
    for i in range(1, n):
        if i == n - 1:
            if i - 2 < 0:
                continue
            hit = diff[i] > diff[i - 1] and diff[i] > diff[i - 2]
        else:
            hit = diff[i] > diff[i - 1] and diff[i] > diff[i + 1]
        if hit:
            extrema += 1
            if extrema >= cutoff:
                return i
    return n


def aggregate(results: list[VectorResult], kind: str) -> list[VectorResult]:   # aggregation.go:107-255
    if not results:
        return results
    order, scores = [], {}
    for r in results:
        if r.id not in scores:
            scores[r.id] = []
            order.append(r.id)
        scores[r.id].append(np.float32(r.score))
    out = []
    for i in order:
        s = scores[i]
        if kind == SUM_AGGREGATION:
            v = np.float32(0)
            for x in s:
                v = np.float32(v + x)
        elif kind == MAX_AGGREGATION:
            v = s[0]
            for x in s[1:]:
                if x > v:
                    v = x
        elif kind == MEAN_AGGREGATION:
            v = np.float32(0)
            for x in s:
                v = np.float32(v + x)
            v = np.float32(v / np.float32(len(s)))
        else:
            raise ValueError(f"unknown aggregation kind: {kind}")
        out.append(VectorResult(i, v))
    out.sort(key=lambda r: r.score)   # stable; the reference's order among ties is undefined
    return out


class VectorSearch:
    """Fluent search builder (VectorSearch, index_search.go:141-279)."""

    def __init__(self, index: "VectorIndex"):
        self.index = index
        self.queries: list[np.ndarray] = []
        self.node_ids: list[int] = []
        self.k = 10                      # defaults: flat_index.go:305-311
        self.nprobes = index.default_nprobes()
        self.ef_search = 0
        self.threshold = 0.0
        self.aggregation = ""
        self.cutoff = -1
        self.document_ids: list[int] = []
        self.mode = 0

    def with_query(self, *queries) -> "VectorSearch":
        self.queries = [np.asarray(q, dtype=np.float32) for q in queries]
        return self

    def with_node(self, *node_ids) -> "VectorSearch":
        self.node_ids = [int(i) for i in node_ids]
        return self

    def with_k(self, k: int) -> "VectorSearch":
        self.k = int(k)
        return self

    def with_n_probes(self, n: int) -> "VectorSearch":
        self.nprobes = int(n)
        return self

    def with_ef_search(self, ef: int) -> "VectorSearch":
        self.ef_search = int(ef)
        return self

    def with_threshold(self, t: float) -> "VectorSearch":
        self.threshold = float(t)
        return self

    def with_score_aggregation(self, kind: str) -> "VectorSearch":
        self.aggregation = kind
        return self

    def with_cutoff(self, cutoff: int) -> "VectorSearch":
        self.cutoff = int(cutoff)
        return self

    def with_document_ids(self, *ids) -> "VectorSearch":
        self.document_ids = [int(i) for i in ids]
        return self

    def with_mode(self, mode: int) -> "VectorSearch":
        """0 auto, 1 strict exact-arithmetic kernels, 2 force the fast path (backend knob, not in the reference)."""
        self.mode = int(mode)
        return self

    def execute(self) -> list[VectorResult]:
        # flat_index_search.go:109-165 (identical shells in the IVF / PQ / IVFPQ / HNSW searches)
        if not self.queries and not self.node_ids:
            raise ValueError("must specify either queries or node IDs")
        agg = self.aggregation or SUM_AGGREGATION
        if agg not in (SUM_AGGREGATION, MAX_AGGREGATION, MEAN_AGGREGATION):
            raise ValueError(f"unknown aggregation kind: {agg}")
        all_q = list(self.queries)
        if self.node_ids:
            all_q += self.index._lookup_node_vectors(self.node_ids)
        for q in all_q:
            if q.ndim != 1 or q.shape[0] != self.index.dim:
                raise ValueError(f"query dimension mismatch: expected {self.index.dim}, got {q.shape[0] if q.ndim == 1 else q.shape}")
        self.index._check_searchable()
        k_cap = self.index._k_cap(self.k, self.nprobes)
        ids, scores, counts = self.index.search_batch(np.stack(all_q), self.k, threshold=self.threshold, nprobes=self.nprobes,
                                                      ef_search=self.ef_search, document_ids=self.document_ids, k_cap=k_cap,
                                                      mode=self.mode)
        all_results = [VectorResult(int(ids[b, i]), np.float32(scores[b, i])) for b in range(len(all_q)) for i in range(min(counts[b], k_cap))]
        res = aggregate(all_results, agg)
        res = res[:sanitize_k(self.k, len(res))]                 # LimitResults limiter.go:28
        if self.cutoff != -1 and res:                            # AutocutResults limiter.go:52
            res = res[:autocut([r.score for r in res], self.cutoff)]
        return res


class VectorIndex:
    """Common part of the GPU indexes (VectorIndex, index.go:32-63)."""

    kind_name = "?"

    def __init__(self, ctx: Context, dim: int, distance_kind: str):
        self.ctx, self.lib = ctx, ctx.lib
        self.dim = int(dim)
        self.distance_kind_ = distance_kind
        self.h = C.c_void_p()

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.comet_index_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- VectorIndex surface --
    def dimensions(self) -> int:
        return int(self.lib.comet_index_dim(self.h))

    def distance_kind(self) -> str:
        return self.distance_kind_

    def kind(self) -> str:
        return self.kind_name

    def trained(self) -> bool:
        return bool(self.lib.comet_index_trained(self.h))

    def __len__(self) -> int:
        return int(self.lib.comet_index_size(self.h))

    def default_nprobes(self) -> int:
        return int(self.lib.comet_index_default_nprobes(self.h))

    def train(self, vectors) -> None:
        v = _f32(vectors)
        if v.ndim != 2 or (v.shape[0] and v.shape[1] != self.dim):
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {v.shape[-1] if v.size else 0}")
        check(self.lib.comet_index_train(self.h, v.ctypes.data_as(C.c_void_p), v.shape[0]))

    def train_dev(self, vecs_dev: int, n: int) -> None:
        """Train on n dense rows that already sit in device memory."""
        check(self.lib.comet_index_train_dev(self.h, C.c_void_p(vecs_dev), int(n)))

    def add_batch_dev(self, ids, vecs_dev: int, n: int) -> int:
        """Add n dense rows that already sit in device memory (ids from the host); returns the rows added."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        idbuf = self.ctx.alloc(max(4, ids.nbytes))
        try:
            self.ctx.upload(idbuf, ids)
            added = C.c_int64()
            check(self.lib.comet_index_add_dev(self.h, C.c_void_p(idbuf), C.c_void_p(vecs_dev), int(n), C.byref(added)))
        finally:
            self.ctx.free(idbuf)
        self.last_added = added.value
        return added.value

    def add(self, node_id: int, vector, write_back: bool = True) -> None:
        """VectorIndex.Add. Like the reference (flat_index.go:182) a cosine index normalises the caller's
        float32 array in place when `vector` is a writable float32 ndarray."""
        v = np.asarray(vector, dtype=np.float32)
        if v.ndim != 1 or v.shape[0] != self.dim:
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {v.shape[0] if v.ndim == 1 else v.shape}")
        out = self.add_batch(np.array([node_id], dtype=np.uint32), v[None, :], return_normalized=True)
        if write_back and isinstance(vector, np.ndarray) and vector.dtype == np.float32 and vector.flags.writeable:
            vector[...] = out[0]

    def add_batch(self, ids, vectors, return_normalized: bool = False):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        v = _f32(vectors)
        if v.ndim != 2 or v.shape[1] != self.dim:
            raise ValueError(f"vector dimension mismatch: expected {self.dim}, got {v.shape[-1]}")
        if ids.shape[0] != v.shape[0]:
            raise ValueError("ids / vectors length mismatch")
        added = C.c_int64()
        norm = np.empty_like(v) if return_normalized else None
        rc = self.lib.comet_index_add(self.h, ids.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), v.shape[0],
                                      C.byref(added), norm.ctypes.data_as(C.c_void_p) if norm is not None else None)
        self.last_added = added.value
        check(rc)
        return norm

    def remove(self, node_id: int) -> None:
        check(self.lib.comet_index_remove(self.h, C.c_uint32(int(node_id))))

    def flush(self) -> None:
        check(self.lib.comet_index_flush(self.h))

    def new_search(self) -> VectorSearch:
        return VectorSearch(self)

    # -- io.WriterTo / io.ReaderFrom (index.go:58-60): the reference's own byte layout --
    def write_to(self, w) -> int:
        """VectorIndex.WriteTo(w): flushes, then streams the reference's on-disk format into `w.write`. Returns bytes written."""
        err: list[BaseException] = []

        def cb(_user, data, n):
            try:
                w.write(C.string_at(data, n))
                return 0
            except BaseException as e:   # surfaced below
                err.append(e)
                return 1
        n = C.c_int64()
        rc = self.lib.comet_index_write_to(self.h, _lib.WRITE_CB(cb), None, C.byref(n))
        if err:
            raise err[0]
        check(rc)
        return n.value

    def read_from(self, r) -> int:
        """VectorIndex.ReadFrom(r): parses the reference's on-disk format from `r.read` (io.ReadFull semantics) and
        replaces this index's contents; the constructor parameters must match the stream. Returns bytes consumed."""
        err: list[BaseException] = []

        def cb(_user, dst, n):
            try:
                got = 0
                while got < n:
                    chunk = r.read(n - got)
                    if not chunk:
                        return 1          # io.ErrUnexpectedEOF
                    C.memmove(dst + got, chunk, len(chunk))
                    got += len(chunk)
                return 0
            except BaseException as e:
                err.append(e)
                return 1
        n = C.c_int64()
        rc = self.lib.comet_index_read_from(self.h, _lib.READ_CB(cb), None, C.byref(n))
        if err:
            raise err[0]
        check(rc)
        return n.value

    def to_bytes(self) -> bytes:
        import io
        b = io.BytesIO()
        self.write_to(b)
        return b.getvalue()

    def from_bytes(self, data: bytes) -> int:
        import io
        return self.read_from(io.BytesIO(data))

    # -- batched entry (one call = B independent queries) --
    def search_batch(self, queries, k: int, threshold: float = 0.0, nprobes: int = 0, ef_search: int = 0,
                     document_ids: Iterable[int] = (), k_cap: int | None = None, mode: int = 0):
        q = _f32(queries)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"query dimension mismatch: expected {self.dim}, got {q.shape[-1]}")
        B = q.shape[0]
        if k_cap is None:
            k_cap = self._k_cap(k, nprobes)
        flt = np.ascontiguousarray(list(document_ids), dtype=np.uint32)
        p = SearchParams(k=int(k), threshold=float(threshold), nprobes=int(nprobes), ef_search=int(ef_search),
                         filter_ids=flt.ctypes.data_as(C.POINTER(C.c_uint32)) if flt.size else None, n_filter=int(flt.size), mode=int(mode))
        ids = np.zeros((B, k_cap), dtype=np.uint32)
        scores = np.zeros((B, k_cap), dtype=np.float32)
        counts = np.zeros(B, dtype=np.int32)
        check(self.lib.comet_index_search(self.h, q.ctypes.data_as(C.c_void_p), B, C.byref(p), ids.ctypes.data_as(C.c_void_p),
                                          scores.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), k_cap))
        return ids, scores, counts

    def search_batch_dev(self, q_dev: int, B: int, k: int, out_ids_dev: int, out_scores_dev: int, out_counts_dev: int,
                         k_cap: int, threshold: float = 0.0, nprobes: int = 0, ef_search: int = 0, mode: int = 0):
        """Device-resident queries and outputs; asynchronous on the context's stream."""
        p = SearchParams(k=int(k), threshold=float(threshold), nprobes=int(nprobes), ef_search=int(ef_search),
                         filter_ids=None, n_filter=0, mode=int(mode))
        check(self.lib.comet_index_search_dev(self.h, C.c_void_p(q_dev), int(B), C.byref(p), C.c_void_p(out_ids_dev),
                                              C.c_void_p(out_scores_dev), C.c_void_p(out_counts_dev), int(k_cap)))

    def search_batch_dev_async(self, q_dev: int, B: int, k: int, out_ids_dev: int, out_scores_dev: int, out_counts_dev: int,
                               k_cap: int, threshold: float = 0.0, nprobes: int = 0, ef_search: int = 0, mode: int = 0) -> int:
        """Enqueue only; returns a ticket for search_wait(). Buffers must stay untouched until the wait returns."""
        p = SearchParams(k=int(k), threshold=float(threshold), nprobes=int(nprobes), ef_search=int(ef_search),
                         filter_ids=None, n_filter=0, mode=int(mode))
        t = C.c_uint64()
        check(self.lib.comet_index_search_dev_async(self.h, C.c_void_p(q_dev), int(B), C.byref(p), C.c_void_p(out_ids_dev),
                                                    C.c_void_p(out_scores_dev), C.c_void_p(out_counts_dev), int(k_cap), C.byref(t)))
        return t.value

    def search_wait(self, ticket: int) -> None:
        check(self.lib.comet_index_search_wait(self.h, C.c_uint64(int(ticket))))

    # -- helpers --
    def _k_cap(self, k: int, nprobes: int) -> int:
        n = len(self)
        return max(1, n if (k <= 0 or k > n) else k)

    def _check_searchable(self) -> None:
        pass

    def stat(self, name: str) -> float:
        out = C.c_double()
        check(self.lib.comet_index_get_stat(self.h, name.encode(), C.byref(out)))
        return out.value

    def export(self, codes_width: int = 0):
        """(ids, list index, PQ codes) of every stored element in Add order."""
        n = len(self)
        ids = np.empty(n, np.uint32); lists = np.empty(n, np.int32)
        codes = np.empty((n, codes_width), np.uint8) if codes_width else None
        check(self.lib.comet_index_export(self.h, ids.ctypes.data_as(C.c_void_p), lists.ctypes.data_as(C.c_void_p),
                                          codes.ctypes.data_as(C.c_void_p) if codes is not None else None))
        return ids, lists, codes

    def list_size(self, lst: int = 0) -> int:
        out = C.c_int64()
        check(self.lib.comet_index_list_size(self.h, int(lst), C.byref(out)))
        return out.value

    def list_read(self, lst: int = 0, codes_width: int = 0, want_vectors: bool = False):
        n = self.list_size(lst)
        ids = np.empty(n, dtype=np.uint32)
        codes = np.empty((n, codes_width), dtype=np.uint8) if codes_width else None
        vecs = np.empty((n, self.dim), dtype=np.float32) if want_vectors else None
        check(self.lib.comet_index_list_read(self.h, int(lst), ids.ctypes.data_as(C.c_void_p),
                                             codes.ctypes.data_as(C.c_void_p) if codes is not None else None,
                                             vecs.ctypes.data_as(C.c_void_p) if vecs is not None else None))
        return ids, codes, vecs

    def _lookup_node_vectors(self, node_ids: list[int]) -> list[np.ndarray]:
        # lookupNodeVectors flat_index_search.go:171-196 (+ the IVF / HNSW siblings): stored vectors by node id
        ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        out = np.empty((len(ids), self.dim), dtype=np.float32)
        rc = self.lib.comet_index_fetch_vectors(self.h, ids.ctypes.data_as(C.c_void_p), len(ids), out.ctypes.data_as(C.c_void_p))
        if rc == _lib.ERR_NOT_FOUND:
            raise KeyError(self.lib.comet_last_error().decode())
        check(rc)
        return [out[i].copy() for i in range(len(ids))]


class SegmentSet:
    """The vector indexes of a persistent store's memtables / segments (storage.go:489-626), oldest first, all resident in HBM:
    one call searches every one of them for a batch of queries and merges on the device (comet_segments_search: highest score per
    id, scores DESCENDING, cut to k — mergeResults / sortResultsByScore storage_merge.go:13-54)."""

    def __init__(self, indexes: Sequence["VectorIndex"]):
        self.indexes = list(indexes)
        if not self.indexes:
            raise ValueError("no segments")
        self.lib = self.indexes[0].lib
        self.dim = self.indexes[0].dim
        self._handles = (C.c_void_p * len(self.indexes))(*[ix.h for ix in self.indexes])

    def _params(self, k, threshold, nprobes, ef_search, flt, mode):
        return SearchParams(k=int(k), threshold=float(threshold), nprobes=int(nprobes), ef_search=int(ef_search),
                            filter_ids=flt.ctypes.data_as(C.POINTER(C.c_uint32)) if flt is not None and flt.size else None,
                            n_filter=int(flt.size) if flt is not None else 0, mode=int(mode))

    def search_batch(self, queries, k: int, threshold: float = 0.0, nprobes: int = 0, ef_search: int = 0,
                     document_ids: Iterable[int] = (), mode: int = 0):
        q = _f32(queries)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"query dimension mismatch: expected {self.dim}, got {q.shape[-1]}")
        B, k_cap = q.shape[0], max(1, int(k))
        flt = np.ascontiguousarray(list(document_ids), dtype=np.uint32)
        p = self._params(k, threshold, nprobes, ef_search, flt, mode)
        ids = np.zeros((B, k_cap), dtype=np.uint32); scores = np.zeros((B, k_cap), dtype=np.float32); counts = np.zeros(B, dtype=np.int32)
        check(self.lib.comet_segments_search(self._handles, len(self.indexes), q.ctypes.data_as(C.c_void_p), B, C.byref(p),
                                             ids.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), k_cap))
        return ids, scores, counts

    def search_batch_dev(self, q_dev: int, B: int, k: int, out_ids_dev: int, out_scores_dev: int, out_counts_dev: int, k_cap: int,
                         threshold: float = 0.0, nprobes: int = 0, ef_search: int = 0, mode: int = 0) -> None:
        """Device-resident queries and outputs; asynchronous on the context's stream apart from the per-segment finish."""
        p = self._params(k, threshold, nprobes, ef_search, None, mode)
        check(self.lib.comet_segments_search_dev(self._handles, len(self.indexes), C.c_void_p(q_dev), int(B), C.byref(p), C.c_void_p(out_ids_dev),
                                                 C.c_void_p(out_scores_dev), C.c_void_p(out_counts_dev), int(k_cap)))


class FlatIndex(VectorIndex):
    """comet.NewFlatIndex(dim, distanceKind) — flat_index.go:127."""
    kind_name = "flat"

    def __init__(self, ctx: Context, dim: int, distance_kind: str):
        if dim <= 0:
            raise ValueError("dimension must be positive")
        super().__init__(ctx, dim, distance_kind)
        check(self.lib.comet_flat_create(ctx.h, dim, _metric_code(distance_kind), C.byref(self.h)))
        # (soft deletes live in the library only — the device-side bitmap a ReadFrom loads included; no host mirror to fall out of step)


class _TrainedIndex(VectorIndex):
    not_trained_msg = "index must be trained before searching"

    def _check_searchable(self) -> None:
        if not self.trained():
            raise RuntimeError(self.not_trained_msg)

    def set_shard(self, rank: int, world: int) -> None:
        """Multi-GPU list sharding (IVF / IVFPQ): keep only the members of the lists this rank owns (before the first add). Lists are dealt by their
        training-set lengths (the same on every rank that trained); a rank that loaded its quantisers instead needs `set_list_owners`."""
        check(self.lib.comet_index_set_shard(self.h, int(rank), int(world)))

    def list_owners(self, nlist: int) -> np.ndarray:
        """owner rank of every inverted list: host-side state a sharded checkpoint has to carry beside the shard files"""
        out = np.zeros(nlist, dtype=np.int32)
        check(self.lib.comet_index_get_list_owners(self.h, out.ctypes.data_as(C.c_void_p), int(nlist)))
        return out

    def set_list_owners(self, owners) -> None:
        o = np.ascontiguousarray(owners, dtype=np.int32)
        check(self.lib.comet_index_set_list_owners(self.h, o.ctypes.data_as(C.c_void_p), int(len(o))))

    def centroids(self, nlist: int) -> np.ndarray:
        out = np.empty((nlist, self.dim), dtype=np.float32)
        check(self.lib.comet_index_get_centroids(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def codebooks(self, M: int, ksub: int, dsub: int) -> np.ndarray:
        out = np.empty((M, ksub, dsub), dtype=np.float32)
        check(self.lib.comet_index_get_codebooks(self.h, out.ctypes.data_as(C.c_void_p)))
        return out


class IVFIndex(_TrainedIndex):
    """comet.NewIVFIndex(dim, nlist, distanceKind) — ivf_index.go:147 (nlist comes BEFORE the distance kind there)."""
    kind_name = "ivf"

    def __init__(self, ctx: Context, dim: int, nlist: int, distance_kind: str):
        if dim <= 0:
            raise ValueError("dimension must be positive")
        if nlist <= 0:
            raise ValueError("nlist must be positive")
        super().__init__(ctx, dim, distance_kind)
        self.nlist = nlist
        check(self.lib.comet_ivf_create(ctx.h, dim, _metric_code(distance_kind), nlist, C.byref(self.h)))

    def _k_cap(self, k, nprobes):
        return max(1, len(self) if (k <= 0 or k > len(self)) else k)


class PQIndex(_TrainedIndex):
    """comet.NewPQIndex(dim, distanceKind, M, Nbits) — pq_index.go:135."""
    kind_name = "pq"
    not_trained_msg = "index not trained"     # pq_index_search.go:224

    def __init__(self, ctx: Context, dim: int, distance_kind: str, M: int, nbits: int):
        _validate_pq(dim, M, nbits)
        super().__init__(ctx, dim, distance_kind)
        self.M, self.nbits, self.ksub, self.dsub = M, nbits, 1 << nbits, dim // M
        check(self.lib.comet_pq_create(ctx.h, dim, _metric_code(distance_kind), M, nbits, C.byref(self.h)))


class IVFPQIndex(_TrainedIndex):
    """comet.NewIVFPQIndex(dim, distanceKind, nlist, m, nbits) — ivfpq_index.go:113."""
    kind_name = "ivfpq"

    def __init__(self, ctx: Context, dim: int, distance_kind: str, nlist: int, M: int, nbits: int):
        if dim <= 0:
            raise ValueError("dimension must be positive")
        if nlist <= 0:
            raise ValueError("nlist must be positive")
        _validate_pq(dim, M, nbits)
        super().__init__(ctx, dim, distance_kind)
        self.nlist, self.M, self.nbits, self.ksub, self.dsub = nlist, M, nbits, 1 << nbits, dim // M
        check(self.lib.comet_ivfpq_create(ctx.h, dim, _metric_code(distance_kind), nlist, M, nbits, C.byref(self.h)))


def _validate_pq(dim: int, M: int, nbits: int) -> None:   # pq_index.go:135-155
    if dim <= 0:
        raise ValueError("dimension must be positive")
    if M <= 0:
        raise ValueError("parameter M must be positive")
    if dim % M != 0:
        raise ValueError(f"dimension {dim} must be divisible by M {M}")
    if nbits <= 0 or nbits > 16:
        raise ValueError("parameter Nbits must be in [1,16]")


def default_nprobes(nlist: int) -> int:    # ivf_index.go:406-413
    return int(math.sqrt(float(nlist)))


# ---------------------------------------------------------------------------------------------------
# BM25 text index (TextIndex index.go:65-81; TextSearch index_search.go:358-430)
# ---------------------------------------------------------------------------------------------------
@dataclass
class TextResult:
    """TextResult{Id, Score} index_search.go:306-312"""
    id: int
    score: np.float32


def aggregate_text(results: list[TextResult], kind: str) -> list[TextResult]:
    """Text aggregation (aggregation.go: text*Aggregation): by doc id, sorted by score DESCENDING."""
    if not results:
        return results
    order, scores = [], {}
    for r in results:
        if r.id not in scores:
            scores[r.id] = []
            order.append(r.id)
        scores[r.id].append(np.float32(r.score))
    out = []
    for i in order:
        s = scores[i]
        if kind == SUM_AGGREGATION:
            v = np.float32(0)
            for x in s:
                v = np.float32(v + x)
        elif kind == MAX_AGGREGATION:
            v = max(s)
        elif kind == MEAN_AGGREGATION:
            v = np.float32(0)
            for x in s:
                v = np.float32(v + x)
            v = np.float32(v / np.float32(len(s)))
        else:
            raise ValueError(f"unknown aggregation kind: {kind}")
        out.append(TextResult(i, v))
    out.sort(key=lambda r: -float(r.score))
    return out


class DocTokenStore:
    """docTokens / deletedDocs of the reference's BM25 index (bm25_index.go: Add stores a document's tokens, Remove soft-deletes, Flush drops) — the host-side state
    WithNode needs (lookupNodeTexts bm25_index_search.go:233-260: a document's own tokens become a query). Mixed into BM25SearchIndex; the device never sees it."""

    def _dt_init(self):
        self._doc_tokens: dict[int, np.ndarray] = {}
        self._deleted_docs: set[int] = set()

    def _dt_add(self, doc_id: int, tokens) -> None:
        self._doc_tokens[int(doc_id)] = np.array(tokens, dtype=np.uint32)              # (a copy: one small array per document, no per-token Python objects)
        self._deleted_docs.discard(int(doc_id))

    def _dt_remove(self, doc_id: int) -> None:
        if int(doc_id) in self._doc_tokens:
            self._deleted_docs.add(int(doc_id))

    def _dt_flush(self) -> None:
        for d in self._deleted_docs:
            self._doc_tokens.pop(d, None)
        self._deleted_docs.clear()

    def _lookup_node_tokens(self, node_ids) -> list[list[int]]:
        out = []
        for i in node_ids:
            if int(i) in self._deleted_docs:
                raise KeyError(f"node ID {int(i)} not found in index (deleted)")
            if int(i) not in self._doc_tokens:
                raise KeyError(f"node ID {int(i)} not found in index")
            out.append([int(t) for t in self._doc_tokens[int(i)]])
        return out


class TextSearch:
    """Fluent text search builder (TextSearch, index_search.go:358-430) over token-id queries."""

    def __init__(self, index: "BM25SearchIndex"):
        self.index = index
        self.queries: list[list[int]] = []
        self.node_ids: list[int] = []
        self.k = 10
        self.aggregation = ""
        self.cutoff = -1
        self.document_ids: list[int] = []

    def with_query(self, *token_lists) -> "TextSearch":
        self.queries = [list(map(int, t)) for t in token_lists]
        return self

    def with_node(self, *node_ids) -> "TextSearch":
        """WithNode: the stored tokens of these documents are searched for as queries of their own, behind the direct queries (bm25_index_search.go:63-66,194-210)"""
        self.node_ids = [int(i) for i in node_ids]
        return self

    def with_k(self, k: int) -> "TextSearch":
        self.k = int(k)
        return self

    def with_score_aggregation(self, kind: str) -> "TextSearch":
        self.aggregation = kind
        return self

    def with_cutoff(self, cutoff: int) -> "TextSearch":
        self.cutoff = int(cutoff)
        return self

    def with_document_ids(self, *ids) -> "TextSearch":
        self.document_ids = [int(i) for i in ids]
        return self

    def execute(self) -> list[TextResult]:
        if not self.queries and not self.node_ids:
            raise ValueError("must specify either queries or node IDs")
        agg = self.aggregation or SUM_AGGREGATION
        all_q = list(self.queries)
        if self.node_ids:
            all_q += self.index._lookup_node_tokens(self.node_ids)
        nd = max(1, self.index.num_docs())
        k_cap = max(1, min(nd, self.k if self.k > 0 else nd))
        ids, sc, _, cnt = self.index.search_batch(all_q, self.k, document_ids=self.document_ids, k_cap=k_cap)
        allr = [TextResult(int(ids[b, i]), np.float32(sc[b, i])) for b in range(len(all_q)) for i in range(min(cnt[b], k_cap))]
        res = aggregate_text(allr, agg)
        res = res[:sanitize_k(self.k, len(res))]
        if self.cutoff != -1 and res:
            res = res[:autocut([r.score for r in res], self.cutoff)]
        return res


class BM25SearchIndex(DocTokenStore):
    """comet.NewBM25SearchIndex() with documents / queries given as token ids (tokenisation stays in Go)."""

    def __init__(self, ctx: Context):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        self._dt_init()
        check(self.lib.comet_bm25_create(ctx.h, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.comet_bm25_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, doc_id: int, tokens) -> None:
        t = np.ascontiguousarray(list(tokens), dtype=np.uint32)
        check(self.lib.comet_bm25_add(self.h, C.c_uint32(int(doc_id)), t.ctypes.data_as(C.c_void_p), int(t.size)))
        self._dt_add(doc_id, t)

    def remove(self, doc_id: int) -> None:
        check(self.lib.comet_bm25_remove(self.h, C.c_uint32(int(doc_id))))
        self._dt_remove(doc_id)

    def flush(self) -> None:
        check(self.lib.comet_bm25_flush(self.h))
        self._dt_flush()

    def num_docs(self) -> int:
        return int(self.lib.comet_bm25_num_docs(self.h))

    def avg_doc_len(self) -> float:
        return float(self.lib.comet_bm25_avg_doc_len(self.h))

    def new_search(self) -> TextSearch:
        return TextSearch(self)

    def search_batch(self, queries, k: int, document_ids: Iterable[int] = (), k_cap: int | None = None):
        B = len(queries)
        nd = max(1, self.num_docs())
        k_cap = k_cap or max(1, min(nd, k if k > 0 else nd))
        offs = np.zeros(B + 1, dtype=np.int32)
        for i, qt in enumerate(queries):
            offs[i + 1] = offs[i] + len(qt)
        toks = np.ascontiguousarray([t for qt in queries for t in qt], dtype=np.uint32)
        flt = np.ascontiguousarray(list(document_ids), dtype=np.uint32)
        ids = np.zeros((B, k_cap), np.uint32)
        sc = np.zeros((B, k_cap), np.float32)
        sc64 = np.zeros((B, k_cap), np.float64)
        cnt = np.zeros(B, np.int32)
        check(self.lib.comet_bm25_search(self.h, toks.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), B, int(k),
                                         flt.ctypes.data_as(C.c_void_p) if flt.size else None, int(flt.size), ids.ctypes.data_as(C.c_void_p),
                                         sc.ctypes.data_as(C.c_void_p), sc64.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), k_cap))
        return ids, sc, sc64, cnt


class HNSWIndex(VectorIndex):
    """comet.NewHNSWIndex(dim, distanceKind, m, efConstruction, efSearch) — hnsw_index.go:160.
    Search runs on the GPU; the graph is loaded (built by the reference / the oracle), see load_graph."""
    kind_name = "hnsw"

    def __init__(self, ctx: Context, dim: int, distance_kind: str, m: int = 0, ef_construction: int = 0, ef_search: int = 0):
        if dim <= 0:
            raise ValueError("dimension must be positive")
        super().__init__(ctx, dim, distance_kind)
        check(self.lib.comet_hnsw_create(ctx.h, dim, _metric_code(distance_kind), m, ef_construction, ef_search, C.byref(self.h)))

    def load_graph(self, ids, levels, vectors, edge_offsets, edges, entry_id: int, max_level: int) -> None:
        ids = np.ascontiguousarray(ids, np.uint32)
        levels = np.ascontiguousarray(levels, np.int32)
        v = _f32(vectors)
        eo = np.ascontiguousarray(edge_offsets, np.int64)
        ed = np.ascontiguousarray(edges, np.uint32)
        check(self.lib.comet_hnsw_load_graph(self.h, ids.shape[0], ids.ctypes.data_as(C.c_void_p), levels.ctypes.data_as(C.c_void_p),
                                             v.ctypes.data_as(C.c_void_p), eo.ctypes.data_as(C.c_void_p), ed.ctypes.data_as(C.c_void_p),
                                             C.c_uint32(int(entry_id)), int(max_level)))

    def add_with_levels(self, ids, vectors, levels) -> None:
        """Add(): insertNode on the GPU for every vector in order, with the given hnswNode levels (the reference draws them from an
        unseeded RNG; plain add()/add_batch() draws them from the index's own seeded stream)."""
        ids = np.ascontiguousarray(ids, np.uint32); v = _f32(vectors); lv = np.ascontiguousarray(levels, np.int32)
        if v.ndim != 2 or v.shape[1] != self.dim or ids.shape[0] != v.shape[0] or lv.shape[0] != v.shape[0]:
            raise ValueError("ids / vectors / levels shape mismatch")
        added = C.c_int64()
        check(self.lib.comet_hnsw_add_with_levels(self.h, ids.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), lv.ctypes.data_as(C.c_void_p),
                                                  v.shape[0], C.byref(added)))

    def set_level_seed(self, seed: int) -> None:
        check(self.lib.comet_hnsw_set_level_seed(self.h, C.c_uint64(int(seed))))

    def export_graph(self):
        """(ids, levels, vectors, edge_offsets, edges, entry_id, max_level) — the arrays load_graph takes."""
        n, slots, ne = C.c_int64(), C.c_int64(), C.c_int64()
        ent, ml = C.c_uint32(), C.c_int32()
        check(self.lib.comet_hnsw_export_graph(self.h, C.byref(n), C.byref(slots), C.byref(ne), None, None, None, None, None, C.byref(ent), C.byref(ml)))
        ids = np.zeros(n.value, np.uint32); levels = np.zeros(n.value, np.int32); vecs = np.zeros((n.value, self.dim), np.float32)
        eoff = np.zeros(slots.value + 1, np.int64); edges = np.zeros(max(1, ne.value), np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self.lib.comet_hnsw_export_graph(self.h, None, None, None, p(ids), p(levels), p(vecs), p(eoff), p(edges), C.byref(ent), C.byref(ml)))
        return ids, levels, vecs, eoff, edges[:ne.value], ent.value, ml.value

    def _k_cap(self, k, nprobes):
        n = len(self)
        return max(1, n if (k <= 0 or k > n) else k)
