"""Hybrid search above the boundary (config 5: IVF + BM25 + Reciprocal Rank Fusion).

This is orchestration that stays on the host in the reference too (hybrid_search_index.go:477-615,
fusion.go:174-243): both sub-searches run on the GPU through their index objects, are cut to k BEFORE fusion
(:518, :555), and the fused scores are sorted descending and cut to k. Metadata filtering (roaring / BSI) is out of
scope; pre-computed candidate ids can be passed with `with_document_ids` and are pushed down to both sub-searches
exactly like the reference pushes metadata candidates (:531-533, :560-562).

Ties: the reference ranks with an exchange sort over Go-map iteration order, so equal scores have no defined rank;
here insertion order (best-first lists from the sub-searches) is kept.
"""
from __future__ import annotations

from dataclasses import dataclass

RECIPROCAL_RANK_FUSION, WEIGHTED_SUM_FUSION, MAX_FUSION, MIN_FUSION = "reciprocal_rank", "weighted_sum", "max", "min"


@dataclass
class HybridSearchResult:
    id: int
    score: float


def score_map_to_ranks(scores: dict[int, float], ascending: bool) -> dict[int, int]:
    """scoreMapToRanks fusion.go:205-243 (O(n^2) exchange sort; 0-based ranks)."""
    s = list(scores.items())
    n = len(s)
    for i in range(n - 1):
        for j in range(i + 1, n):
            swap = s[i][1] > s[j][1] if ascending else s[i][1] < s[j][1]
            if swap:
                s[i], s[j] = s[j], s[i]
    return {doc: rank for rank, (doc, _) in enumerate(s)}


def reciprocal_rank_fusion(vector: dict[int, float], text: dict[int, float], k: float = 60.0) -> dict[int, float]:
    """reciprocalRankFusion.Combine fusion.go:174-203: vector ranks ascending (distances), text descending."""
    out: dict[int, float] = {}
    for doc, r in score_map_to_ranks(vector, True).items():
        out[doc] = 1.0 / (k + float(r))
    for doc, r in score_map_to_ranks(text, False).items():
        v = 1.0 / (k + float(r))
        out[doc] = out[doc] + v if doc in out else v
    return out


def reciprocal_rank_fusion_batch(v_ids, v_counts, t_ids, t_counts, k_out: int, k: float = 60.0):
    """reciprocalRankFusion.Combine + the descending sort and the cut to k of hybridSearch.Execute (fusion.go:174-203,
    hybrid_search_index.go:603-611) for a whole BATCH of queries at once, vectorised: v_ids[b, :v_counts[b]] are query b's vector hits
    best first (ascending distance), t_ids[b, :t_counts[b]] its text hits best first (descending relevance) — a hit's 0-based rank is its
    position, exactly what score_map_to_ranks() gives for lists that arrive sorted. Same float64 arithmetic in the same order as the
    per-query form (vector term first, text term added), ties in insertion order (vector hits, then text-only hits).
    Returns (ids uint32 [B, k_out], scores float64 [B, k_out], counts int32 [B])."""
    import numpy as np
    v_ids = np.asarray(v_ids); t_ids = np.asarray(t_ids)
    v_counts = np.asarray(v_counts, np.int64); t_counts = np.asarray(t_counts, np.int64)
    B, kv = v_ids.shape
    kt = t_ids.shape[1]
    if kv == 0 or kt == 0:       # a leg without columns (k = 0 on that side): the other leg alone, by the same rule (argmax over an empty axis raises)
        one_ids, one_cnt, kk = (t_ids, t_counts, kt) if kv == 0 else (v_ids, v_counts, kv)
        r1 = 1.0 / (k + np.arange(kk, dtype=np.float64))
        ok = np.arange(kk)[None, :] < one_cnt[:, None]
        sc = np.where(ok, r1[None, :], -np.inf) if kk else np.zeros((B, 0))
        order = np.argsort(-sc, axis=1, kind="stable")[:, :k_out]
        return (np.take_along_axis(one_ids, order, 1).astype(np.uint32), np.take_along_axis(sc, order, 1), np.minimum(k_out, ok.sum(1)).astype(np.int32))
    rv = 1.0 / (k + np.arange(kv, dtype=np.float64))
    rt = 1.0 / (k + np.arange(kt, dtype=np.float64))
    v_ok = np.arange(kv)[None, :] < v_counts[:, None]
    t_ok = np.arange(kt)[None, :] < t_counts[:, None]
    eq = (v_ids[:, :, None] == t_ids[:, None, :]) & v_ok[:, :, None] & t_ok[:, None, :]         # [B, kv, kt]; ids are unique inside a list
    hit_t = eq.argmax(2); has_t = eq.any(2)
    sv = np.where(has_t, rv[None, :] + rt[hit_t], rv[None, :])                                  # existing + rrfScore
    t_only = t_ok & ~eq.any(1)
    ids = np.concatenate([v_ids, t_ids], axis=1)
    sc = np.concatenate([np.where(v_ok, sv, -np.inf), np.where(t_only, rt[None, :], -np.inf)], axis=1)
    order = np.argsort(-sc, axis=1, kind="stable")[:, :k_out]
    out_ids = np.take_along_axis(ids, order, 1).astype(np.uint32)
    out_sc = np.take_along_axis(sc, order, 1)
    counts = np.minimum(k_out, v_ok.sum(1) + t_only.sum(1)).astype(np.int32)
    return out_ids, out_sc, counts


def weighted_sum_fusion(vector: dict[int, float], text: dict[int, float], wv: float = 1.0, wt: float = 1.0) -> dict[int, float]:
    """weightedSumFusion.Combine (the reference's default, hybrid_search_index.go:237): wv*vector + wt*text."""
    out = {d: wv * s for d, s in vector.items()}
    for d, s in text.items():
        out[d] = out.get(d, 0.0) + wt * s
    return out


def max_fusion(vector: dict[int, float], text: dict[int, float]) -> dict[int, float]:
    """maxFusion.Combine (fusion.go:252-271): the union of the two maps, the larger score where a document is in both."""
    out = dict(vector)
    for d, s in text.items():
        out[d] = max(out[d], s) if d in out else s
    return out


def min_fusion(vector: dict[int, float], text: dict[int, float]) -> dict[int, float]:
    """minFusion.Combine (fusion.go:291-306): only documents present in BOTH maps, the smaller score."""
    return {d: min(s, text[d]) for d, s in vector.items() if d in text}


class HybridSearch:
    """hybridSearch builder (hybrid_search_index.go:326-365) over a GPU vector index and a GPU BM25 index."""

    def __init__(self, vector_index=None, text_index=None):
        self.vector_index, self.text_index = vector_index, text_index
        self.vector_query = None
        self.text_queries: list[list[int]] = []
        self.k = 10
        self.n_probes = 1            # hybrid default nProbes = 1 (hybrid_search_index.go:236)
        self.ef_search = 0
        self.threshold = 0.0
        self.fusion_kind = WEIGHTED_SUM_FUSION
        self.rrf_k = 60.0
        self.vector_weight, self.text_weight = 1.0, 1.0          # DefaultFusionConfig (fusion.go)
        self.score_aggregation = "sum"   # SumAggregation, cutoff -1: hybrid_search_index.go:230-239
        self.cutoff = -1
        self.document_ids: list[int] = []

    def with_vector(self, q): self.vector_query = q; return self
    def with_text(self, *token_lists): self.text_queries = [list(t) for t in token_lists]; return self
    def with_k(self, k): self.k = int(k); return self
    def with_n_probes(self, n): self.n_probes = int(n); return self
    def with_ef_search(self, ef): self.ef_search = int(ef); return self
    def with_threshold(self, t): self.threshold = float(t); return self
    def with_fusion_kind(self, kind, rrf_k: float = 60.0): self.fusion_kind = kind; self.rrf_k = rrf_k; return self
    def with_document_ids(self, *ids): self.document_ids = [int(i) for i in ids]; return self
    def with_score_aggregation(self, kind): self.score_aggregation = kind; return self      # handed to both sub-searches (:512, :549)
    def with_cutoff(self, cutoff): self.cutoff = int(cutoff); return self                   # Autocut inside both sub-searches (:513, :550)

    def with_fusion(self, kind, vector_weight: float = 1.0, text_weight: float = 1.0, rrf_k: float = 60.0):
        """WithFusion(NewFusion(kind, &FusionConfig{VectorWeight, TextWeight, K})) (hybrid_search_index.go:456-459, fusion.go:86-103)"""
        if kind not in (RECIPROCAL_RANK_FUSION, WEIGHTED_SUM_FUSION, MAX_FUSION, MIN_FUSION):
            raise ValueError(f"unknown fusion kind: {kind}")
        self.fusion_kind, self.vector_weight, self.text_weight, self.rrf_k = kind, float(vector_weight), float(text_weight), float(rrf_k)
        return self

    def execute(self) -> list[HybridSearchResult]:
        vres: dict[int, float] = {}
        tres: dict[int, float] = {}
        if self.vector_query is not None:
            if self.vector_index is None:
                raise ValueError("vector query specified but no vector index configured")
            s = self.vector_index.new_search().with_query(self.vector_query).with_k(self.k).with_score_aggregation(self.score_aggregation).with_cutoff(self.cutoff)
            if self.n_probes > 0:
                s = s.with_n_probes(self.n_probes)
            if self.ef_search > 0:
                s = s.with_ef_search(self.ef_search)
            if self.threshold > 0:
                s = s.with_threshold(self.threshold)
            if self.document_ids:
                s = s.with_document_ids(*self.document_ids)
            vres = {r.id: float(r.score) for r in s.execute()}
        if self.text_queries:
            if self.text_index is None:
                raise ValueError("text query specified but no text index configured")
            s = self.text_index.new_search().with_query(*self.text_queries).with_k(self.k).with_score_aggregation(self.score_aggregation).with_cutoff(self.cutoff)
            if self.document_ids:
                s = s.with_document_ids(*self.document_ids)
            tres = {r.id: float(r.score) for r in s.execute()}
        if vres and tres:
            if self.fusion_kind == RECIPROCAL_RANK_FUSION:
                comb = reciprocal_rank_fusion(vres, tres, self.rrf_k)
            elif self.fusion_kind == WEIGHTED_SUM_FUSION:
                comb = weighted_sum_fusion(vres, tres, self.vector_weight, self.text_weight)
            elif self.fusion_kind == MAX_FUSION:
                comb = max_fusion(vres, tres)
            elif self.fusion_kind == MIN_FUSION:
                comb = min_fusion(vres, tres)
            else:
                raise ValueError(f"unknown fusion kind: {self.fusion_kind}")
        else:
            comb = vres or tres
        if not comb and self.document_ids:
            comb = {i: 1.0 for i in self.document_ids}
        res = [HybridSearchResult(i, s) for i, s in comb.items()]
        res.sort(key=lambda r: -r.score)      # sort.Slice(desc) hybrid_search_index.go:603
        return res[:self.k] if len(res) > self.k else res


def hybrid_rrf_search_batch(vector_index, text_index, queries, token_lists, k: int = 10, n_probes: int = 1, ef_search: int = 0, rrf_k: float = 60.0):
    """B hybrid searches with Reciprocal Rank Fusion in ONE call on the device (comet_hybrid_rrf_search): the vector leg on a second execution lane beside
    the text leg, the fusion one wave per query, one block of results back — what B times HybridSearch(...).with_fusion_kind(RECIPROCAL_RANK_FUSION).execute()
    returns (hybrid_search_index.go:477-615). queries: B x dim float32; token_lists: B lists of token ids. Returns (ids uint32 [B, k], scores float64 [B, k], counts)."""
    import ctypes as C
    import numpy as np
    from ._lib import check
    Q = np.ascontiguousarray(queries, dtype=np.float32)
    B = len(Q)
    if isinstance(token_lists, tuple) and len(token_lists) == 2 and isinstance(token_lists[0], np.ndarray):
        toks, offs = token_lists               # already flat: (uint32 tokens, int32 offsets[B + 1]) — what a Go caller hands the C ABI
        toks = np.ascontiguousarray(toks, dtype=np.uint32); offs = np.ascontiguousarray(offs, dtype=np.int32)
        if len(offs) != B + 1:
            raise ValueError("one token list per vector query")
    else:
        if B != len(token_lists):
            raise ValueError("one token list per vector query")
        offs = np.zeros(B + 1, dtype=np.int32)
        for i, qt in enumerate(token_lists):
            offs[i + 1] = offs[i] + len(qt)
        toks = np.ascontiguousarray([t for qt in token_lists for t in qt], dtype=np.uint32)
    ids = np.zeros((B, k), np.uint32); sc = np.zeros((B, k), np.float64); cnt = np.zeros(B, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(vector_index.lib.comet_hybrid_rrf_search(vector_index.h, text_index.h, p(Q), p(toks), p(offs), B, int(k), int(n_probes), int(ef_search), C.c_double(rrf_k),
                                                   p(ids), p(sc), p(cnt)))
    return ids, sc, cnt


# ---- segment layer (storage.go:489-626): one hybrid index per memtable / segment, results merged on the host -------------
def merge_results(results: list[HybridSearchResult]) -> list[HybridSearchResult] | None:
    """mergeResults storage_merge.go:13-46: deduplicate by document id, keep the HIGHEST score; nil for no input.
    (The reference's output order is Go map order — unspecified; this mirror and the device merge return ascending ids, which the
    stable descending sort below keeps among equal scores.)"""
    if not results:
        return None
    best: dict[int, float] = {}
    for r in results:
        if r.id not in best or r.score > best[r.id]:
            best[r.id] = r.score
    return [HybridSearchResult(i, best[i]) for i in sorted(best)]


def sort_results_by_score(results: list[HybridSearchResult]) -> None:
    """sortResultsByScore storage_merge.go:50-54: descending, in place."""
    results.sort(key=lambda r: -r.score)


class SegmentedHybridSearch:
    """persistentHybridSearch.Execute storage.go:489-626 without the storage engine: the same query runs against every
    segment's (vector index, text index) pair — newest first, as the reference walks memtables then segments — and the
    per-segment lists are merged (highest score per id), sorted descending and cut to k.

    Two forms. `with_query(fn)`: fn configures every per-segment HybridSearch (vector, text, fusion ...); every per-segment
    search is a GPU search through the C ABI and the merge is O(segments * k) on the host, where the reference has it.
    `with_vector(q)` (a vector-only query): ONE call into the library (`comet_segments_search`, index.SegmentSet) searches all
    the segments' vector indexes, resident in HBM, and merges on the device."""

    def __init__(self, segments):
        self.segments = list(segments)          # [(vector_index | None, text_index | None), ...], oldest first
        self.configure = None                   # applied to every per-segment HybridSearch (with_vector / with_text / ...)
        self.vector_query = None
        self.k = 10
        self.n_probes, self.ef_search, self.threshold, self.document_ids = 1, 0, 0.0, []   # hybrid defaults (hybrid_search_index.go:236)

    def with_k(self, k): self.k = int(k); return self
    def with_query(self, fn): self.configure = fn; return self
    def with_vector(self, q): self.vector_query = q; return self
    def with_n_probes(self, n): self.n_probes = int(n); return self
    def with_ef_search(self, ef): self.ef_search = int(ef); return self
    def with_threshold(self, t): self.threshold = float(t); return self
    def with_document_ids(self, *ids): self.document_ids = [int(i) for i in ids]; return self

    def _per_segment(self, h: HybridSearch) -> HybridSearch:
        if self.configure is not None:
            return self.configure(h)
        h = h.with_vector(self.vector_query).with_n_probes(self.n_probes).with_ef_search(self.ef_search).with_threshold(self.threshold)
        return h.with_document_ids(*self.document_ids) if self.document_ids else h

    def execute_on_host(self) -> list[HybridSearchResult]:
        if self.k < 0:
            raise ValueError("k must not be negative")      # merged[:k] panics in the reference
        allr: list[HybridSearchResult] = []
        for vec, txt in reversed(self.segments):
            allr.extend(self._per_segment(HybridSearch(vec, txt).with_k(self.k)).execute())
        merged = merge_results(allr) or []
        sort_results_by_score(merged)
        return merged[:self.k] if len(merged) > self.k else merged      # storage.go:621-623

    def execute(self) -> list[HybridSearchResult]:
        if self.configure is not None or self.vector_query is None:
            return self.execute_on_host()
        import numpy as np
        from .index import SegmentSet
        vecs = [v for v, _ in self.segments if v is not None]
        if len(vecs) != len(self.segments):
            raise ValueError("vector query specified but no vector index configured")
        ids, sc, cn = SegmentSet(vecs).search_batch(np.asarray(self.vector_query, dtype=np.float32)[None, :], self.k,
                                                    threshold=self.threshold if self.threshold > 0 else 0.0, nprobes=max(self.n_probes, 0),
                                                    ef_search=max(self.ef_search, 0), document_ids=self.document_ids)
        return [HybridSearchResult(int(i), float(s)) for i, s in zip(ids[0, :cn[0]], sc[0, :cn[0]])]
