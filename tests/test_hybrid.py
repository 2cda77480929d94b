"""Host-side fusion (fusion.go) — CPU test against the reference's KAT — and the config-5 hybrid path on the GPU
(IVF + BM25 + RRF) against the oracle's restatement of the same pipeline."""
import json
from pathlib import Path

import numpy as np
import pytest

from comet_amd.hybrid import RECIPROCAL_RANK_FUSION, HybridSearch, reciprocal_rank_fusion, score_map_to_ranks

KATS = json.loads((Path(__file__).parent / "golden" / "reference_kats.json").read_text())


def test_rrf_reference_kat():
    b = KATS["rrf"]
    got = reciprocal_rank_fusion({int(k): v for k, v in b["vector"].items()}, {int(k): v for k, v in b["text"].items()}, b["K"])
    for k, e in b["expected"].items():
        assert abs(got[int(k)] - e) <= b["tolerance"]
    assert got[1] > max(got[2], got[3], got[4])
    assert score_map_to_ranks({1: 0.1, 2: 0.3, 3: 0.5}, True) == {1: 0, 2: 1, 3: 2}       # fusion_test.go:394-438
    assert score_map_to_ranks({1: 20.0, 2: 15.0, 4: 10.0}, False) == {1: 0, 2: 1, 4: 2}


def test_weighted_sum_max_min_fusion_reference_kats():
    """fusion_test.go:8-135 (weighted sum: equal / custom / zero weights :483-508, one side empty, both empty), :203-243 (max), :245-315 (min, no overlap) — the
    reference's maps and the values its tests compare with `!=`, on the host mirror's Combine restatements."""
    from comet_amd.hybrid import max_fusion, min_fusion, weighted_sum_fusion
    v, t = {1: 0.5, 2: 0.3, 3: 0.8}, {1: 10.0, 2: 20.0, 4: 15.0}
    assert weighted_sum_fusion(v, t, 1.0, 1.0) == {1: 10.5, 2: 20.3, 3: 0.8, 4: 15.0}
    assert weighted_sum_fusion({1: 0.5}, {1: 10.0}, 2.0, 0.5) == {1: 6.0}
    assert weighted_sum_fusion({1: 0.5}, {1: 100.0}, 1.0, 0.0) == {1: 0.5}
    assert weighted_sum_fusion({1: 0.5, 2: 0.3}, {}) == {1: 0.5, 2: 0.3}                   # DefaultFusion: weights 1 / 1 (fusion.go: DefaultFusionConfig)
    assert weighted_sum_fusion({}, {1: 10.0, 2: 20.0}) == {1: 10.0, 2: 20.0}
    assert weighted_sum_fusion({}, {}) == {}
    v, t = {1: 0.5, 2: 0.8, 3: 0.3}, {1: 10.0, 2: 5.0, 4: 15.0}
    assert max_fusion(v, t) == {1: 10.0, 2: 5.0, 3: 0.3, 4: 15.0}
    assert min_fusion(v, t) == {1: 0.5, 2: 0.8}
    assert min_fusion({1: 0.5, 2: 0.8}, {3: 10.0, 4: 15.0}) == {}


def test_min_max_fusion_follow_the_reference():
    """minFusion keeps only documents in BOTH result maps (fusion.go:291-306); maxFusion the union (fusion.go:252-271)."""
    from comet_amd.hybrid import MAX_FUSION, MIN_FUSION
    from comet_amd.index import TextResult, VectorResult

    class FakeSearch:
        def __init__(self, res): self.res = res
        def __getattr__(self, name): return lambda *a, **k: self
        def execute(self): return self.res

    class FakeIndex:
        def __init__(self, res): self.res = res
        def new_search(self): return FakeSearch(self.res)

    vec = FakeIndex([VectorResult(1, 0.5), VectorResult(2, 0.25), VectorResult(3, 2.0)])
    txt = FakeIndex([TextResult(2, 1.5), TextResult(3, 0.75), TextResult(4, 9.0)])
    run = lambda kind: {r.id: r.score for r in HybridSearch(vec, txt).with_vector([0.0]).with_text([1]).with_k(10).with_fusion_kind(kind).execute()}
    assert run(MIN_FUSION) == {2: 0.25, 3: 0.75}
    assert run(MAX_FUSION) == {1: 0.5, 2: 1.5, 3: 2.0, 4: 9.0}
    assert [r.id for r in HybridSearch(vec, txt).with_vector([0.0]).with_text([1]).with_k(10).with_fusion_kind(MAX_FUSION).execute()] == [4, 3, 2, 1]


@pytest.mark.gpu
def test_config5_hybrid_ivf_bm25_rrf(ctx):
    import ctypes as C
    import oracle_lib as orc
    from comet_amd import BM25SearchIndex, IVFIndex, L2_SQUARED
    n, d, nlist, k = 4000, 32, 16, 10
    X = orc.synth(3, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = IVFIndex(ctx, d, nlist, L2_SQUARED); o = orc.IVF(d, "l2_squared", nlist)
    g.train(X[:1000]); o.train(X[:1000]); g.add_batch(ids, X); o.add_batch(ids, X)
    rng = np.random.default_rng(4)
    tg, to = BM25SearchIndex(ctx), orc.BM25()
    for i in range(1, n + 1):
        toks = np.minimum((rng.pareto(1.1, int(rng.integers(8, 40))) * 3).astype(np.int64), 199).astype(np.uint32)
        tg.add(i, toks); to.add(i, toks)
    for qi in range(5):
        q = X[qi * 37] + np.float32(0.01)
        terms = [int(t) for t in rng.integers(0, 20, 3)]
        res = HybridSearch(g, tg).with_vector(q).with_text(terms).with_k(k).with_n_probes(4).with_fusion_kind(RECIPROCAL_RANK_FUSION).execute()
        # oracle pipeline: both sub-searches cut to k, then RRF (orc_rrf), sort desc, cut to k
        nv, vi, vs = o.search(q, k, 4)
        nt, ti, ts32, ts = to.search(terms, k)
        vsd, tsd = vs.astype(np.float64), ts32.astype(np.float64)       # float64(result.GetScore()) of float32 scores
        oi, os_ = np.zeros(2 * k, np.uint32), np.zeros(2 * k, np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        m = orc.lib().orc_rrf(C.c_double(60.0), p(vi), p(vsd), nv, p(ti), p(tsd), nt, p(oi), p(os_))
        want = sorted(zip(os_[:m].tolist(), oi[:m].tolist()), key=lambda t: -t[0])[:k]
        assert sorted(r.score for r in res) == sorted(s for s, _ in want)
        assert {r.id for r in res if r.score > want[-1][0]} == {i for s, i in want if s > want[-1][0]}


def test_merge_results_reference_kats():
    """storage_merge_test.go:8-101 (TestMergeResults, TestMergeResults_Nil) and :103-190 (TestSortResultsByScore)."""
    from comet_amd.hybrid import HybridSearchResult as R, merge_results, sort_results_by_score
    cases = [([(1, 0.5), (2, 0.8), (3, 0.3)], {1: 0.5, 2: 0.8, 3: 0.3}),
             ([(1, 0.5), (2, 0.8), (1, 0.9), (3, 0.3), (2, 0.6)], {1: 0.9, 2: 0.8, 3: 0.3}),
             ([(1, 0.1), (1, 0.5), (1, 0.9), (1, 0.3)], {1: 0.9})]
    for inp, want in cases:
        got = merge_results([R(i, s) for i, s in inp])
        assert {r.id: r.score for r in got} == want and len(got) == len(want)
    assert merge_results(None) is None and merge_results([]) is None
    rs = [R(1, 0.3), R(2, 0.9), R(3, 0.5), R(4, 0.7)]
    sort_results_by_score(rs)
    assert [r.id for r in rs] == [2, 4, 3, 1]
    e = []; sort_results_by_score(e); assert e == []
    one = [R(1, 0.5)]; sort_results_by_score(one); assert one[0].id == 1


@pytest.mark.gpu
def test_segmented_search_equals_one_index(ctx):
    """storage.go:489-626: the same vector query over three segment indexes, merged on the host, returns what one index
    holding all the rows returns (vector-only queries keep the distances as scores, so 'highest score' keeps the WORST copy of
    a duplicated id, as in the reference; ids here are unique across segments)."""
    import numpy as np
    import oracle_lib as orc
    from comet_amd import FlatIndex, L2_SQUARED
    from comet_amd.hybrid import HybridSearch, SegmentedHybridSearch
    n, d = 3000, 32
    X = orc.synth(91, 0, n * d).reshape(n, d); q = orc.synth(92, 0, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    whole = FlatIndex(ctx, d, L2_SQUARED); whole.add_batch(ids, X)
    segs = []
    for lo, hi in ((0, 1000), (1000, 1800), (1800, 3000)):
        f = FlatIndex(ctx, d, L2_SQUARED); f.add_batch(ids[lo:hi], X[lo:hi]); segs.append((f, None))
    k = 3000          # hybrid results sort by score DESCENDING (hybrid_search_index.go:603): compare the full lists
    one = HybridSearch(whole, None).with_vector(q).with_k(k).execute()
    many = SegmentedHybridSearch(segs).with_k(k).with_query(lambda s: s.with_vector(q)).execute()
    assert [(r.id, r.score) for r in one] == [(r.id, r.score) for r in many]


def _host_merge(per_segment, k):
    """storage.go:600-623 on plain (ids, scores) lists: highest score per id, descending (equal scores: ascending id), cut to k."""
    from comet_amd.hybrid import HybridSearchResult as R, merge_results, sort_results_by_score
    allr = [R(int(i), float(s)) for ids, sc in per_segment for i, s in zip(ids, sc)]
    merged = merge_results(allr) or []
    sort_results_by_score(merged)
    merged = merged[:k] if len(merged) > k else merged
    return [(r.id, np.float32(r.score)) for r in merged]


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2_squared", "cosine", "l2"])
def test_segments_fused_search_matches_oracle_segments(ctx, metric):
    """comet_segments_search: per-segment top-k + mergeResults + sortResultsByScore + cut on the device equals the ORACLE's
    per-segment Flat searches merged on the host — batch of queries, documents re-written in newer segments (the same id in several
    segments), duplicated vectors under different ids (equal scores), a segment shorter than k, an empty segment."""
    import oracle_lib as orc
    from comet_amd import FlatIndex
    from comet_amd.index import SegmentSet
    d, B = 48, 9
    X = orc.synth(301, 0, 2600 * d).reshape(2600, d); Q = orc.synth(302, 0, B * d).reshape(B, d)
    X[1500:1520] = X[100:120]                                   # the same vectors under other ids, in another segment: equal scores
    ids = np.arange(10, 2610, dtype=np.uint32)
    ids[1200:1260] = ids[40:100]                                # documents re-written in a newer segment (other vectors)
    bounds = [(0, 1000), (1000, 1000), (1000, 1007), (1007, 1800), (1800, 2600)]   # one empty, one of 7 rows
    gsegs, osegs = [], []
    for lo, hi in bounds:
        g = FlatIndex(ctx, d, metric); o = orc.Flat(d, metric)
        if hi > lo:
            g.add_batch(ids[lo:hi], X[lo:hi]); assert o.add_batch(ids[lo:hi], X[lo:hi]) == 0
        gsegs.append(g); osegs.append(o)
    ss = SegmentSet(gsegs)
    for k in (1, 10, 100, 1500, 3000):
        gi, gs, gc = ss.search_batch(Q, k)
        for qi in range(B):
            per = []
            for o in osegs:
                n, oi, osc = o.search(Q[qi], k)
                per.append((oi[:min(n, k)], osc[:min(n, k)]) if n > 0 else ([], []))
            want = _host_merge(per, k)
            got = list(zip(gi[qi, :gc[qi]].tolist(), gs[qi, :gc[qi]]))
            assert gc[qi] == len(want), (metric, k, qi)
            assert [g[0] for g in got] == [w[0] for w in want], (metric, k, qi)
            assert [g[1].tobytes() for g in got] == [w[1].tobytes() for w in want], (metric, k, qi)


@pytest.mark.gpu
def test_segments_fused_search_equals_host_mirror(ctx):
    """The device merge and the host mirror of storage.go:489-626 (per-segment GPU searches merged in Python) agree for mixed index kinds
    (Flat + IVF + HNSW segments), with a threshold, a document filter, soft deletes, and beyond the LDS merge (segments x k > 8192)."""
    import oracle_lib as orc
    from comet_amd import FlatIndex, HNSWIndex, IVFIndex, L2_SQUARED
    from comet_amd.hybrid import SegmentedHybridSearch
    d = 32
    X = orc.synth(311, 0, 9000 * d).reshape(9000, d)
    ids = np.arange(1, 9001, dtype=np.uint32)
    f1 = FlatIndex(ctx, d, L2_SQUARED); f1.add_batch(ids[:3000], X[:3000]); f1.remove(17); f1.remove(2999)
    iv = IVFIndex(ctx, d, 16, L2_SQUARED); iv.train(X[3000:6000]); iv.add_batch(ids[3000:6000], X[3000:6000])
    hn = HNSWIndex(ctx, d, L2_SQUARED, 8, 64, 64); hn.add_batch(ids[6000:6400], X[6000:6400])
    f2 = FlatIndex(ctx, d, L2_SQUARED); f2.add_batch(ids[6400:], X[6400:])
    segs = [(f1, None), (iv, None), (hn, None), (f2, None)]
    for seed, (k, thr, docs, npb) in enumerate([(10, 0.0, (), 1), (50, 0.0, (), 4), (2500, 0.0, (), 16), (40, -1.0, (), 2),
                                                 (30, 0.0, tuple(range(5, 9000, 7)), 3)]):
        q = orc.synth(320 + seed, 0, d)
        if thr < 0:     # a threshold that cuts the unthresholded result list in half
            sc = sorted(r.score for r in SegmentedHybridSearch(segs).with_k(k).with_vector(q).with_n_probes(npb).with_ef_search(64).execute_on_host())
            thr = float(np.float32(sc[len(sc) // 2]))
        def build():
            s = SegmentedHybridSearch(segs).with_k(k).with_vector(q).with_n_probes(npb).with_ef_search(64)
            if thr: s = s.with_threshold(thr)
            if docs: s = s.with_document_ids(*docs)
            return s
        fused = build().execute()
        host = build().execute_on_host()
        assert len(host) > 0
        assert [(r.id, np.float32(r.score).tobytes()) for r in fused] == [(r.id, np.float32(r.score).tobytes()) for r in host], (k, thr, npb)


@pytest.mark.gpu
def test_segments_fused_search_edges(ctx):
    """k == 0 is merged[:0] (storage.go:621-623), k < 0 a slice-bounds panic (an error here), a zero query under cosine is the
    memtable search's error, mismatched dimensions are refused."""
    import oracle_lib as orc
    from comet_amd import COSINE, CometError, FlatIndex, ZeroVectorError
    from comet_amd.index import SegmentSet
    d = 16
    X = orc.synth(331, 0, 300 * d).reshape(300, d)
    a = FlatIndex(ctx, d, COSINE); a.add_batch(np.arange(300, dtype=np.uint32), X)
    b = FlatIndex(ctx, d, COSINE); b.add_batch(np.arange(300, 600, dtype=np.uint32), X[::-1].copy())
    ss = SegmentSet([a, b])
    ids, sc, cn = ss.search_batch(X[:4], 0)
    assert cn.tolist() == [0, 0, 0, 0]
    with pytest.raises(CometError):
        ss.search_batch(X[:1], -1)
    with pytest.raises(ZeroVectorError):
        ss.search_batch(np.zeros((1, d), np.float32), 5)
    ids, sc, cn = ss.search_batch(X[:3], 5)
    assert cn.tolist() == [5, 5, 5] and np.all(np.diff(sc, axis=1) <= 0)         # descending
    c8 = FlatIndex(ctx, 8, COSINE)
    with pytest.raises(CometError):
        SegmentSet([a, c8]).search_batch(X[:1], 5)
    with pytest.raises(ValueError):
        SegmentSet([])


@pytest.mark.gpu
def test_segments_search_device_buffers(ctx):
    """comet_segments_search_dev (queries and results resident in HBM, what a batched host loop uses) returns what the host-buffer
    entry point returns, for IVFPQ segments too (their two-stage search runs inside the fused call)."""
    import oracle_lib as orc
    from comet_amd import FlatIndex, IVFPQIndex, L2_SQUARED
    from comet_amd.index import SegmentSet
    d, B, k = 32, 12, 7
    X = orc.synth(341, 0, 5000 * d).reshape(5000, d); Q = orc.synth(342, 0, B * d).reshape(B, d)
    ids = np.arange(1, 5001, dtype=np.uint32)
    a = FlatIndex(ctx, d, L2_SQUARED); a.add_batch(ids[:2000], X[:2000])
    b = IVFPQIndex(ctx, d, L2_SQUARED, 8, 8, 8); b.train(X[2000:4000]); b.add_batch(ids[2000:5000], X[2000:5000])
    ss = SegmentSet([a, b])
    hi, hs, hc = ss.search_batch(Q, k, nprobes=4)
    q_dev = ctx.alloc(Q.nbytes); ctx.upload(q_dev, Q)
    o_ids, o_sc, o_cn = ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    ss.search_batch_dev(q_dev, B, k, o_ids, o_sc, o_cn, k, nprobes=4)
    ctx.sync()
    di, ds, dc = ctx.download(o_ids, (B, k), np.uint32), ctx.download(o_sc, (B, k), np.float32), ctx.download(o_cn, (B,), np.int32)
    assert np.array_equal(dc, hc) and np.all(dc == k)
    assert np.array_equal(di, hi) and np.array_equal(ds.view(np.uint32), hs.view(np.uint32))
    for p in (q_dev, o_ids, o_sc, o_cn):
        ctx.free(p)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ivf", "flat_cos", "ivfpq"])
def test_hybrid_rrf_on_the_device_equals_the_per_query_form(ctx, kind):
    """comet_hybrid_rrf_search (vector leg on a second lane beside the text leg, rrf_fuse_kernel, one block back) against B calls of the per-query host mirror
    HybridSearch(...).with_fusion_kind(RECIPROCAL_RANK_FUSION).execute() — hybridSearch.Execute hybrid_search_index.go:477-615 — and, for the fused scores,
    against the oracle's RRF (fusion.go:174-243). Includes queries whose text matches nothing (the vector leg's own scores come back, sorted descending)."""
    import ctypes as C
    import oracle_lib as orc
    from comet_amd import COSINE, L2_SQUARED, BM25SearchIndex, FlatIndex, IVFIndex, IVFPQIndex
    from comet_amd.hybrid import hybrid_rrf_search_batch
    n, d, k, B = 6000, 32, 10, 48
    X = orc.synth(13, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    if kind == "ivf":
        g = IVFIndex(ctx, d, 16, L2_SQUARED); g.train(X[:1500]); npb = 4
    elif kind == "flat_cos":
        g = FlatIndex(ctx, d, COSINE); npb = 0
    else:
        g = IVFPQIndex(ctx, d, L2_SQUARED, 16, 8, 6); g.train(X[:1500]); npb = 4
    g.add_batch(ids, X)
    rng = np.random.default_rng(14)
    tg = BM25SearchIndex(ctx)
    for i in range(1, n + 1):
        tg.add(i, np.minimum((rng.pareto(1.1, int(rng.integers(8, 40))) * 3).astype(np.int64), 199).astype(np.uint32))
    Q = (X[(np.arange(B) * 53) % n] + orc.synth(15, 0, B * d).reshape(B, d) * np.float32(0.05)).astype(np.float32)
    toks = [[int(t) for t in rng.integers(0, 20, 3)] for _ in range(B)]
    toks[3] = [100000, 100001]                 # unknown terms: no text hits -> the vector hits with their own scores
    toks[7] = []                               # no text at all
    gi, gs, gc = hybrid_rrf_search_batch(g, tg, Q, toks, k=k, n_probes=npb)
    for b in range(B):
        s = HybridSearch(g, tg).with_vector(Q[b]).with_k(k).with_fusion_kind(RECIPROCAL_RANK_FUSION)
        if npb:
            s = s.with_n_probes(npb)
        if toks[b]:
            s = s.with_text(toks[b])
        want = s.execute()
        assert gc[b] == len(want), (b, gc[b], len(want))
        assert np.array_equal(np.sort(gs[b, :gc[b]])[::-1], gs[b, :gc[b]])                                  # descending
        assert sorted(r.score for r in want) == sorted(gs[b, :gc[b]].tolist()), (b, [r.score for r in want], gs[b, :gc[b]])
        cut = min(r.score for r in want) if want else 0.0
        assert {r.id for r in want if r.score > cut} == {int(i) for i, sc in zip(gi[b, :gc[b]], gs[b, :gc[b]]) if sc > cut}, b
    # same call with the context on ONE lane (both legs on lane 0)
    ctx.set_lanes(1)
    try:
        hi, hs, hc = hybrid_rrf_search_batch(g, tg, Q, toks, k=k, n_probes=npb)
    finally:
        ctx.set_lanes(4)
    assert np.array_equal(hi, gi) and np.array_equal(hs.view(np.uint64), gs.view(np.uint64)) and np.array_equal(hc, gc)
