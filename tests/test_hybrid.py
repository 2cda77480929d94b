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
