"""The product's host-side Execute() shells (comet_amd.index.VectorSearch / TextSearch, comet_amd.hybrid.HybridSearch) run on the CPU over oracle-backed
indexes (tests/shell_adapters.py) through the reference's own tests of those shells: ivf_index_search_test.go:8-205 (a query and a node id in one Execute(),
several of each, with a threshold: results de-duplicated by node id), :208-295 (validation), hybrid_search_index_test.go:10-90 (vector only, text only),
:316-402 (weights), flat_index_search_test.go (k bounds through the shell). No GPU."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd.hybrid import WEIGHTED_SUM_FUSION, HybridSearch
from shell_adapters import OracleTextIndex, OracleVectorIndex


def ivf_index(rows):
    o = orc.IVF(3, "l2", 2)
    assert o.train(np.array([[0, 0, 0], [10, 10, 10], [1, 0, 0], [11, 10, 10]], np.float32)) == 0
    idx = OracleVectorIndex(o, 3, "ivf")
    return idx, [idx.add(v) for v in rows]


def unique_ids(res):
    ids = [r.id for r in res]
    assert len(set(ids)) == len(ids), ids
    return ids


def test_ivf_query_and_node_in_one_execute():
    idx, ids = ivf_index([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [10, 10, 10], [11, 10, 10]])
    res = idx.new_search().with_query([0, 1, 0]).with_node(ids[0]).with_k(2).with_n_probes(2).execute()
    assert 2 <= len(unique_ids(res)) <= 2                       # two queries x k 2, de-duplicated (sum aggregation), then LimitResults(k)
    # what the shell must have done: per-query top-2 = {[0,1,0] d 0, [1,0,0] d sqrt 2 (first of the ties in scan order)} and {[1,0,0] d 0, [2,0,0] d 1};
    # summed per id -> id0: sqrt2 + 0, id1: 0, id3: 1 -> ascending: id1 (0), id3 (1); cut to k = 2
    assert [r.id for r in res] == [ids[1], ids[3]] and [float(r.score) for r in res] == [0.0, 1.0]


def test_ivf_several_queries_and_nodes():
    idx, ids = ivf_index([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [10, 10, 10]])
    res = idx.new_search().with_query([1.1, 0, 0], [0, 1.1, 0]).with_node(ids[2], ids[3]).with_k(2).with_n_probes(2).execute()
    assert len(unique_ids(res)) == 2
    with pytest.raises(KeyError):
        idx.new_search().with_node(9999).with_k(2).execute()    # lookupNodeVectors: unknown node (ivf_index_search.go: "node %d not found")


def test_ivf_query_and_node_with_threshold():
    idx, ids = ivf_index([[1, 0, 0], [0, 1, 0], [0, 0, 1], [5, 0, 0], [0, 5, 0]])
    res = idx.new_search().with_query([1, 0, 0]).with_node(ids[1]).with_k(10).with_n_probes(2).with_threshold(2.0).execute()
    got = unique_ids(res)
    assert got and set(got) == {ids[0], ids[1], ids[2]}         # the two rows at distance 4+ from both queries never appear
    # sums over the two queries: [1,0,0]: 0 + sqrt2, [0,1,0]: sqrt2 + 0, [0,0,1]: sqrt2 + sqrt2
    r2 = np.float32(np.sqrt(np.float64(2.0)))
    assert [float(r.score) for r in res] == [float(r2), float(r2), float(np.float32(r2 + r2))]


def test_search_validation_through_the_shell():
    idx, ids = ivf_index([[1, 0, 0], [0, 1, 0]])
    with pytest.raises(ValueError, match="must specify either queries or node IDs"):
        idx.new_search().with_k(1).execute()
    with pytest.raises(ValueError, match="query dimension mismatch: expected 3, got 2"):
        idx.new_search().with_query([1, 0]).with_k(1).execute()
    with pytest.raises(ValueError, match="unknown aggregation kind"):
        idx.new_search().with_query([1, 0, 0]).with_score_aggregation("median").execute()
    # k bounds through LimitResults (flat_index_search_test.go:348-389: 0 / -1 / 100 -> all, 1 -> 1)
    for k, want in ((0, 2), (-1, 2), (100, 2), (1, 1)):
        assert len(idx.new_search().with_query([1, 0, 0]).with_k(k).with_n_probes(2).execute()) == want


def test_hybrid_vector_only_and_text_only():
    v = OracleVectorIndex(orc.Flat(3, "cosine"), 3, "flat")
    for row in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1.0, 0.1, 0.0]):
        v.add(row)
    res = HybridSearch(v, None).with_vector([1.0, 0.0, 0.0]).with_k(2).execute()
    assert len(res) == 2 and res[0].score > res[1].score        # the test's own check: the hybrid shell sorts DESCENDING, distances included
    assert [r.id for r in res] == [3, 1] and res[1].score == 0.0
    t = OracleTextIndex()
    for i, text in enumerate(("the quick brown fox jumps over the lazy dog", "the quick brown cat climbs a tree", "a lazy dog sleeps all day"), 1):
        t.add(i, text)
    res = HybridSearch(None, t).with_text(t.tok("quick brown")).with_k(2).execute()
    assert sorted(r.id for r in res) == [1, 2] and res[0].score >= res[1].score
    with pytest.raises(ValueError, match="no vector index configured"):
        HybridSearch(None, t).with_vector([1, 0, 0]).execute()


def test_hybrid_weights():
    v = OracleVectorIndex(orc.Flat(3, "cosine"), 3, "flat")
    t = OracleTextIndex()
    for i, (row, text) in enumerate((([1.0, 0.0, 0.0], "machine learning algorithms"), ([0.0, 1.0, 0.0], "machine learning basics")), 1):
        v.add(row, i); t.add(i, text)
    q = t.tok("machine learning")
    r1 = HybridSearch(v, t).with_vector([1.0, 0.0, 0.0]).with_text(q).with_k(10).execute()
    r2 = HybridSearch(v, t).with_vector([1.0, 0.0, 0.0]).with_text(q).with_fusion(WEIGHTED_SUM_FUSION, 10.0, 0.1).with_k(10).execute()
    assert len(r1) == 2 and len(r2) == 2
    # the two documents have the same length and both hold both query terms: equal BM25 scores s; cosine distances 0 and 1 -> combined s and 1 + s
    # (10 * 0 + 0.1 s and 10 + 0.1 s with the vector-heavy weights): the shell ranks the document at distance 1 FIRST (descending), as the reference does
    n, ti, ts32, _ = t.o.search(q, 10)
    s = float(np.float32(ts32[0])); assert n == 2 and ts32[0] == ts32[1]
    assert [r.id for r in r1] == [2, 1] and [r.score for r in r1] == [1.0 + s, 0.0 + s]
    assert [r.id for r in r2] == [2, 1] and [r.score for r in r2] == [10.0 * 1.0 + 0.1 * s, 10.0 * 0.0 + 0.1 * s]


def text_index(docs):
    t = OracleTextIndex()
    for i, text in docs:
        t.add(i, text)
    return t


def descending(res):
    return all(res[i].score <= res[i - 1].score for i in range(1, len(res)))


def test_text_search_shell_tables():
    """bm25_index_search_test.go:55-93 (WithK), :95-139 (aggregation kinds over two queries), :141-182 (cutoff), :184-271 (Execute table: empty and unmatched
    queries give no results and no error), :273-304 (several queries), :378-415 (ordering) — TextSearch.execute over the oracle's BM25"""
    from comet_amd.index import MAX_AGGREGATION, MEAN_AGGREGATION, SUM_AGGREGATION
    t = text_index([(i, "the quick brown fox jumps") for i in range(1, 11)])
    for k, want in ((3, 3), (5, 5), (10, 10), (0, 10), (-1, 10), (100, 10)):
        assert len(t.new_search().with_query(t.tok("quick")).with_k(k).execute()) == want, k
    t = text_index([(1, "fox dog cat"), (2, "fox dog"), (3, "cat mouse"), (4, "dog")])
    for kind in (SUM_AGGREGATION, MAX_AGGREGATION, MEAN_AGGREGATION):
        res = t.new_search().with_query(t.tok("fox"), t.tok("dog")).with_score_aggregation(kind).with_k(5).execute()
        assert res and descending(res) and {r.id for r in res} == {1, 2, 4}, kind
    t = text_index([(1, "fox fox fox fox"), (2, "fox fox"), (3, "the lazy dog sleeps"), (4, "cat and mouse"), (5, "quick brown fox jumps")])
    full = t.new_search().with_query(t.tok("fox")).with_k(10).with_cutoff(-1).execute()
    assert {r.id for r in full} == {1, 2, 5}
    for cutoff in (1, 2):
        cut = t.new_search().with_query(t.tok("fox")).with_k(10).with_cutoff(cutoff).execute()
        assert 1 <= len(cut) <= len(full) and [r.id for r in cut] == [r.id for r in full[:len(cut)]]
    t = text_index([(1, "the quick brown fox jumps over the lazy dog"), (2, "the lazy cat sleeps under the warm sun"), (3, "quick brown rabbits run through the forest"),
                    (4, "the forest is dark and mysterious"), (5, "dogs and cats are popular pets")])
    for query, least in (("fox", 1), ("quick brown", 2), ("elephant", 0), ("", 0)):
        res = t.new_search().with_query(t.tok(query)).with_k(5).execute()
        assert len(res) >= least and all(r.id != 0 and r.score >= 0 for r in res) and descending(res), query
        if query in ("elephant", ""):
            assert res == []
    t = text_index([(1, "fox and dog"), (2, "fox and cat"), (3, "dog and cat"), (4, "rabbit and mouse")])
    res = t.new_search().with_query(t.tok("fox"), t.tok("dog")).with_k(5).execute()
    assert res[0].id == 1 and {r.id for r in res} == {1, 2, 3}           # document 1 holds both query terms: its summed score leads
    t = text_index([(1, "fox fox fox fox fox"), (2, "fox fox fox"), (3, "fox"), (4, "the quick brown fox jumps"), (5, "cat and dog")])
    res = t.new_search().with_query(t.tok("fox")).with_k(10).execute()
    assert len(res) == 4 and res[0].id == 1 and descending(res)
    with pytest.raises(ValueError, match="must specify either queries or node IDs"):
        t.new_search().with_k(3).execute()


def pq_fixture():
    """createTrainedPQIndex(8, 4, 6, Euclidean), pq_index_search_test.go:9-53"""
    o = orc.PQ(8, "l2", 4, 6)
    assert o.train(np.array([[(i * 8 + j) % 10 for j in range(8)] for i in range(100)], np.float32)) == 0
    idx = OracleVectorIndex(o, 8, "pq")
    rows = [[1, 0, 0, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0, 0], [2, 0, 0, 0, 0, 0, 0, 0], [3, 0, 0, 0, 0, 0, 0, 0]]
    return idx, [idx.add(v) for v in rows]


def test_pq_search_shell_tables():
    """pq_index_search_test.go:56-276 (simple, threshold, by node, several nodes, unknown node, query + node, batch queries), :356-390 (k bounds 1/3/5/6/100 -> 1/3/5/6/6)"""
    idx, ids = pq_fixture()
    e0 = [1, 0, 0, 0, 0, 0, 0, 0]; e1 = [0, 1, 0, 0, 0, 0, 0, 0]; e2 = [0, 0, 1, 0, 0, 0, 0, 0]
    assert len(idx.new_search().with_query(e0).with_k(2).execute()) == 2
    res = idx.new_search().with_query(e0).with_k(10).with_threshold(1.5).execute()
    assert all(float(r.score) <= 1.5 for r in res)
    assert len(idx.new_search().with_node(ids[0]).with_k(3).execute()) == 3
    for s in (idx.new_search().with_node(ids[0], ids[1]).with_k(2), idx.new_search().with_query(e1).with_node(ids[0]).with_k(2),
              idx.new_search().with_query(e0, e1).with_node(ids[0], ids[1]).with_k(2), idx.new_search().with_query(e0, e1, e2).with_k(2)):
        assert len(unique_ids(s.execute())) == 2
    with pytest.raises(KeyError):
        idx.new_search().with_node(99999).with_k(3).execute()
    for k, want in ((1, 1), (3, 3), (5, 5), (6, 6), (100, 6)):
        assert len(idx.new_search().with_query(e0).with_k(k).execute()) == want
    # an index that was trained and holds nothing (:392-423)
    o = orc.PQ(8, "l2", 4, 6); assert o.train(np.array([[i] * 8 for i in range(100)], np.float32)) == 0
    assert OracleVectorIndex(o, 8, "pq").new_search().with_query(e0).with_k(5).execute() == []


def test_ivfpq_search_shell_tables():
    """ivfpq_index_search_test.go:9-72 (dim 8, nlist 2, M 4, nbits 4, 100 ramp training rows: 1..3 results for k 3), :596-661 (k bounds 1/3/5/10/100 -> 1/3/5/5/5)"""
    ramp = np.array([[i * 8 + j for j in range(8)] for i in range(100)], np.float32)
    o = orc.IVFPQ(8, "l2", 2, 4, 4); assert o.train(ramp) == 0
    idx = OracleVectorIndex(o, 8, "ivfpq")
    for v in ([1, 2, 3, 4, 5, 6, 7, 8], [2, 3, 4, 5, 6, 7, 8, 9], [10, 11, 12, 13, 14, 15, 16, 17], [0] * 8, [1] * 8):
        idx.add(v)
    res = idx.new_search().with_query([1, 2, 3, 4, 5, 6, 7, 8]).with_k(3).with_n_probes(2).execute()
    assert 1 <= len(res) <= 3 and len(res) == 3                  # both lists probed: all five rows are candidates
    o = orc.IVFPQ(8, "l2", 2, 4, 4); assert o.train(ramp) == 0
    idx = OracleVectorIndex(o, 8, "ivfpq")
    for i in range(5):
        idx.add([i * 10 + j for j in range(8)])
    for k, want in ((1, 1), (3, 3), (5, 5), (10, 5), (100, 5)):
        assert len(idx.new_search().with_query([0, 1, 2, 3, 4, 5, 6, 7]).with_k(k).with_n_probes(2).execute()) == want


def hnsw_index(rows, efs=200):
    o = orc.HNSW(3, "l2", 16, 200, efs, seed=7)
    idx = OracleVectorIndex(o, 3, "hnsw")
    return idx, [idx.add(v) for v in rows]


def test_hnsw_search_shell_tables():
    """hnsw_index_search_test.go:333-431 (by node: the query node leads; two nodes far apart -> two de-duplicated results; unknown node), :1047-1290 (efSearch default,
    override, 0 and -1 fall back to the index's efSearch) — VectorSearch.execute over the oracle's HNSW"""
    idx, ids = hnsw_index([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0]])
    res = idx.new_search().with_node(ids[0]).with_k(2).execute()
    assert [r.id for r in res] == [ids[0], ids[3]] and [float(r.score) for r in res] == [0.0, 1.0]
    idx, ids = hnsw_index([[i, 0, 0] for i in range(5)])
    res = idx.new_search().with_node(ids[0], ids[4]).with_k(2).execute()
    assert [r.id for r in res] == [ids[0], ids[4]] and [float(r.score) for r in res] == [0.0, 0.0]      # per query {0, 1} and {4, 3}: four ids, the two zero sums lead
    with pytest.raises(KeyError):
        hnsw_index([[1, 0, 0]])[0].new_search().with_node(9999).with_k(1).execute()
    rows5 = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [2, 0, 0]]
    idx, ids = hnsw_index(rows5)
    res = idx.new_search().with_query([1, 0, 0]).with_k(2).execute()
    assert len(res) == 2 and res[0].id == ids[0]
    rows10 = rows5 + [[0, 2, 0], [0, 0, 2], [1, 1, 1], [2, 2, 0], [0, 2, 2]]
    idx, ids = hnsw_index(rows10, efs=50)
    r1 = idx.new_search().with_query([1, 0, 0]).with_k(5).with_ef_search(200).execute()
    r2 = idx.new_search().with_query([1, 0, 0]).with_k(5).execute()
    assert len(r1) == 5 and [(r.id, float(r.score)) for r in r1] == [(r.id, float(r.score)) for r in r2]   # ten fully linked nodes: any ef >= 10 sees them all
    for ef in (0, -1):
        idx, ids = hnsw_index([[1, 0, 0], [0, 1, 0], [0, 0, 1]])
        res = idx.new_search().with_query([1, 0, 0]).with_k(2).with_ef_search(ef).execute()
        assert len(res) == 2 and res[0].id == ids[0], ef


def test_constructor_validation_tables():
    """TestNewFlatIndex / TestNewIVFIndex / TestNewPQIndex / TestNewIVFPQIndex / TestNewHNSWIndex (flat_index_test.go:11, ivf_index_test.go:11-40, pq_index_test.go:46-70,
    ivfpq_index_test.go:21-50, hnsw_index_test.go:15): the error rows whose checks come before any device work, with the reference's messages (flat_index.go:127,
    ivf_index.go:147-160, pq_index.go:135-155, ivfpq_index.go:113-140). No context is needed to fail."""
    from comet_amd import FlatIndex, HNSWIndex, IVFIndex, IVFPQIndex, PQIndex
    from comet_amd.index import _metric_code
    E, dim_msg = "l2", "dimension must be positive"
    for d in (0, -1):
        for make in (lambda: FlatIndex(None, d, E), lambda: IVFIndex(None, d, 10, E), lambda: PQIndex(None, d, E, 8, 8), lambda: IVFPQIndex(None, d, E, 10, 8, 8),
                     lambda: HNSWIndex(None, d, E)):
            with pytest.raises(ValueError, match=dim_msg):
                make()
    for nl in (0, -1):
        with pytest.raises(ValueError, match="nlist must be positive"):
            IVFIndex(None, 128, nl, E)
        with pytest.raises(ValueError, match="nlist must be positive"):
            IVFPQIndex(None, 128, E, nl, 8, 8)
    for M in (0, -1):
        with pytest.raises(ValueError, match="parameter M must be positive"):
            PQIndex(None, 128, E, M, 8)
        with pytest.raises(ValueError, match="parameter M must be positive"):
            IVFPQIndex(None, 128, E, 10, M, 8)
    with pytest.raises(ValueError, match="dimension 100 must be divisible by M 8"):
        PQIndex(None, 100, E, 8, 8)
    with pytest.raises(ValueError, match="dimension 100 must be divisible by M 8"):
        IVFPQIndex(None, 100, E, 10, 8, 8)
    for nb in (0, -1, 17):
        with pytest.raises(ValueError, match=r"parameter Nbits must be in \[1,16\]"):
            PQIndex(None, 128, E, 8, nb)
        with pytest.raises(ValueError, match=r"parameter Nbits must be in \[1,16\]"):
            IVFPQIndex(None, 128, E, 10, 8, nb)
    with pytest.raises(Exception, match="unknown distance kind"):           # NewDistance: distance.go:9 — the rows "invalid distance kind"
        _metric_code("invalid")
    assert [_metric_code(k) for k in ("l2", "l2_squared", "cosine")] == [0, 1, 2]


def test_hybrid_hands_aggregation_and_cutoff_to_both_sub_searches():
    """hybridSearch.Execute passes WithScoreAggregation / WithCutoff to the vector and the text search (hybrid_search_index.go:508-513,545-550; defaults Sum / -1, :230-239):
    a cutoff of 1 cuts each leg at its first score jump before the fusion; an unknown aggregation kind fails in the sub-search"""
    from comet_amd.index import MAX_AGGREGATION
    v = OracleVectorIndex(orc.Flat(3, "l2"), 3, "flat")
    for i, row in enumerate(([1, 0, 0], [1.01, 0, 0], [9, 0, 0], [9.5, 0, 0]), 1):
        v.add(row, i)
    t = text_index([(1, "fox fox fox"), (2, "fox"), (3, "dog"), (4, "dog cat")])
    q = t.tok("fox")
    full = HybridSearch(v, t).with_vector([1, 0, 0]).with_text(q).with_k(4).execute()
    assert {r.id for r in full} == {1, 2, 3, 4}
    cut = HybridSearch(v, t).with_vector([1, 0, 0]).with_text(q).with_k(4).with_cutoff(1).with_score_aggregation(MAX_AGGREGATION).execute()
    vec_cut = [r.id for r in v.new_search().with_query([1, 0, 0]).with_k(4).with_cutoff(1).execute()]
    txt_cut = [r.id for r in t.new_search().with_query(q).with_k(4).with_cutoff(1).execute()]
    assert 1 <= len(vec_cut) < 4 and 1 <= len(txt_cut) <= 2
    assert {r.id for r in cut} == set(vec_cut) | set(txt_cut) and len(cut) < len(full)
    with pytest.raises(ValueError, match="unknown aggregation kind"):
        HybridSearch(v, None).with_vector([1, 0, 0]).with_score_aggregation("median").execute()


def test_text_search_with_node():
    """bm25_index_search_test.go:32-53 (WithNode), :306-329 (a node and a direct query together), :345-358 (unknown node) + lookupNodeTexts' messages (bm25_index_search.go:233-260):
    a document's own tokens are searched for as a query; soft-deleted and unknown documents fail by name"""
    t = text_index([(1, "quick brown fox"), (2, "lazy brown dog"), (3, "quick rabbit"), (4, "slow turtle")])
    res = t.new_search().with_node(1).with_k(5).execute()
    assert res and res[0].id == 1 and {r.id for r in res} == {1, 2, 3}                    # "quick brown fox" finds itself first, then the documents sharing a word
    both = t.new_search().with_node(1).with_query(t.tok("lazy dog")).with_k(5).execute()
    assert {r.id for r in both} == {1, 2, 3} and descending(both)
    by_hand = t.new_search().with_query(t.tok("lazy dog"), t.tok("quick brown fox")).with_k(5).execute()       # direct queries first, node queries behind them: the same sums
    assert [(r.id, float(r.score)) for r in both] == [(r.id, float(r.score)) for r in by_hand]
    with pytest.raises(KeyError, match="node ID 99 not found in index"):
        t.new_search().with_node(99).with_k(5).execute()
    t.remove(3)
    with pytest.raises(KeyError, match=r"node ID 3 not found in index \(deleted\)"):
        t.new_search().with_node(3).execute()
    assert {r.id for r in t.new_search().with_node(1).with_k(5).execute()} == {1, 2}     # the soft-deleted document is no hit either
    t.flush()
    with pytest.raises(KeyError, match="node ID 3 not found in index'$"):                 # (str(KeyError) quotes its message)
        t.new_search().with_node(3).execute()


def test_bm25_search_index_keeps_the_token_store_in_step_with_the_library():
    """BM25SearchIndex.add / remove / flush update the host-side docTokens / deletedDocs behind the library calls (a stub library stands in for libcomet_hip.so here:
    every entry point returns COMET_OK) — what TextSearch.with_node reads"""
    from comet_amd.index import BM25SearchIndex

    class StubLib:
        def __getattr__(self, name):
            return lambda *a: 0

    class StubCtx:
        lib, h = StubLib(), None

    ix = BM25SearchIndex(StubCtx())
    ix.add(7, np.array([3, 1, 2], np.uint32)); ix.add(8, [5, 5])
    assert ix._lookup_node_tokens([8, 7]) == [[5, 5], [3, 1, 2]]
    ix.remove(7)
    with pytest.raises(KeyError, match=r"node ID 7 not found in index \(deleted\)"):
        ix._lookup_node_tokens([7])
    ix.flush()
    with pytest.raises(KeyError, match="node ID 7 not found in index'$"):
        ix._lookup_node_tokens([7])
    ix.add(7, [9]); assert ix._lookup_node_tokens([7]) == [[9]]                           # re-added after the flush
    ix.h = None                                                                          # (nothing to destroy)
