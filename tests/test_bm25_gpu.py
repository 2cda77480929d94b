"""BM25 on the GPU vs the CPU oracle (bm25_index.go, bm25_index_search.go:278-397), from token ids.
Scores are compared as float64 bit patterns (the accumulation order is the reference's) and as float32."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import BM25SearchIndex

pytestmark = pytest.mark.gpu


def make_docs(n_docs, vocab, seed):
    """Zipf-like token ids, doc length 8..64, deterministic."""
    rng = np.random.default_rng(seed)
    docs = {}
    for d in range(1, n_docs + 1):
        ln = int(rng.integers(8, 65))
        toks = np.minimum((rng.pareto(1.1, ln) * 3).astype(np.int64), vocab - 1)
        docs[d * 3] = toks.astype(np.uint32)          # ids not contiguous
    return docs


def build(ctx, docs):
    g, o = BM25SearchIndex(ctx), orc.BM25()
    for i, t in docs.items():
        g.add(i, t); o.add(i, t)
    return g, o


def check(g, o, queries, k, **kw):
    ids, sc, sc64, cnt = g.search_batch(queries, k, document_ids=kw.get("filter_ids", ()))
    for b, q in enumerate(queries):
        n, oi, os32, os64 = o.search(q, k, filter_ids=kw.get("filter_ids", ()))
        assert cnt[b] == n, (b, cnt[b], n)
        m = min(n, ids.shape[1])
        assert np.array_equal(ids[b, :m], oi[:m]), (b, ids[b, :m], oi[:m])
        assert np.array_equal(sc64[b, :m].view(np.uint64), os64[:m].view(np.uint64)), (b, sc64[b, :m], os64[:m])
        assert np.array_equal(sc[b, :m].view(np.uint32), os32[:m].view(np.uint32))


def test_bm25_matches_oracle(ctx):
    docs = make_docs(3000, 500, 1)
    g, o = build(ctx, docs)
    assert g.num_docs() == o.num_docs() == 3000
    assert g.avg_doc_len() == o.avg_doc_len()
    rng = np.random.default_rng(2)
    queries = [list(rng.integers(0, 40, int(rng.integers(1, 6)))) for _ in range(24)]
    queries += [[0, 0, 1], [499], [100000], [3, 100000, 3], []]      # duplicate tokens count twice (:299); unknown tokens; empty
    check(g, o, queries, 10)
    check(g, o, queries, 1)
    check(g, o, queries[:6], 0)                                       # k <= 0 -> all hits
    check(g, o, queries, 25, filter_ids=[i * 3 for i in range(1, 3001, 2)] + [5])
    # soft delete: N still counts the document (bm25_index_search.go:288), results skip it
    for i in (3, 6, 9, 300, 9000):
        g.remove(i); o.remove(i)
    check(g, o, queries, 10)
    # re-adding an id replaces the document (bm25_index.go:172-174)
    g.add(30, [1, 1, 2, 7]); o.add(30, [1, 1, 2, 7])
    check(g, o, queries, 10)
    res = g.new_search().with_query([1, 2]).with_k(5).execute()
    n, oi, os32, _ = o.search([1, 2], 5)
    assert [r.id for r in res] == oi.tolist() and [r.score for r in res] == os32.tolist()


def test_bm25_empty_and_flush(ctx):
    g = BM25SearchIndex(ctx)
    ids, sc, sc64, cnt = g.search_batch([[1, 2]], 5)
    assert cnt[0] == 0
    g.add(1, [5, 6]); g.add(2, [6, 7, 7])
    g.remove(1); g.flush()
    assert g.num_docs() == 1
    ids, sc, sc64, cnt = g.search_batch([[6]], 5)
    assert cnt[0] == 1 and ids[0, 0] == 2


def test_bm25_dense_topk_form(ctx):
    """Queries that touch most of the collection and want few results go through the dense form of the top-K (two walks over the
    accumulator row, bm25_topk_kernel): frequent tokens, duplicated documents (equal scores: lowest document index first), soft
    deletes, a document filter, k from 1 to 64 and beyond (65: the list form again) — float64 bit patterns as always."""
    docs = make_docs(6000, 300, 7)
    keys = sorted(docs)
    for j in range(0, 400, 4):                       # copies of earlier documents: exact score ties
        docs[keys[2000 + j]] = docs[keys[j]].copy()
    g, o = build(ctx, docs)
    frequent = [[0], [0, 1], [1, 2, 0, 3], [0, 0, 1], [2, 7, 0]]            # token 0 is in nearly every document
    for k in (1, 10, 64, 65):
        check(g, o, frequent, k)
    for d in keys[:300:3]:
        g.remove(d); o.remove(d)
    check(g, o, frequent, 10)
    check(g, o, frequent, 10, filter_ids=keys[1::2])
    g.flush(); o.flush()
    check(g, o, frequent, 25)
