"""Concurrent use of one index through the C ABI, after flat_index_search_test.go:392-465
(TestFlatIndexConcurrentSearch: 50 goroutines searching; TestFlatIndexConcurrentAddAndSearch: adds racing searches).
ctypes releases the GIL around every ABI call, so these threads really enter libcomet_hip.so concurrently."""
import threading

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, FlatIndex, IVFPQIndex, L2_SQUARED

pytestmark = pytest.mark.gpu


def run_threads(fns):
    errs = []

    def wrap(f):
        try:
            f()
        except BaseException as e:   # noqa: BLE001 - collected and re-raised in the main thread
            errs.append(e)
    th = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]


def test_reference_shape_50_concurrent_searches(ctx):
    g = FlatIndex(ctx, 3, EUCLIDEAN)
    o = orc.Flat(3, "l2")
    for i in range(100):
        v = np.array([i, 0, 0], np.float32)
        g.add(i + 1, v); o.add(i + 1, v)

    def one(i):
        q = np.array([i % 10, 0, 0], np.float32)
        res = g.new_search().with_query(q).with_k(10).execute()
        n, oi, os_ = o.search(q, 10)
        assert [r.id for r in res] == oi.tolist() and [r.score for r in res] == os_.tolist()
    run_threads([lambda i=i: one(i) for i in range(50)])


@pytest.mark.parametrize("metric", [L2_SQUARED, COSINE])
def test_searches_race_adds_removes_and_other_indexes(ctx, metric):
    n, d, B, K = 20_000, 64, 16, 10
    X = orc.synth(71, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = FlatIndex(ctx, d, metric); g.add_batch(ids, X)
    o = orc.Flat(d, metric); assert o.add_batch(ids, X) == 0
    # a second index kind on the same context, searched concurrently (shared stream / scratch arena)
    p = IVFPQIndex(ctx, d, L2_SQUARED, 8, 8, 4); p.train(X[:2000]); p.add_batch(ids[:5000], X[:5000])
    po = orc.IVFPQ(d, "l2_squared", 8, 8, 4); assert po.train(X[:2000]) == 0 and po.add_batch(ids[:5000], X[:5000]) == 0
    Q = orc.synth(72, 0, 8 * B * d).reshape(8, B, d)
    want = [[o.search(q, K) for q in Q[t]] for t in range(8)]
    pwant = [po.search(q, K, 4) for q in Q[0]]
    # mutator: adds vectors far outside the data (never in anybody's top-K) and soft-deletes some of them again
    noise = orc.synth(73, 0, 64 * d).reshape(64, d)
    # L2: coordinates near 50; cosine: the all-negative diagonal, whose distance to a random query is 1 +- 0.15 while the
    # best of 20k random rows sits near 0.55
    far = (50.0 + noise) if metric != COSINE else (-1.0 + noise * np.float32(1e-3))
    far = far.astype(np.float32)
    stop = threading.Event()

    def searcher(t):
        for rep in range(12):
            gi, gs, gc = g.search_batch(Q[t], K, mode=rep % 3)        # auto / strict / fast paths interleaved
            for b in range(B):
                cnt, oi, os_ = want[t][b]
                assert gc[b] == cnt and np.array_equal(gi[b, :cnt], oi) and np.array_equal(gs[b, :cnt].view(np.uint32), os_.view(np.uint32)), (t, rep, b)

    def pq_searcher():
        for rep in range(12):
            gi, gs, gc = p.search_batch(Q[0], K, nprobes=4)
            for b in range(B):
                cnt, oi, os_ = pwant[b]
                assert gc[b] == cnt and np.array_equal(gi[b, :cnt], oi) and np.array_equal(gs[b, :cnt].view(np.uint32), os_.view(np.uint32)), (rep, b)

    def mutator():
        for i in range(64):
            g.add(1_000_000 + i, far[i].copy())
            if i % 4 == 3:
                g.remove(1_000_000 + i - 1)
        stop.set()

    def fluent():
        while not stop.is_set():
            r = g.new_search().with_query(Q[1][0]).with_k(K).execute()
            assert [x.id for x in r] == want[1][0][1].tolist()
    run_threads([lambda t=t: searcher(t) for t in range(8)] + [pq_searcher, mutator, fluent])
    assert len(g) == n + 64
    g.flush()
    assert len(g) == n + 64 - 16
