"""Sizes beyond the on-chip fast paths: the reference has no limit on k, efSearch or the BM25 result count, so neither does
the backend — selections of more than 4096 results sort in HBM, HNSW heaps larger than the LDS ones live in HBM, BM25 keeps
any number of hits. Everything is compared with the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, L2_SQUARED, BM25SearchIndex, FlatIndex, HNSWIndex
from comet_amd._lib import check as rc_check

pytestmark = pytest.mark.gpu


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [L2_SQUARED, COSINE])
def test_flat_k_beyond_the_lds_sort(ctx, metric):
    n, d = 12000, 16
    X = synth(1, n, d)
    X[7000:7400] = X[100:500]                                  # 400 exact duplicates: score ties resolved by insertion order
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = FlatIndex(ctx, d, metric); g.add_batch(ids, X)
    o = orc.Flat(d, metric); o.add_batch(ids, X)
    Q = np.vstack([synth(2, 2, d), X[100:101]])
    for k in (6000, 0, 11999):
        gi, gs, gc = g.search_batch(Q, k)
        for b in range(len(Q)):
            cnt, oi, os_ = o.search(Q[b], k)
            assert gc[b] == cnt == (n if k == 0 else k)
            assert np.array_equal(gi[b, :cnt], oi) and np.array_equal(bits(gs[b, :cnt]), bits(os_))


def test_merge_beyond_the_lds_sort(ctx):
    n, d, B, k, R = 18000, 8, 2, 5000, 3
    X = synth(3, n, d); Q = synth(4, B, d)
    X[n // 3 + 3] = X[5]; X[n - 1] = X[5]
    ids = np.arange(1, n + 1, dtype=np.uint32)
    full = orc.Flat(d, "l2_squared"); full.add_batch(ids, X)
    all_ids = np.zeros((R, B, k), np.uint32); all_sc = np.zeros((R, B, k), np.float32); all_cn = np.zeros((R, B), np.int32)
    for r in range(R):
        lo, hi = n * r // R, n * (r + 1) // R
        sh = FlatIndex(ctx, d, L2_SQUARED); sh.add_batch(ids[lo:hi], X[lo:hi])
        all_ids[r], all_sc[r], all_cn[r] = sh.search_batch(Q, k)
    bufs = [ctx.alloc(a.nbytes) for a in (all_ids, all_sc, all_cn)]
    for p, a in zip(bufs, (all_ids, all_sc, all_cn)):
        ctx.upload(p, a)
    o_ids, o_sc, o_cn = ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    rc_check(ctx.lib.comet_merge_topk_dev(ctx.h, C.c_void_p(bufs[0]), C.c_void_p(bufs[1]), C.c_void_p(bufs[2]), R, B, k, k,
                                          C.c_void_p(o_ids), C.c_void_p(o_sc), C.c_void_p(o_cn)))
    ctx.sync()
    mi, ms, mc = ctx.download(o_ids, (B, k), np.uint32), ctx.download(o_sc, (B, k), np.float32), ctx.download(o_cn, (B,), np.int32)
    for b in range(B):
        cnt, oi, os_ = full.search(Q[b], k)
        assert mc[b] == cnt == k and np.array_equal(mi[b], oi) and np.array_equal(bits(ms[b]), bits(os_))
    for p in bufs + [o_ids, o_sc, o_cn]:
        ctx.free(p)


def test_hnsw_ef_beyond_the_lds_heaps(ctx):
    n, d, m = 6000, 24, 8
    X = synth(5, n, d)
    o = orc.HNSW(d, "l2_squared", m, 60, 40, seed=3)
    assert o.add_batch(np.arange(1, n + 1), X) == 0
    ids, levels, vecs, eoff, edges = o.export()
    g = HNSWIndex(ctx, d, L2_SQUARED, m, 60, 40)
    g.load_graph(ids, levels, vecs, eoff, edges, o.entry(), o.max_level())
    Q = synth(6, 5, d)
    for k, ef in ((10, 3000), (2500, 3000), (0, 5000), (10, 100000)):      # ef > n: every reachable node is a result
        gi, gs, gc = g.search_batch(Q, k, ef_search=ef)
        for b in range(len(Q)):
            cnt, oi, os_ = o.search(Q[b], k, ef)
            assert gc[b] == cnt, (k, ef, gc[b], cnt)
            assert np.array_equal(gi[b, :cnt], oi[:cnt]) and np.array_equal(bits(gs[b, :cnt]), bits(os_[:cnt]))
    assert g.stat("hnsw_distance_evals") > 0


def test_bm25_more_than_2048_hits(ctx):
    rng = np.random.default_rng(8)
    g, o = BM25SearchIndex(ctx), orc.BM25()
    for i in range(1, 6001):
        t = rng.integers(0, 30, int(rng.integers(4, 20))).astype(np.uint32)
        g.add(i, t); o.add(i, t)
    queries = [[0, 1, 2], [3], [5, 5, 29], [100]]
    for k in (0, 3000, 5999, 10):
        ids, sc, sc64, cnt = g.search_batch(queries, k)
        for b, q in enumerate(queries):
            n, oi, os32, os64 = o.search(q, k)
            assert cnt[b] == n and ids.shape[1] >= n
            assert np.array_equal(ids[b, :n], oi[:n])
            assert np.array_equal(sc64[b, :n].view(np.uint64), os64[:n].view(np.uint64))
    # the accumulator rows are clean again: a second, different batch gives the oracle's answer too
    ids, sc, sc64, cnt = g.search_batch([[7], [8, 9]], 5)
    for b, q in enumerate([[7], [8, 9]]):
        n, oi, os32, os64 = o.search(q, 5)
        assert cnt[b] == n and np.array_equal(ids[b, :n], oi[:n]) and np.array_equal(sc64[b, :n].view(np.uint64), os64[:n].view(np.uint64))
