"""The MFMA fast path of the Flat scan (fp16 shadow GEMM proposes candidates, exact kernels decide) must
return results bit-identical to the strict exact-arithmetic path and to the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, CometError, FlatIndex

pytestmark = pytest.mark.gpu
METRICS = [COSINE, L2_SQUARED, EUCLIDEAN]


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    (i1, s1, c1), (i2, s2, c2) = a, b
    assert np.array_equal(c1, c2)
    for q in range(len(c1)):
        n = c1[q]
        assert np.array_equal(i1[q, :n], i2[q, :n]), (q, i1[q, :n], i2[q, :n])
        assert np.array_equal(bits(s1[q, :n]), bits(s2[q, :n])), q


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,d,B,k", [(20000, 128, 256, 100), (150000, 64, 256, 10), (9000, 768, 70, 1), (33333, 96, 300, 37), (70001, 40, 33, 5)])
def test_fast_equals_strict(ctx, metric, n, d, B, k):
    X = synth(0xC0FFEE + n, n, d)
    Q = synth(0xBEEF + n, B, d)
    g = FlatIndex(ctx, d, metric)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    strict = g.search_batch(Q, k, mode=1)
    assert g.stat("fast_queries") == 0 and g.stat("strict_queries") == B
    fast = g.search_batch(Q, k, mode=2)
    assert g.stat("fast_queries") + g.stat("fast_overflows") == B
    if n >= 256 * 32 * k:                              # enough 256-row tiles per requested result: (almost) no overflow
        assert g.stat("fast_queries") >= B * 0.9
    same(fast, strict)
    # a few queries against the CPU oracle too
    o = orc.Flat(d, metric); o.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    for q in (0, B // 2, B - 1):
        cnt, oi, os_ = o.search(Q[q], k)
        assert fast[2][q] == cnt and np.array_equal(fast[0][q, :cnt], oi) and np.array_equal(bits(fast[1][q, :cnt]), bits(os_))


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("d,B", [(64, 64), (128, 200)])     # the narrow tile; the wide tile with 64-row key units (small index: kernels_fast.hip flat_fast_unit_rows)
def test_fast_with_filter_delete_threshold(ctx, metric, d, B):
    n, k = 12000, 20
    X = synth(5, n, d); Q = synth(6, B, d)
    g = FlatIndex(ctx, d, metric)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    for i in range(1, 3000, 7):
        g.remove(i)
    flt = list(range(2, n, 2))
    ref = g.search_batch(Q, k, mode=1)
    thr = float(np.sort(ref[1][:, k // 2])[B // 2])
    for kw in (dict(), dict(document_ids=flt), dict(threshold=thr), dict(document_ids=flt, threshold=thr)):
        same(g.search_batch(Q, k, mode=2, **kw), g.search_batch(Q, k, mode=1, **kw))
    g.flush()
    same(g.search_batch(Q, k, mode=2), g.search_batch(Q, k, mode=1))
    # n not a multiple of the 256-row tile, tiny k, k larger than two keys per tile can supply
    same(g.search_batch(Q, 200, mode=2), g.search_batch(Q, 200, mode=1))


def test_fast_adversarial_ties_and_clusters(ctx):
    """Clustered / duplicated data inserted consecutively puts many winners in one 256-row tile: tiles get
    expanded or the query overflows to the strict path — results must not change."""
    d, B, k = 32, 48, 50
    base = synth(9, 64, d)
    X = np.concatenate([np.repeat(base[:8], 300, axis=0), synth(10, 9000, d) * np.float32(3.0)])   # 8 x 300 exact duplicates first
    n = len(X)
    Q = np.concatenate([base[:8] + np.float32(1e-3), synth(11, B - 8, d)])
    for metric in METRICS:
        g = FlatIndex(ctx, d, metric)
        g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
        fast = g.search_batch(Q, k, mode=2)
        assert g.stat("fast_expansions") > 0 or g.stat("fast_overflows") > 0
        same(fast, g.search_batch(Q, k, mode=1))


def test_fast_path_refuses_out_of_range_values(ctx):
    """Values beyond the fp16 range would poison the shadow: the index falls back to the strict kernels
    (auto) and refuses mode=2."""
    n, d = 9000, 32
    X = synth(12, n, d); X[17, 3] = 1e6
    g = FlatIndex(ctx, d, L2_SQUARED)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    Q = synth(13, 40, d)
    auto = g.search_batch(Q, 10)
    assert g.stat("fast_queries") == 0
    same(auto, g.search_batch(Q, 10, mode=1))
    with pytest.raises(CometError):
        g.search_batch(Q, 10, mode=2)


def test_auto_mode_picks_fast_for_big_batches(ctx):
    n, d = 100000, 32
    X = synth(14, n, d)
    g = FlatIndex(ctx, d, COSINE)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    g.search_batch(synth(15, 64, d), 10)
    assert g.stat("fast_queries") > 0
    g.search_batch(synth(15, 4, d), 10)          # small batches too: the fp16 shadow is half the bytes of the exact scan
    assert g.stat("fast_queries") == 4
    g.search_batch(synth(15, 64, d), 100)        # 100000 >= 128 * 4 * 100 rows: still selective enough
    assert g.stat("fast_queries") > 0
    g.search_batch(synth(15, 64, d), 500)        # too few key units per requested result: exact kernel
    assert g.stat("fast_queries") == 0


def test_async_pipeline_equals_sync(ctx):
    """comet_index_search_dev_async / comet_index_search_wait: several searches in flight on the stream give the same
    rows as the synchronous call, including batches whose queries overflow to the strict path at wait time."""
    n, d, B, k = 60000, 64, 128, 5
    base = synth(31, 16, d)
    X = np.concatenate([np.repeat(base[:4], 400, axis=0), synth(32, n - 1600, d)])       # 4 x 400 duplicates -> overflows
    g = FlatIndex(ctx, d, L2_SQUARED)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    batches = [np.concatenate([base[:4] + np.float32(1e-3), synth(40 + i, B - 4, d)]) for i in range(4)]
    want = [g.search_batch(Qb, k, mode=1) for Qb in batches]
    qd = [ctx.alloc(B * d * 4) for _ in batches]
    outs = [(ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)) for _ in batches]
    for p, Qb in zip(qd, batches):
        ctx.upload(p, Qb)
    tickets = [g.search_batch_dev_async(qd[i], B, k, outs[i][0], outs[i][1], outs[i][2], k, mode=2) for i in range(4)]
    for i in (0, 1, 2, 3):
        g.search_wait(tickets[i])
        got = (ctx.download(outs[i][0], (B, k), np.uint32), ctx.download(outs[i][1], (B, k), np.float32), ctx.download(outs[i][2], (B,), np.int32))
        same(got, want[i])
    g.search_wait(tickets[0])          # waiting twice is a no-op
    for p in qd + [x for o in outs for x in o]:
        ctx.free(p)


def test_fast_path_beyond_the_fused_post_stage(ctx):
    """More than 8192 key units (> 1,048,576 rows): the fused post-scan kernel does not apply and the five-launch pipeline
    (tile-key selection, collection, rescoring, selection, gather) runs; results must still equal the strict path and the
    oracle."""
    n, d, B, k = 1_100_000, 16, 32, 10
    X = synth(31, n, d)
    Q = synth(32, B, d)
    g = FlatIndex(ctx, d, L2_SQUARED)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    for lo in range(0, n, 200_000):
        g.add_batch(ids[lo:lo + 200_000], X[lo:lo + 200_000])
    fast = g.search_batch(Q, k, mode=2)
    assert g.stat("fast_queries") == B
    same(fast, g.search_batch(Q, k, mode=1))
    o = orc.Flat(d, L2_SQUARED); assert o.add_batch(ids, X) == 0
    for b in range(4):
        c, oi, os_ = o.search(Q[b], k)
        assert fast[2][b] == c and np.array_equal(fast[0][b, :c], oi) and np.array_equal(fast[1][b, :c].view(np.uint32), os_.view(np.uint32))


def test_wide_rows_and_large_k(ctx):
    """Rows wider than the wave-ingest limit (2048 floats) take the lane-per-row ingest; K = 512 on the fast path keeps
    thousands of candidates per query. Both must equal the strict path / the oracle."""
    n, d = 700, 2500
    X = synth(41, n, d); Q = synth(42, 5, d)
    for metric in (COSINE, L2_SQUARED):
        g = FlatIndex(ctx, d, metric); ids = np.arange(1, n + 1, dtype=np.uint32); g.add_batch(ids, X)
        o = orc.Flat(d, metric); assert o.add_batch(ids, X) == 0
        r = g.search_batch(Q, 7)
        for b in range(len(Q)):
            c, oi, os_ = o.search(Q[b], 7)
            assert r[2][b] == c and np.array_equal(r[0][b, :c], oi) and np.array_equal(r[1][b, :c].view(np.uint32), os_.view(np.uint32))
    n, d, B, k = 300_000, 16, 40, 512
    X = synth(43, n, d); Q = synth(44, B, d)
    g = FlatIndex(ctx, d, L2_SQUARED)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    fast = g.search_batch(Q, k, mode=2)
    assert g.stat("fast_queries") + g.stat("fast_overflows") >= B
    same(fast, g.search_batch(Q, k, mode=1))
