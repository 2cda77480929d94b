"""The int8 shadow of the Flat scan (kernels_fast.hip: per-tile scale, measured residual bound, v_mfma_i32_32x32x32_i8) and the
exact-anchored threshold of the post stage: whatever the data does to the quantiser, results stay bit-identical to the strict path;
data the int8 screen is too coarse for sends the index back to the fp16 shadow."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, FlatIndex

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


def make(ctx, d, metric, policy):
    old = os.environ.get("COMET_FLAT_I8")
    os.environ["COMET_FLAT_I8"] = str(policy)          # read when the index is created
    try:
        return FlatIndex(ctx, d, metric)
    finally:
        if old is None:
            os.environ.pop("COMET_FLAT_I8")
        else:
            os.environ["COMET_FLAT_I8"] = old


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("B", [7, 64, 256])
def test_i8_screen_is_used_and_exact(ctx, metric, B):
    n, d, k = 40000, 96, 10
    X, Q = synth(21, n, d), synth(22, B, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g8, g16 = make(ctx, d, metric, 1), make(ctx, d, metric, 0)
    g8.add_batch(ids, X); g16.add_batch(ids, X)
    r8 = g8.search_batch(Q, k, mode=2)
    assert g8.stat("i8_slices") == 1 and g8.stat("fast_queries") + g8.stat("fast_overflows") == B and g8.stat("i8_max_residual") > 0
    r16 = g16.search_batch(Q, k, mode=2)
    assert g16.stat("i8_slices") == 0
    same(r8, r16)
    same(r8, g8.search_batch(Q, k, mode=1))
    o = orc.Flat(d, metric); o.add_batch(ids, X)
    for q in (0, B - 1):
        cnt, oi, os_ = o.search(Q[q], k)
        assert r8[2][q] == cnt and np.array_equal(r8[0][q, :cnt], oi) and np.array_equal(bits(r8[1][q, :cnt]), bits(os_))


@pytest.mark.parametrize("metric", METRICS)
def test_i8_hostile_rows(ctx, metric):
    """Outlier components (one row dictates its tile's scale), rows of very different magnitude in one tile, non-negative data, constant
    rows, a zero row (L2 family), rows appended in small batches (the open tile is re-quantised when its scale grows)."""
    d, B, k = 80, 130, 25
    rng = np.random.default_rng(5)
    X = synth(31, 30000, d).copy()
    X[100, 3] = 60.0                                        # an outlier component
    X[5000:5256] *= np.float32(1e-3)                        # a tile of tiny rows
    X[7000:7100] = np.abs(X[7000:7100])                     # non-negative rows
    X[9000:9010] = np.float32(0.25)                         # constant rows
    X[12000:12256:2] *= np.float32(40.0)                    # large and small rows interleaved in one tile
    if metric != COSINE:
        X[15000] = 0.0
    Q = np.concatenate([synth(32, B - 6, d), X[[100, 5001, 7005, 9001, 12000, 12001]] * np.float32(1.01)])
    g = make(ctx, d, metric, 1)
    pos = 0
    for m in (1, 255, 1, 300, 4096, 10, 30000):             # ragged batches: tiles are filled in several steps
        m = min(m, len(X) - pos)
        if m <= 0:
            break
        g.add_batch(np.arange(pos + 1, pos + m + 1, dtype=np.uint32), X[pos:pos + m]); pos += m
    assert pos == len(X)
    r = g.search_batch(Q, k, mode=2)
    assert g.stat("i8_slices") == 1
    same(r, g.search_batch(Q, k, mode=1))
    for i in range(1, 4000, 3):
        g.remove(i)
    same(g.search_batch(Q, k, mode=2, document_ids=list(range(2, 30000, 2))), g.search_batch(Q, k, mode=1, document_ids=list(range(2, 30000, 2))))
    g.flush()
    same(g.search_batch(Q, k, mode=2), g.search_batch(Q, k, mode=1))


def test_i8_backs_off_where_it_is_too_coarse(ctx):
    """High-dimensional rows whose distances to a query all sit within the int8 bound of each other: the int8 slice proposes (nearly)
    everything, the index switches to the fp16 shadow for the following searches — and every result is the strict one."""
    n, d, B, k = 90000, 256, 80, 10                         # auto mode screens on int8 from 60 000 + 1500 k rows (wide tile)
    rng = np.random.default_rng(7)
    X = (np.float32(1.0) + np.float32(1e-3) * rng.standard_normal((n, d))).astype(np.float32)   # all rows within ~0.03 of each other
    Q = (np.float32(1.0) + np.float32(1e-3) * rng.standard_normal((B, d))).astype(np.float32)
    g = make(ctx, d, L2_SQUARED, -1)
    g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
    strict = g.search_batch(Q, k, mode=1)
    first = g.search_batch(Q, k, mode=2)
    assert g.stat("i8_slices") == 1 and g.stat("i8_backoffs") == 1
    same(first, strict)
    second = g.search_batch(Q, k, mode=2)
    assert g.stat("i8_slices") == 0                         # fp16 shadow now
    same(second, strict)


def test_anchored_threshold_keeps_ties(ctx):
    """Many rows at exactly the K-th distance (duplicates), more candidates than 2K + 64 under the plain threshold: the anchored
    threshold must keep every tie of the K-th exact distance."""
    d, B, k = 64, 70, 8
    base = synth(41, 40, d)
    X = np.concatenate([np.repeat(base, 30, axis=0), synth(42, 20000, d)])
    ids = np.arange(1, len(X) + 1, dtype=np.uint32)
    Q = np.concatenate([base[:20] * np.float32(1.001), synth(43, B - 20, d)])
    for metric in METRICS:
        g = make(ctx, d, metric, 1)
        g.add_batch(ids, X)
        same(g.search_batch(Q, k, mode=2), g.search_batch(Q, k, mode=1))
        same(g.search_batch(Q, 30, mode=2), g.search_batch(Q, 30, mode=1))


@pytest.mark.parametrize("metric", METRICS)
def test_final_order_cut_with_thresholds(ctx, metric):
    """The post stage's final order only takes the candidates at or under phase 0's measured bound U (round 4). A score threshold below U leaves fewer
    than K results, one between the K-th score and U must not lose any, one above changes nothing; K beyond the anchored phase's rows, K = 1: every
    combination returns the strict path's rows."""
    n, d, B = 120000, 128, 70
    X, Q = synth(61, n, d), synth(62, B, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = make(ctx, d, metric, 1)
    g.add_batch(ids, X)
    for k in (1, 40, 100):
        strict = g.search_batch(Q, k, mode=1)
        fast = g.search_batch(Q, k, mode=2)
        assert g.stat("i8_slices") == 1 and g.stat("fast_queries") + g.stat("fast_overflows") == B and g.stat("fast_candidates") >= B * k
        same(fast, strict)
        sc = strict[1]
        for thr in (float(sc[0, k // 2]), float(sc[0, k - 1]), float(np.nextafter(sc[0, k - 1], np.float32(np.inf))), float(sc[:, k - 1].max()) * 1.01, float(sc[:, 0].min()) * 0.999):
            same(g.search_batch(Q, k, mode=2, threshold=thr), g.search_batch(Q, k, mode=1, threshold=thr))


# ---------------------------------------------------------------------------------------------- the IVF list scan's int8 shadow
def make_ivf(ctx, d, nlist, metric, policy):
    from comet_amd import IVFIndex
    old = os.environ.get("COMET_IVF_I8")
    os.environ["COMET_IVF_I8"] = str(policy)
    try:
        return IVFIndex(ctx, d, nlist, metric)
    finally:
        if old is None:
            os.environ.pop("COMET_IVF_I8")
        else:
            os.environ["COMET_IVF_I8"] = old


@pytest.mark.parametrize("metric", METRICS)
def test_ivf_i8_screen_is_used_and_exact(ctx, metric):
    n, d, nlist, B, k = 30000, 200, 48, 150, 12         # d = 200: rows padded to 256 codes (two 128-wide K steps)
    centers = synth(51, 40, d)
    X = (centers[np.arange(n) % 40] + synth(52, n, d) * np.float32(0.2)).astype(np.float32)
    X[77] *= np.float32(30.0)                            # a row that dictates its unit's scale
    Q = (centers[np.arange(B) % 40] + synth(53, B, d) * np.float32(0.2)).astype(np.float32)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    res = {}
    for pol in (1, 0):
        g = make_ivf(ctx, d, nlist, metric, pol)
        g.train(X[:6000]); g.add_batch(ids, X)
        for npb in (1, 6, 0):
            r = g.search_batch(Q, k, nprobes=npb, mode=2)
            assert g.stat("i8_slices") == (1 if pol else 0), (pol, npb)
            assert g.stat("fast_queries") + g.stat("fast_overflows") == B
            same(r, g.search_batch(Q, k, nprobes=npb, mode=1))
            if pol:
                res[npb] = r
            else:
                same(r, res[npb])
        if pol:                                          # deletes + filter + re-layout after more rows (the shadow is rebuilt with the slot layout)
            for i in range(1, 3000, 5):
                g.remove(i)
            flt = list(range(2, n, 3))
            same(g.search_batch(Q, k, nprobes=6, mode=2, document_ids=flt), g.search_batch(Q, k, nprobes=6, mode=1, document_ids=flt))
            g.add_batch(np.arange(n + 1, n + 501, dtype=np.uint32), X[:500] * np.float32(1.5))
            r = g.search_batch(Q, k, nprobes=6, mode=2)
            assert g.stat("i8_slices") == 1
            same(r, g.search_batch(Q, k, nprobes=6, mode=1))
