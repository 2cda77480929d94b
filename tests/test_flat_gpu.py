"""Parity of the GPU Flat path (C ABI -> HIP kernels) against the CPU oracle, bit-exact ids and scores.
Reference behaviour: flat_index.go, flat_index_search.go; fixtures from flat_index_search_test.go."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, CometError, FlatIndex, ZeroVectorError

pytestmark = pytest.mark.gpu
KATS = json.loads((Path(__file__).parent / "golden" / "reference_kats.json").read_text())
METRICS = [EUCLIDEAN, L2_SQUARED, COSINE]


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def build_pair(ctx, metric, X, ids=None):
    n, d = X.shape
    ids = np.arange(1, n + 1, dtype=np.uint32) if ids is None else np.asarray(ids, np.uint32)
    g = FlatIndex(ctx, d, metric)
    g.add_batch(ids, X)
    o = orc.Flat(d, metric)
    assert o.add_batch(ids, X) == 0
    return g, o


def assert_same(g, o, Q, k, **kw):
    ids, sc, cnt = g.search_batch(Q, k, threshold=kw.get("threshold", 0.0), document_ids=kw.get("filter_ids", ()))
    for b, q in enumerate(Q):
        n, oi, os_ = o.search(q, k, threshold=kw.get("threshold", 0.0), filter_ids=kw.get("filter_ids", ()))
        assert cnt[b] == n, (b, cnt[b], n)
        assert np.array_equal(ids[b, :n], oi), (b, ids[b, :n], oi)
        assert np.array_equal(sc[b, :n].view(np.uint32), os_.view(np.uint32)), (b, sc[b, :n], os_)


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,d,B,k", [(1000, 128, 5, 10), (777, 17, 3, 7), (5000, 100, 20, 25), (300, 3, 1, 300), (4097, 64, 33, 100)])
def test_flat_matches_oracle_bit_exact(ctx, metric, n, d, B, k):
    X = synth(0xC0FFEE + n, n, d)
    Q = synth(0xBEEF + n, B, d)
    g, o = build_pair(ctx, metric, X)
    assert len(g) == n
    assert_same(g, o, Q, k)


def test_config1_flat_l2sq_10k_x128_k10(ctx):
    """BASELINE config 1: comet.NewFlatIndex(128, L2Squared), 10k random vectors, K=10."""
    X = synth(0xC0FFEE + 1, 10_000, 128)
    Q = synth(0xBEEF + 1, 16, 128)
    g, o = build_pair(ctx, L2_SQUARED, X)
    assert_same(g, o, Q, 10)
    # and through the reference-shaped fluent API, one query per Execute()
    res = g.new_search().with_query(Q[0]).with_k(10).execute()
    n, oi, os_ = o.search(Q[0], 10)
    assert [r.id for r in res] == oi.tolist()
    assert np.array_equal(np.array([r.score for r in res], np.float32).view(np.uint32), os_.view(np.uint32))


@pytest.mark.parametrize("metric", METRICS)
def test_threshold_filter_softdelete_flush(ctx, metric):
    n, d = 2000, 48
    X = synth(7, n, d)
    Q = synth(8, 6, d)
    g, o = build_pair(ctx, metric, X)
    # threshold (active only if > 0, flat_index_search.go:269)
    ref = o.search(Q[0], 50)[2]
    thr = float(ref[20])
    assert_same(g, o, Q, 50, threshold=thr)
    assert_same(g, o, Q, 50, threshold=-1.0)
    # WithDocumentIDs pre-filter (document_filter.go) — includes ids that do not exist
    flt = list(range(5, 900, 3)) + [10_000_000]
    assert_same(g, o, Q, 15, filter_ids=flt)
    # soft delete, then flush
    for i in (1, 17, 500, 1999, 2000):
        g.remove(i); assert o.remove(i) == 0
    with pytest.raises(CometError):
        g.remove(17)                      # already deleted (flat_index.go:236)
    with pytest.raises(CometError):
        g.remove(999_999)                 # not found (flat_index.go:233)
    assert_same(g, o, Q, 15)
    assert_same(g, o, Q, 15, filter_ids=flt, threshold=thr)
    g.flush(); o.flush()
    assert len(g) == n - 5
    assert_same(g, o, Q, 15)


def test_ties_are_broken_by_scan_order(ctx):
    """Duplicate vectors -> equal scores; canonical order is insertion order (what a stable sort gives)."""
    base = synth(3, 40, 16)
    X = np.concatenate([base, base, base])           # every vector three times
    ids = np.arange(100, 100 + len(X), dtype=np.uint32)
    g, o = build_pair(ctx, L2_SQUARED, X, ids)
    Q = base[:4] + np.float32(0.25)
    assert_same(g, o, Q, 7)
    assert_same(g, o, Q, 120)                        # k == n: full ordering
    # reference-style integer-pattern data: vec[j] = (i*dim+j) % 10 (pq_index_search_test.go:20-26): massive ties
    n, d = 600, 8
    P = np.array([[(i * d + j) % 10 for j in range(d)] for i in range(n)], np.float32)
    g, o = build_pair(ctx, EUCLIDEAN, P)
    assert_same(g, o, P[:3], 50)


def test_reference_fixtures_through_the_fluent_api(ctx):
    b = KATS["flat_search_simple"]
    idx = FlatIndex(ctx, b["dim"], b["metric"])
    vecs = np.asarray(b["vectors"], np.float32)
    for i, v in enumerate(vecs):
        idx.add(i + 1, v.copy())
    res = idx.new_search().with_query(b["query"]).with_k(b["k"]).execute()
    assert len(res) == b["expected_len"] and list(vecs[res[0].id - 1]) == b["first_vector"]

    b = KATS["flat_search_threshold"]
    idx = FlatIndex(ctx, b["dim"], b["metric"])
    for i, v in enumerate(b["vectors"]):
        idx.add(i + 1, v)
    res = idx.new_search().with_query(b["query"]).with_k(b["k"]).with_threshold(b["threshold"]).execute()
    assert len(res) == b["expected_len"]

    b = KATS["flat_search_cosine"]
    idx = FlatIndex(ctx, b["dim"], b["metric"])
    vecs = [np.asarray(v, np.float32) for v in b["vectors"]]
    for i, v in enumerate(vecs):
        idx.add(i + 1, v)                            # normalises the caller's array in place (flat_index.go:182)
    assert np.allclose(np.linalg.norm(vecs[1]), 1.0, atol=1e-6)
    res = idx.new_search().with_query(b["query"]).with_k(b["k"]).execute()
    assert len(res) == 1 and np.allclose(vecs[res[0].id - 1], b["first_vector"], atol=b["vector_tolerance"])

    b = KATS["flat_search_k_bounds"]
    idx = FlatIndex(ctx, b["dim"], b["metric"])
    for i, v in enumerate(b["vectors"]):
        idx.add(i + 1, v)
    for c in b["cases"]:
        assert len(idx.new_search().with_query(b["query"]).with_k(c["k"]).execute()) == c["len"], c

    b = KATS["flat_search_ordered"]
    idx = FlatIndex(ctx, b["dim"], b["metric"])
    vecs = np.asarray(b["vectors"], np.float32)
    for i, v in enumerate(vecs):
        idx.add(i + 1, v)
    res = idx.new_search().with_query(b["query"]).with_k(b["k"]).execute()
    assert [float(vecs[r.id - 1][0]) for r in res] == b["expected_first_coords"]


def test_validation_and_errors(ctx):
    idx = FlatIndex(ctx, 4, COSINE)
    with pytest.raises(ValueError, match="must specify either queries or node IDs"):
        idx.new_search().with_k(3).execute()
    assert idx.new_search().with_query([1, 0, 0, 0]).execute() == []      # empty index
    idx.add(1, [1, 0, 0, 0]); idx.add(2, [0, 1, 0, 0])
    with pytest.raises(ValueError, match="query dimension mismatch: expected 4, got 3"):
        idx.new_search().with_query([1, 0, 0]).execute()
    with pytest.raises(ZeroVectorError):
        idx.new_search().with_query([0, 0, 0, 0]).execute()               # ErrZeroVector from Preprocess(query)
    with pytest.raises(ZeroVectorError):
        idx.add(3, [0, 0, 0, 0])                                          # ErrZeroVector from Add
    assert len(idx) == 2
    # a batch stops at the first zero vector, keeping what came before it (n sequential Add calls)
    with pytest.raises(ZeroVectorError):
        idx.add_batch([10, 11, 12], [[1, 1, 0, 0], [0, 0, 0, 0], [0, 0, 1, 1]])
    assert len(idx) == 3 and idx.last_added == 1
    with pytest.raises(ValueError):
        FlatIndex(ctx, 0, COSINE)


def test_multi_query_aggregation_and_node_queries(ctx):
    """Several queries in one Execute() aggregate by node id (flat_index_search.go:143-153,
    flat_index_search_test.go:230-278): sum by default."""
    X = synth(21, 200, 12)
    g, o = build_pair(ctx, L2_SQUARED, X)
    q1, q2 = X[3] + np.float32(0.1), X[9] - np.float32(0.05)
    res = g.new_search().with_query(q1, q2).with_k(5).execute()
    per = {}
    for q in (q1, q2):
        n, oi, os_ = o.search(q, 5)
        for i, s in zip(oi, os_):
            per.setdefault(int(i), []).append(np.float32(s))
    want = sorted(((np.float32(sum(v, np.float32(0))), i) for i, v in per.items()))[:5]
    assert len(res) == 5 and all(ids == {r.id for r in res} for ids in [{i for _, i in want}])
    assert [r.score for r in res] == [s for s, _ in want]
    assert all(res[i].score <= res[i + 1].score for i in range(4))
    # WithNode: the node's own stored vector is the query -> it is its own nearest neighbour
    res = g.new_search().with_node(42).with_k(1).execute()
    assert res[0].id == 42 and res[0].score == 0.0


def test_distance_singletons_bit_exact(ctx):
    rng = np.random.default_rng(5)
    for d in (1, 2, 3, 31, 128, 769):
        a = rng.standard_normal(d).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        for m in METRICS:
            if m == COSINE:
                a2, b2 = orc.normalize(a), orc.normalize(b)
            else:
                a2, b2 = a, b
            assert np.float32(ctx.distance(m, a2, b2)).view(np.uint32) == orc.distance(m, a2, b2).view(np.uint32)
        pg = ctx.preprocess(COSINE, a)
        po, _ = orc.preprocess("cosine", a)
        assert np.array_equal(pg.view(np.uint32), po.view(np.uint32))
    for blk, m in (("euclidean_calculate", EUCLIDEAN), ("l2squared_calculate", L2_SQUARED), ("cosine_calculate", COSINE)):
        for c in KATS[blk]["cases"]:
            assert abs(float(ctx.distance(m, c["a"], c["b"])) - c["expected"]) <= KATS["distance_epsilon"]
    with pytest.raises(ZeroVectorError):
        ctx.preprocess(COSINE, [0, 0, 0])
    # sqrt is correctly rounded for awkward inputs (L2 = float32(math.Sqrt(float64(sum))))
    vals = np.concatenate([rng.random(4096).astype(np.float32) * np.float32(1e-3), rng.random(4096).astype(np.float32) * np.float32(1e6)])
    pairs = vals.reshape(-1, 2)                                                # sqrt(a*a + b*b), 4096 awkward sums
    got = ctx.distance_batch(EUCLIDEAN, pairs, np.zeros(2, np.float32))
    assert np.array_equal(got.view(np.uint32), orc.distance_batch("l2", pairs, np.zeros(2, np.float32)).view(np.uint32))


def test_synth_fill_matches_oracle_stream(ctx):
    n = 100_003
    p = ctx.alloc(n * 4)
    ctx.synth_fill(p, 0xC0FFEE, 12345, n)
    ctx.sync()
    got = ctx.download(p, (n,), np.float32)
    ctx.free(p)
    assert np.array_equal(got.view(np.uint32), orc.synth(0xC0FFEE, 12345, n).view(np.uint32))


def test_synth_mixture_matches_oracle(ctx):
    rows, dim = 777, 96
    p = ctx.alloc(rows * dim * 4)
    for centers, sub in ((64, 0), (64, 4096), (0, 0)):
        ctx.synth_mixture(p, 0xC0FFEE + 5, centers, 0.15, sub, 0.02, 123456, rows, dim)
        ctx.sync()
        got = ctx.download(p, (rows, dim), np.float32)
        assert np.array_equal(got.view(np.uint32), orc.synth_mixture(0xC0FFEE + 5, centers, 0.15, sub, 0.02, 123456, rows, dim).view(np.uint32))
    ctx.free(p)


def test_long_rows_sample_bound_and_its_generic_fallback(ctx):
    """Rows longer than 2 x 8192 candidates take the sample-bound selection (bound from the row's prefix, one filtering
    pass); orders that defeat the bound — distances DESCENDING along the row, or every distance equal — must fall
    through to the multi-pass radix path on the device and still give the canonical (score, position) order."""
    n, d = 40_000, 8
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((6, d)).astype(np.float32)
    g, o = build_pair(ctx, L2_SQUARED, X)
    for k in (1, 10, 64):
        ids, sc, cnt = g.search_batch(Q, k, mode=1)
        for b, q in enumerate(Q):
            m, oi, os_ = o.search(q, k)
            assert cnt[b] == m and np.array_equal(ids[b, :m], oi) and np.array_equal(sc[b, :m].view(np.uint32), os_.view(np.uint32))
    # threshold + soft deletes on the bound path
    ref = o.search(Q[0], 40)[2]
    assert_same(g, o, Q, 25, threshold=float(ref[17]))
    for i in o.search(Q[1], 5)[1].tolist():
        g.remove(int(i)); assert o.remove(int(i)) == 0
    assert_same(g, o, Q, 10)
    # adversarial: the query is the origin and |x_i| decreases with i -> the best candidates are the LAST ones
    X2 = np.zeros((n, d), np.float32); X2[:, 0] = np.linspace(1000.0, 1.0, n, dtype=np.float32)
    g2, o2 = build_pair(ctx, L2_SQUARED, X2)
    Z = np.zeros((3, d), np.float32); Z[1, 0] = 1.0; Z[2, 0] = 500.0
    for k in (1, 10, 100):
        ids, sc, cnt = g2.search_batch(Z, k, mode=1)
        for b, q in enumerate(Z):
            m, oi, os_ = o2.search(q, k)
            assert cnt[b] == m and np.array_equal(ids[b, :m], oi) and np.array_equal(sc[b, :m].view(np.uint32), os_.view(np.uint32))
    # all distances equal: ties resolved by scan position, by the generic path
    X3 = np.ones((n, d), np.float32)
    g3, o3 = build_pair(ctx, L2_SQUARED, X3)
    ids, sc, cnt = g3.search_batch(Z[:1], 10, mode=1)
    assert ids[0, :10].tolist() == list(range(1, 11)) == o3.search(Z[0], 10)[1].tolist()


def test_norm_normalize_scale_bit_exact(ctx):
    """Norm / Normalize / Scale (distance.go:312-428) through comet_norm_batch / comet_normalize_batch / comet_scale_batch: bit-identical to
    the oracle's restatement — serial float32 sum, float32(sqrt(float64)), multiplication by 1 / norm; a zero vector comes back unchanged
    from Normalize; the reference's documented examples (Norm([3, 4]) = 5; Scale([1, 2, 3], 2) = [2, 4, 6])."""
    import json
    from pathlib import Path
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    kats = json.loads((Path(__file__).parent / "golden" / "reference_kats.json").read_text())
    for cse in kats["norm"]["cases"]:                  # distance_test.go:533-575 (TestNorm)
        assert abs(float(ctx.norm(np.array(cse["v"], np.float32))) - cse["expected"]) <= 1e-6
    for cse in kats["normalize"]["cases"]:             # distance_test.go:640-690 (TestNormalize)
        assert np.allclose(ctx.normalize(np.array(cse["v"], np.float32)), np.array(cse["expected"], np.float32), atol=1e-6)
    for cse in kats["scale"]["cases"]:                 # distance_test.go:577-600 (TestScale)
        assert ctx.scale(np.array(cse["v"], np.float32), cse["s"]).tolist() == cse["expected"]
    rng = np.random.default_rng(15)
    assert ctx.norm(np.array([3, 4], np.float32)) == np.float32(5.0)
    assert ctx.scale(np.array([1, 2, 3], np.float32), 2.0).tolist() == [2.0, 4.0, 6.0]
    assert ctx.scale(np.array([1, 2, 3], np.float32), -1.0).tolist() == [-1.0, -2.0, -3.0]
    z = np.zeros(7, np.float32)
    assert ctx.norm(z) == 0.0 and np.array_equal(ctx.normalize(z), z)
    for d in (1, 2, 3, 31, 128, 769):
        X = (rng.standard_normal((37, d)) * rng.choice([1e-20, 1e-3, 1.0, 1e6, 1e18], (37, 1))).astype(np.float32)
        X[5] = 0.0
        gn, gz = ctx.norm(X), ctx.normalize(X)
        for i in range(len(X)):
            assert bits(gn[i]) == bits(orc.norm(X[i])), (d, i)
            assert np.array_equal(bits(gz[i]), bits(orc.normalize(X[i]))), (d, i)
        for sc in (0.0, -1.0, 0.3333333, 1e30, float("inf")):
            with np.errstate(all="ignore"):
                gs = ctx.scale(X, sc)
            for i in (0, 5, 36):
                assert np.array_equal(bits(gs[i]), bits(orc.scale(X[i], sc))), (d, i, sc)
    assert np.array_equal(bits(ctx.normalize(X[3])), bits(orc.normalize(X[3])))       # single-vector form
