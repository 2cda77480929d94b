"""Parity of k-means, IVF, PQ and IVFPQ (train / add / search) on the GPU against the CPU oracle.
Everything is compared bit-exactly: centroids, codebooks, list membership, PQ codes, result ids and scores.
Reference: clustering.go, ivf_index*.go, pq_index*.go, ivfpq_index*.go; fixture shapes from
ivfpq_index_search_test.go:9-72 and pq_index_search_test.go:9-53."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, CometError, IVFIndex, IVFPQIndex, PQIndex

pytestmark = pytest.mark.gpu
KATS = json.loads((Path(__file__).parent / "golden" / "reference_kats.json").read_text())
METRICS = [EUCLIDEAN, L2_SQUARED, COSINE]


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def clustered(seed, n, d, ncl, sigma=0.15):
    """mixture of `ncl` gaussian-ish blobs built from the deterministic synthetic stream"""
    centers = synth(seed, ncl, d)
    noise = synth(seed + 1, n, d) * np.float32(sigma)
    return (centers[np.arange(n) % ncl] + noise).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def check_search(g, o, Q, k, nprobes=0, **kw):
    ids, sc, cnt = g.search_batch(Q, k, nprobes=(nprobes or 0), threshold=kw.get("threshold", 0.0), document_ids=kw.get("filter_ids", ()))
    for b, q in enumerate(Q):
        if nprobes is None:
            n, oi, os_ = o.search(q, k, threshold=kw.get("threshold", 0.0), filter_ids=kw.get("filter_ids", ()))
        else:
            n, oi, os_ = o.search(q, k, nprobes, threshold=kw.get("threshold", 0.0), filter_ids=kw.get("filter_ids", ()))
        assert cnt[b] == n, (b, cnt[b], n)
        assert np.array_equal(ids[b, :n], oi), (b, ids[b, :n], oi)
        assert np.array_equal(bits(sc[b, :n]), bits(os_)), (b, sc[b, :n], os_)


# ---------------------------------------------------------------------------------------------- k-means
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,d,k", [(500, 16, 8), (1500, 40, 33), (300, 3, 300), (2100, 96, 64)])
def test_kmeans_bit_exact(ctx, metric, n, d, k):
    X = clustered(11 + n, n, d, max(2, k // 2))
    gc, ga = ctx.kmeans(X, k, metric, 20)
    oc, oa = orc.kmeans(X, k, metric, 20)
    assert gc.shape == oc.shape
    assert np.array_equal(ga, oa)
    assert np.array_equal(bits(gc), bits(oc))


def test_kmeans_reference_fixtures(ctx):
    for name in ("kmeans_basic", "kmeans_convergence", "kmeans_centroid_accuracy"):
        b = KATS[name]
        gc, ga = ctx.kmeans(b["vectors"], b["k"], L2_SQUARED, b.get("max_iter", 20))
        oc, oa = orc.kmeans(b["vectors"], b["k"], "l2_squared", b.get("max_iter", 20))
        assert np.array_equal(ga, oa) and np.array_equal(bits(gc), bits(oc))
        for grp in b.get("same_cluster_groups", []):
            assert len({int(ga[i]) for i in grp}) == 1
    b = KATS["kmeans_k_greater_than_n"]
    gc, ga = ctx.kmeans(b["vectors"], b["k"])
    assert gc.shape[0] == b["expected_k"] and len(set(ga.tolist())) == b["unique_clusters"]
    assert ctx.kmeans(np.zeros((0, 2), np.float32), 2) == (None, None)
    assert ctx.kmeans([[1.0, 2.0]], 0) == (None, None)
    # max_iter <= 0 -> DefaultMaxIter
    X = clustered(5, 400, 8, 5)
    assert np.array_equal(bits(ctx.kmeans(X, 5, L2_SQUARED, 0)[0]), bits(orc.kmeans(X, 5, "l2_squared", 0)[0]))
    # iteration cap honoured
    assert np.array_equal(bits(ctx.kmeans(X, 7, L2_SQUARED, 2)[0]), bits(orc.kmeans(X, 7, "l2_squared", 2)[0]))


def test_nearest_centroid(ctx):
    b = KATS["nearest_centroid_tie"]
    assert ctx.nearest_centroid(b["v"], b["centroids"], L2_SQUARED)[0] == b["source_says"]
    X, C = synth(1, 700, 24), synth(2, 50, 24)
    for m in METRICS:
        got = ctx.nearest_centroid(X, C, m)
        want = [orc.nearest_centroid(x, C, m) for x in X]
        assert got.tolist() == want


# ---------------------------------------------------------------------------------------------- IVF
def build_ivf(ctx, metric, X, train, nlist):
    n, d = X.shape
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = IVFIndex(ctx, d, nlist, metric)
    o = orc.IVF(d, metric, nlist)
    g.train(train); assert o.train(train) == 0
    assert g.trained()
    assert np.array_equal(bits(g.centroids(nlist)), bits(o.centroids()))
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    assert [g.list_size(l) for l in range(nlist)] == o.list_sizes()
    return g, o


@pytest.mark.parametrize("metric", METRICS)
def test_ivf_coarse_fast_ranking_matches_exact(ctx, metric):
    """nlist >= 64 and nprobes <= nlist / 4: the coarse quantiser ranks the centroids with approximate (float32 FMA) distances
    and re-scores exactly only the ones that can be among the nprobes nearest (coarse_dot / coarse_pick kernels); the probe
    lists — and with them every result — must equal the exact ranking's (oracle), incl. queries sitting ON a centroid and
    queries far outside the data."""
    n, d, nlist = 9000, 40, 256
    X = synth(71, n, d) * np.float32(2.0)
    g, o = build_ivf(ctx, metric, X, X[:5120], nlist)
    C = g.centroids(nlist)
    Q = np.vstack([synth(72, 24, d), X[:4], C[5:7], synth(73, 2, d) * np.float32(50.0)])
    for nprobes in (1, 8, 64, 65):                        # 65 > nlist / 4: the exact ranking
        check_search(g, o, Q, 10, nprobes)
    check_search(g, o, Q, 0, 3)


@pytest.mark.parametrize("d", [256, 288, 800])
def test_ivf_coarse_product_k_splits(ctx, d):
    """The coarse product runs on MFMA with K split over up to four workgroups per tile when there are at least eight 32-wide K chunks
    (coarse_dot_mfma_kernel; planes added by coarse_pick_kernel): 8 chunks (4 x 2), 9 (3 x 3, uneven tail), 25 (4 splits of 7, 7, 7, 4)."""
    n, nlist = 5000, 64
    X = clustered(91 + d, n, d, 40, 0.3)
    g, o = build_ivf(ctx, L2_SQUARED, X, X[:2000], nlist)
    Q = np.vstack([clustered(92 + d, 10, d, 40, 0.3), X[:3], g.centroids(nlist)[2:4]])
    for nprobes in (1, 5, 16):
        check_search(g, o, Q, 10, nprobes)


@pytest.mark.parametrize("metric", METRICS)
def test_ivf_matches_oracle(ctx, metric):
    n, d, nlist = 3000, 48, 24
    X = clustered(31, n, d, 20)
    Q = clustered(31, 12, d, 20) + np.float32(0.01)
    g, o = build_ivf(ctx, metric, X, X[:1200], nlist)
    assert g.default_nprobes() == 4                       # floor(sqrt(24)) ivf_index.go:406-413
    for nprobes in (1, 4, 24, 0, 100):                    # <=0 or > nlist -> nlist (ivf_index_search.go:233-236)
        check_search(g, o, Q, 10, nprobes)
    check_search(g, o, Q, 0, 3)                           # k = 0 -> all candidates of the probed lists
    ref = o.search(Q[0], 40, 6)[2]
    check_search(g, o, Q, 40, 6, threshold=float(ref[15]))
    check_search(g, o, Q, 15, 6, filter_ids=list(range(1, 1500, 2)))
    for i in (5, 6, 1000, 2999):
        g.remove(i); assert o.remove(i) == 0
    check_search(g, o, Q, 15, 6)
    with pytest.raises(CometError):
        g.remove(5)
    # fluent API uses the index default nprobes
    res = g.new_search().with_query(Q[1]).with_k(5).execute()
    n_, oi, os_ = o.search(Q[1], 5, 4)
    assert [r.id for r in res] == oi.tolist()


def test_ivf_untrained_and_training_errors(ctx):
    g = IVFIndex(ctx, 8, 10, L2_SQUARED)
    with pytest.raises(RuntimeError, match="index must be trained before searching"):
        g.new_search().with_query(np.zeros(8, np.float32)).execute()
    with pytest.raises(CometError, match="need at least 10 training vectors"):
        g.train(synth(1, 5, 8))
    with pytest.raises(CometError, match="must be trained"):
        g.add(1, np.ones(8, np.float32))
    with pytest.raises(ValueError):
        IVFIndex(ctx, 8, 0, L2_SQUARED)


# ---------------------------------------------------------------------------------------------- PQ
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("d,M,nbits", [(32, 8, 4), (48, 6, 6), (20, 5, 3), (64, 64, 2)])
def test_pq_matches_oracle(ctx, metric, d, M, nbits):
    n = 1200
    X = clustered(41, n, d, 30)
    Q = clustered(41, 7, d, 30) + np.float32(0.02)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = PQIndex(ctx, d, metric, M, nbits)
    o = orc.PQ(d, metric, M, nbits)
    g.train(X[:600]); assert o.train(X[:600]) == 0
    assert np.array_equal(bits(g.codebooks(M, 1 << nbits, d // M)), bits(o.codebooks()))
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    gi, gcodes, _ = g.list_read(0, codes_width=M)
    assert np.array_equal(gi, ids) and np.array_equal(gcodes, o.codes())
    check_search(g, o, Q, 10, None)
    check_search(g, o, Q, 0, None)                        # all results
    ref = o.search(Q[0], 50)[2]
    check_search(g, o, Q, 50, None, threshold=float(ref[10]))
    check_search(g, o, Q, 9, None, filter_ids=list(range(100, 400)))
    for i in (1, 2, 777):
        g.remove(i); assert o.remove(i) == 0
    check_search(g, o, Q, 9, None)
    g.flush()
    assert len(g) == n - 3


def test_adc_fused_filter_edges(ctx):
    """The ADC scan's fused top-K filter (K <= 64) on a PQ index whose single list spans several 8192-code segments: odd batch
    sizes (a duo with a hole), K at and beyond the fused limit, duplicated vectors (equal distances: the tie order is the scan
    position), thresholds, document filters, soft deletes — and a corpus of identical vectors, where every candidate ties with
    the bound and all of them survive the filter (select-from-composites' radix path). Always the oracle's ids and scores."""
    n, d, M, nbits = 21000, 32, 8, 6
    X = clustered(61, n, d, 40)
    X[9000:9300] = X[10:310]                               # 300 exact duplicates in another segment
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = PQIndex(ctx, d, L2_SQUARED, M, nbits); o = orc.PQ(d, L2_SQUARED, M, nbits)
    g.train(X[:2000]); assert o.train(X[:2000]) == 0
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    for B in (1, 3, 8):
        Q = np.vstack([clustered(62, B - 1, d, 40) + np.float32(0.01), X[10:11]]) if B > 1 else X[10:11].copy()
        for k in (1, 10, 64, 65):
            check_search(g, o, Q, k, None)
    Q = np.vstack([clustered(63, 4, d, 40), X[12:13]])
    ref = o.search(Q[0], 200)[2]
    check_search(g, o, Q, 20, None, threshold=float(ref[7]))              # fewer than k within the threshold
    check_search(g, o, Q, 12, None, filter_ids=list(range(5, 9500, 3)))
    for i in (11, 13, 9011, 20999):
        g.remove(i); assert o.remove(i) == 0
    check_search(g, o, Q, 12, None)
    # mass ties
    Y = np.repeat(X[:1], 9000, axis=0)
    g2 = PQIndex(ctx, d, L2_SQUARED, M, nbits); o2 = orc.PQ(d, L2_SQUARED, M, nbits)
    g2.train(X[:2000]); assert o2.train(X[:2000]) == 0
    g2.add_batch(ids[:9000], Y); assert o2.add_batch(ids[:9000], Y) == 0
    check_search(g2, o2, Q[:3], 10, None)
    check_search(g2, o2, Q[:3], 64, None)


def test_pq_reference_fixture_shape(ctx):
    """pq_index_search_test.go:9-53: dim 8, M 4, nbits 4 (Ksub 16), vec[j] = (i*dim+j) % 10 — heavy ties."""
    d, M, nbits, n = 8, 4, 4, 100
    X = np.array([[(i * d + j) % 10 for j in range(d)] for i in range(n)], np.float32)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g, o = PQIndex(ctx, d, EUCLIDEAN, M, nbits), orc.PQ(d, "l2", M, nbits)
    g.train(X); assert o.train(X) == 0
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    assert np.array_equal(g.list_read(0, codes_width=M)[1], o.codes())
    check_search(g, o, X[:5], 5, None)
    check_search(g, o, X[:5], 100, None)
    # errors
    with pytest.raises(CometError, match="need at least 16 vectors for training"):
        PQIndex(ctx, d, EUCLIDEAN, M, nbits).train(X[:10])
    with pytest.raises(RuntimeError, match="index not trained"):
        PQIndex(ctx, d, EUCLIDEAN, M, nbits).new_search().with_query(X[0]).execute()
    with pytest.raises(ValueError, match="must be divisible"):
        PQIndex(ctx, 10, EUCLIDEAN, 4, 8)
    with pytest.raises(ValueError, match="Nbits"):
        PQIndex(ctx, 8, EUCLIDEAN, 4, 17)
    assert PQIndex(ctx, d, EUCLIDEAN, M, nbits).train(X) is None
    g2 = PQIndex(ctx, d, EUCLIDEAN, M, nbits); g2.train(X)
    assert g2.new_search().with_query(X[0]).execute() == []           # trained but empty (pq_index_search.go:232-234)


# ---------------------------------------------------------------------------------------------- IVFPQ
def build_ivfpq(ctx, metric, X, train, nlist, M, nbits):
    n, d = X.shape
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = IVFPQIndex(ctx, d, metric, nlist, M, nbits)
    o = orc.IVFPQ(d, metric, nlist, M, nbits)
    g.train(train); assert o.train(train) == 0
    assert np.array_equal(bits(g.centroids(nlist)), bits(o.centroids()))
    assert np.array_equal(bits(g.codebooks(M, 1 << nbits, d // M)), bits(o.codebooks()))
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    for l in range(nlist):
        gi, gc, _ = g.list_read(l, codes_width=M)
        assert np.array_equal(gi, o.list_ids(l)), l
        assert np.array_equal(gc, o.list_codes(l)), l
    return g, o


@pytest.mark.parametrize("metric", METRICS)
def test_ivfpq_matches_oracle(ctx, metric):
    n, d, nlist, M, nbits = 2500, 32, 12, 8, 5
    X = clustered(51, n, d, 16)
    Q = clustered(51, 9, d, 16) + np.float32(0.015)
    g, o = build_ivfpq(ctx, metric, X, X[:1000], nlist, M, nbits)
    assert g.default_nprobes() == 3
    for nprobes in (1, 3, 12, 0, 50):
        check_search(g, o, Q, 10, nprobes)
    check_search(g, o, Q, 0, 2)
    ref = o.search(Q[0], 60, 4)[2]
    check_search(g, o, Q, 60, 4, threshold=float(ref[20]))
    check_search(g, o, Q, 12, 4, filter_ids=list(range(3, 2000, 3)))
    for i in (9, 10, 11, 2500):
        g.remove(i); assert o.remove(i) == 0
    check_search(g, o, Q, 12, 4)
    check_search(g, o, Q, 12, 4, filter_ids=list(range(3, 2000, 3)), threshold=float(ref[20]))
    g.flush()
    assert len(g) == n - 4
    check_search(g, o, Q, 12, 4)


def test_ivfpq_reference_fixture_shape(ctx):
    """ivfpq_index_search_test.go:9-72: dim 8, nlist 2, M 4, nbits 4, 100 ramp training vectors vec[j] = i*dim + j."""
    d, nlist, M, nbits, n = 8, 2, 4, 4, 100
    X = np.array([[i * d + j for j in range(d)] for i in range(n)], np.float32)
    g, o = build_ivfpq(ctx, L2_SQUARED, X, X, nlist, M, nbits)
    check_search(g, o, X[:6] + np.float32(0.5), 5, 1)
    check_search(g, o, X[:6] + np.float32(0.5), 10, 2)
    with pytest.raises(CometError, match="need at least 20 vectors for training"):
        IVFPQIndex(ctx, d, L2_SQUARED, nlist, M, nbits).train(X[:19])
    with pytest.raises(RuntimeError, match="index must be trained before searching"):
        IVFPQIndex(ctx, d, L2_SQUARED, nlist, M, nbits).new_search().with_query(X[0]).execute()


def test_ivfpq_wide_codes_m96(ctx):
    """The C4 code shape (M = 96, nbits = 8, dsub = 8 -> 96 KiB lookup table in LDS) at a size the oracle can train."""
    n, d, nlist, M, nbits = 3000, 768, 8, 96, 8
    X = clustered(61, n, d, 12, sigma=0.3)
    Q = clustered(61, 4, d, 12, sigma=0.3) + np.float32(0.01)
    g, o = build_ivfpq(ctx, L2_SQUARED, X, X[:600], nlist, M, nbits)
    check_search(g, o, Q, 10, 3)
    check_search(g, o, Q, 10, 8)


def test_ivfpq_table_sub_batches(ctx):
    """360 queries x 32 probes x 96 KiB of tables exceed the 1 GiB table budget: the batch goes through the table / scan kernels
    in sub-batches, each with its own slots, queues and filter bounds."""
    n, d, nlist, M, nbits = 3000, 768, 32, 96, 8
    X = clustered(61, n, d, 12, sigma=0.3)
    g, o = build_ivfpq(ctx, L2_SQUARED, X, X[:600], nlist, M, nbits)
    Q = clustered(62, 360, d, 12, sigma=0.3) + np.float32(0.01)
    check_search(g, o, Q, 10, 32)


def test_merge_topk(ctx):
    """comet_merge_topk_dev: merging per-shard Flat results equals searching the unsharded index."""
    import ctypes as C
    from comet_amd import FlatIndex
    from comet_amd._lib import check
    n, d, B, k, R = 4000, 32, 6, 20, 4
    X = synth(71, n, d); Q = synth(72, B, d)
    X[n // 2 + 3] = X[5]; X[n - 1] = X[5]                      # cross-shard score ties
    ids = np.arange(1, n + 1, dtype=np.uint32)
    full = orc.Flat(d, "l2_squared"); full.add_batch(ids, X)
    all_ids = np.zeros((R, B, k), np.uint32); all_sc = np.zeros((R, B, k), np.float32); all_cn = np.zeros((R, B), np.int32)
    for r in range(R):
        lo, hi = n * r // R, n * (r + 1) // R
        sh = FlatIndex(ctx, d, L2_SQUARED); sh.add_batch(ids[lo:hi], X[lo:hi])
        all_ids[r], all_sc[r], all_cn[r] = sh.search_batch(np.vstack([Q[:-1], X[5:6]]), k)
    Qm = np.vstack([Q[:-1], X[5:6]])
    bufs = [ctx.alloc(a.nbytes) for a in (all_ids, all_sc, all_cn)]
    for p, a in zip(bufs, (all_ids, all_sc, all_cn)):
        ctx.upload(p, a)
    o_ids, o_sc, o_cn = ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    check(ctx.lib.comet_merge_topk_dev(ctx.h, C.c_void_p(bufs[0]), C.c_void_p(bufs[1]), C.c_void_p(bufs[2]), R, B, k, k,
                                       C.c_void_p(o_ids), C.c_void_p(o_sc), C.c_void_p(o_cn)))
    ctx.sync()
    mi, ms, mc = ctx.download(o_ids, (B, k), np.uint32), ctx.download(o_sc, (B, k), np.float32), ctx.download(o_cn, (B,), np.int32)
    for b in range(B):
        cnt, oi, os_ = full.search(Qm[b], k)
        assert mc[b] == cnt and np.array_equal(mi[b, :cnt], oi) and np.array_equal(bits(ms[b, :cnt]), bits(os_))
    for p in bufs + [o_ids, o_sc, o_cn]:
        ctx.free(p)


@pytest.mark.parametrize("kind,policy", [("ivf", "members"), ("ivfpq", "members"), ("ivf", "lists"), ("ivfpq", "lists")])
def test_sharded_lists_merge_equals_unsharded(ctx, kind, policy):
    """SURVEY §8(e) for the inverted-list indexes: every rank trains on the same vectors (the GPU k-means is deterministic,
    so centroids / codebooks are replicated bit for bit), holds a round-robin share of the members ("members") or the whole
    lists dealt to it ("lists", comet_index_set_shard: balanced by list length), probes the same lists,
    and the per-shard top-K merged by comet_merge_topk_dev equal the unsharded search (scores bit for bit; ids wherever the
    score is unique — inside runs of equal scores the merged order is (shard, position) instead of scan position)."""
    import ctypes as C
    from comet_amd._lib import check
    n, d, B, k, R, nlist = 6000, 32, 8, 15, 3, 16
    X = clustered(81, n, d, 24); Q = clustered(82, B, d, 24)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    train = X[:2000]

    def make():
        g = IVFIndex(ctx, d, nlist, L2_SQUARED) if kind == "ivf" else IVFPQIndex(ctx, d, L2_SQUARED, nlist, 8, 6)
        g.train(train)
        return g
    full = make(); full.add_batch(ids, X)
    f_ids, f_sc, f_cn = full.search_batch(Q, k, nprobes=5)
    all_ids = np.zeros((R, B, k), np.uint32); all_sc = np.zeros((R, B, k), np.float32); all_cn = np.zeros((R, B), np.int32)
    owned = np.zeros((R, nlist), np.int64)
    for r in range(R):
        sh = make()
        assert np.array_equal(sh.centroids(nlist), full.centroids(nlist))          # replicated quantiser
        if policy == "members":
            sh.add_batch(ids[r::R], X[r::R])
        else:                                          # comet_index_set_shard: handed every vector, keeps the lists dealt to rank r (by length: LPT on the training set's list sizes)
            sh.set_shard(r, R)
            sh.add_batch(ids, X)
            owned[r] = [sh.list_size(l) for l in range(nlist)]
            assert sh.stat("shard_est_rows_max") <= 1.25 * sh.stat("shard_est_rows_mean")          # balanced by estimated rows, whatever the lists' indices
        all_ids[r], all_sc[r], all_cn[r] = sh.search_batch(Q, k, nprobes=5)
    if policy == "lists":                              # every list whole on exactly one rank
        for l in range(nlist):
            assert sorted(owned[:, l].tolist()) == [0] * (R - 1) + [full.list_size(l)], (l, owned[:, l])
    bufs = [ctx.alloc(a.nbytes) for a in (all_ids, all_sc, all_cn)]
    for p, a in zip(bufs, (all_ids, all_sc, all_cn)):
        ctx.upload(p, a)
    o_ids, o_sc, o_cn = ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    check(ctx.lib.comet_merge_topk_dev(ctx.h, C.c_void_p(bufs[0]), C.c_void_p(bufs[1]), C.c_void_p(bufs[2]), R, B, k, k,
                                       C.c_void_p(o_ids), C.c_void_p(o_sc), C.c_void_p(o_cn)))
    ctx.sync()
    mi, ms, mc = ctx.download(o_ids, (B, k), np.uint32), ctx.download(o_sc, (B, k), np.float32), ctx.download(o_cn, (B,), np.int32)
    for b in range(B):
        c = int(f_cn[b])
        assert mc[b] == c
        assert np.array_equal(bits(ms[b, :c]), bits(f_sc[b, :c]))
        uniq = np.array([np.sum(f_sc[b, :c] == s) == 1 for s in f_sc[b, :c]])
        assert np.array_equal(mi[b, :c][uniq], f_ids[b, :c][uniq])
        assert sorted(mi[b, :c].tolist()) == sorted(f_ids[b, :c].tolist()) or not uniq.all()
    for p in bufs + [o_ids, o_sc, o_cn]:
        ctx.free(p)


def test_merge_topk_packed_blocks(ctx):
    """comet_merge_topk_packed_dev: the same merge over per-shard blocks [ids | scores | counts] as one all-gather stacks them."""
    import ctypes as C
    from comet_amd._lib import check
    R, B, k = 3, 5, 8
    rng = np.random.default_rng(9)
    sc = np.sort(rng.random((R, B, k)).astype(np.float32), axis=2)
    ids = rng.integers(1, 1 << 30, (R, B, k)).astype(np.uint32)
    cn = rng.integers(0, k + 1, (R, B)).astype(np.int32)
    block = 2 * B * k + B + 7                                  # blocks may be padded
    packed = np.zeros((R, block), np.uint32)
    for r in range(R):
        packed[r, :B * k] = ids[r].ravel(); packed[r, B * k:2 * B * k] = sc[r].ravel().view(np.uint32); packed[r, 2 * B * k:2 * B * k + B] = cn[r].view(np.uint32)
    pk = ctx.alloc(packed.nbytes); ctx.upload(pk, packed)
    o_ids, o_sc, o_cn = ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    check(ctx.lib.comet_merge_topk_packed_dev(ctx.h, C.c_void_p(pk), block, R, B, k, k, C.c_void_p(o_ids), C.c_void_p(o_sc), C.c_void_p(o_cn)))
    ctx.sync()
    from comet_amd.dist import merge_topk_host
    ei, es, ec = merge_topk_host(ids, sc, cn, k)
    mi, ms, mc = ctx.download(o_ids, (B, k), np.uint32), ctx.download(o_sc, (B, k), np.float32), ctx.download(o_cn, (B,), np.int32)
    assert np.array_equal(mc, ec)
    for b in range(B):
        assert np.array_equal(mi[b, :ec[b]], ei[b, :ec[b]]) and np.array_equal(bits(ms[b, :ec[b]]), bits(es[b, :ec[b]]))
    for p in (pk, o_ids, o_sc, o_cn):
        ctx.free(p)


@pytest.mark.parametrize("metric", METRICS)
def test_ivfpq_two_stage_lower_bound_pruning(ctx, metric):
    """The two-stage IVFPQ search (nearest lists first, then an exact lower bound per (query, list) pair — the serial float32 sum of the
    pair's table row minima — removes pairs whose candidates cannot pass the query's bound) returns the oracle's ids and scores, and
    the same as the every-candidate search (mode 1), on a clustered corpus where most pairs are removed and on an unclustered one
    where hardly any is; with thresholds, document filters, soft deletes (removed candidates never count towards a bound), a K
    above the nearest list's size (the bound stays +inf), queries whose nearest list is not where their neighbours are."""
    d, M, nbits, nlist = 64, 8, 8, 24
    for tag, X in (("clustered", clustered(71, 16000, d, 24, 0.05)), ("uniform", synth(72, 6000, d))):
        n = len(X)
        ids = np.arange(1, n + 1, dtype=np.uint32)
        g = IVFPQIndex(ctx, d, metric, nlist, M, nbits); o = orc.IVFPQ(d, metric, nlist, M, nbits)
        g.train(X[:4000]); assert o.train(X[:4000]) == 0
        g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
        Q = np.vstack([X[5:25] + np.float32(0.003), synth(73, 7, d) * np.float32(0.6), clustered(71, 9, d, 24, 0.4)])   # near corpus points, far from everything, between clusters
        g.stat("adc_stats_on")                                # the counters are opt-in (device atomics)
        a0, b0 = g.stat("adc_pairs_alive"), g.stat("adc_pairs_behind_nearest")
        for k, npb in ((1, 2), (5, 8), (10, 24), (64, 6)):
            check_search(g, o, Q, k, npb)
            m0 = g.search_batch(Q, k, nprobes=npb)
            m1 = g.search_batch(Q, k, nprobes=npb, mode=1)
            assert np.array_equal(m0[2], m1[2]) and np.array_equal(m0[0], m1[0]) and np.array_equal(bits(m0[1]), bits(m1[1])), (tag, k, npb)
        alive, behind = g.stat("adc_pairs_alive") - a0, g.stat("adc_pairs_behind_nearest") - b0
        assert behind > 0 and alive <= behind
        if tag == "clustered":
            assert alive < 0.5 * behind, (alive, behind)          # the bound does remove most of the far lists here
        ref = o.search(Q[0], 50, 8)[2]
        check_search(g, o, Q, 10, 8, threshold=float(ref[5]))
        check_search(g, o, Q, 10, 8, filter_ids=list(range(3, n, 5)))
        for i in range(6, 26, 2):
            g.remove(i); assert o.remove(i) == 0
        check_search(g, o, Q, 10, 8)
        check_search(g, o, Q, 64, 24)


@pytest.mark.parametrize("d,M", [(32, 8), (128, 8), (96, 8), (48, 24)])      # dsub 4, 16: the one-kernel lower bound; 12, 2: the two-kernel form
def test_ivfpq_two_stage_subspace_widths(ctx, d, M):
    """The lower-bound kernels of the two-stage search are instantiated per subspace width (pq_bound_prep_kernel + pq_bound3_kernel for dsub 4 / 8 / 16
    with 8-bit codebooks, pq_rowmin_kernel + pq_lb_kernel otherwise): every width returns the oracle's rows, and the pruned search the every-candidate one's."""
    nlist, nbits = 16, 8
    X = clustered(81 + d, 9000, d, 16, 0.05)
    ids = np.arange(1, len(X) + 1, dtype=np.uint32)
    g = IVFPQIndex(ctx, d, L2_SQUARED, nlist, M, nbits); o = orc.IVFPQ(d, L2_SQUARED, nlist, M, nbits)
    g.train(X[:3000]); assert o.train(X[:3000]) == 0
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    Q = np.vstack([X[7:19] + np.float32(0.002), clustered(82 + d, 5, d, 16, 0.3)])
    g.stat("adc_stats_on")
    a0, b0 = g.stat("adc_pairs_alive"), g.stat("adc_pairs_behind_nearest")
    for k, npb in ((3, 4), (10, 16)):
        check_search(g, o, Q, k, npb)
        m0 = g.search_batch(Q, k, nprobes=npb); m1 = g.search_batch(Q, k, nprobes=npb, mode=1)
        assert np.array_equal(m0[2], m1[2]) and np.array_equal(m0[0], m1[0]) and np.array_equal(bits(m0[1]), bits(m1[1]))
    assert g.stat("adc_pairs_behind_nearest") > b0 and g.stat("adc_pairs_alive") - a0 < g.stat("adc_pairs_behind_nearest") - b0


@pytest.mark.parametrize("d,M", [(288, 72), (1040, 65)])      # dsub 4 / 16; more than 64 subspaces: a second round of pq_bound3_kernel with 8 / 1 live lanes
@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1.0e-3, 0.0), (1.0e3, 0.0), (1.0, 40.0)])
def test_ivfpq_lower_bound_rounds_and_margins(ctx, d, M, scale, offset):
    """pq_bound3_kernel's bound is NOT the table's arithmetic (fused multiply-adds on |c|^2 - 2 r.c + |r|^2, a tree sum, explicit margins): it must never
    remove a pair that holds one of the K best. Tiny and large magnitudes, rows far from the origin (large codeword and residual norms against small
    distances), subspace counts that need a second round: the pruned search returns the oracle's rows and the every-candidate search's."""
    nlist, nbits = 12, 8
    X = (clustered(91 + d, 5000, d, 12, 0.05) * np.float32(scale) + np.float32(offset)).astype(np.float32)
    ids = np.arange(1, len(X) + 1, dtype=np.uint32)
    g = IVFPQIndex(ctx, d, L2_SQUARED, nlist, M, nbits); o = orc.IVFPQ(d, L2_SQUARED, nlist, M, nbits)
    g.train(X[:2000]); assert o.train(X[:2000]) == 0
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    Q = np.vstack([X[3:11] + np.float32(0.002 * scale), (clustered(92 + d, 4, d, 12, 0.3) * np.float32(scale) + np.float32(offset)).astype(np.float32)])
    g.stat("adc_stats_on")
    b0 = g.stat("adc_pairs_behind_nearest")
    for k, npb in ((1, 12), (10, 6), (64, 12)):
        check_search(g, o, Q, k, npb)
        m0 = g.search_batch(Q, k, nprobes=npb); m1 = g.search_batch(Q, k, nprobes=npb, mode=1)
        assert np.array_equal(m0[2], m1[2]) and np.array_equal(m0[0], m1[0]) and np.array_equal(bits(m0[1]), bits(m1[1]))
    assert g.stat("adc_pairs_behind_nearest") > b0


def test_ivfpq_adaptive_staging_and_bound_refinement_keep_results(ctx):
    """Round 5: (i) on rows without cluster structure the lower bound of the two-stage search removes nothing, and after the first sampled search (every 32nd
    runs two-stage with counters on) the searches in between run single-stage; on clustered rows they stay two-stage. (ii) The fused filter's bound is
    refined from the survivors' row. Whatever the policy decides, every search returns what the every-candidate search (mode 1) and the oracle return."""
    import oracle_lib as orc
    from comet_amd import L2_SQUARED, IVFPQIndex
    n, d, nlist, M, B, k = 40_000, 64, 64, 8, 48, 10
    for kind in ("uniform", "clustered"):
        if kind == "uniform":
            X = orc.synth(41, 0, n * d).reshape(n, d)
            Qs = [orc.synth(42 + i, 0, B * d).reshape(B, d) for i in range(3)]
        else:
            cen = orc.synth(51, 0, 200 * d).reshape(200, d)
            X = (cen[np.arange(n) % 200] + orc.synth(52, 0, n * d).reshape(n, d) * np.float32(0.05)).astype(np.float32)
            Qs = [(cen[(np.arange(B) * 3 + i) % 200] + orc.synth(53 + i, 0, B * d).reshape(B, d) * np.float32(0.05)).astype(np.float32) for i in range(3)]
        g = IVFPQIndex(ctx, d, L2_SQUARED, nlist, M, 8)
        g.train(X[:8000]); g.add_batch(np.arange(1, n + 1, dtype=np.uint32), X)
        o = orc.IVFPQ(d, "l2_squared", nlist, M, 8)
        blob = g.to_bytes(); assert o.from_bytes(blob) == len(blob)
        want = [g.search_batch(Q, k, nprobes=16, mode=1) for Q in Qs]
        for b in range(0, B, 7):
            cnt, oi, os_ = o.search(Qs[0][b], k, 16)
            assert want[0][2][b] == cnt and np.array_equal(want[0][0][b, :cnt], oi) and np.array_equal(want[0][1][b, :cnt].view(np.uint32), np.asarray(os_, np.float32).view(np.uint32))
        for it in range(70):                                     # samples at searches 0, 32, 64 of the default mode
            got = g.search_batch(Qs[it % 3], k, nprobes=16)
            w = want[it % 3]
            assert np.array_equal(got[2], w[2]) and np.array_equal(got[0], w[0]) and np.array_equal(got[1].view(np.uint32), w[1].view(np.uint32)), (kind, it)
        ctx.sync()
        assert g.stat("adc_auto_searches") >= 70
        assert g.stat("adc_auto_one_stage") == (1.0 if kind == "uniform" else 0.0), kind
