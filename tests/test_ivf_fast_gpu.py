"""The MFMA fast path of the IVF list scan (kernels_ivf.hip; reference: ivf_index_search.go:217-322): fp16 screening of the probed
lists on the matrix cores + exact float32 rescoring must return what the exact kernels (search mode 1) and the CPU oracle return,
bit for bit: ids, scores, counts, order — for every metric, ragged and empty lists, lists shared by more than one query group,
batches beyond one slice, filters, soft deletes, thresholds, ties and k beyond the candidates."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, CometError, IVFIndex

pytestmark = pytest.mark.gpu
METRICS = [EUCLIDEAN, L2_SQUARED, COSINE]
OM = {EUCLIDEAN: "l2", L2_SQUARED: "l2_squared", COSINE: "cosine"}


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def clustered(seed, n, d, ncl, sigma=0.15):
    centers = synth(seed, ncl, d)
    noise = synth(seed + 1, n, d) * np.float32(sigma)
    return (centers[np.arange(n) % ncl] + noise).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b, what=""):
    ai, as_, ac = a
    bi, bs, bc = b
    assert np.array_equal(ac, bc), (what, ac, bc)
    for q in range(len(ac)):
        n = min(int(ac[q]), ai.shape[1])
        assert np.array_equal(ai[q, :n], bi[q, :n]), (what, q, ai[q, :n], bi[q, :n])
        assert np.array_equal(bits(as_[q, :n]), bits(bs[q, :n])), (what, q)


def build(ctx, metric, X, T, nlist, with_oracle=True):
    g = IVFIndex(ctx, X.shape[1], nlist, metric)
    g.train(T)
    ids = np.arange(1, len(X) + 1, dtype=np.uint32)
    g.add_batch(ids, X)
    o = None
    if with_oracle:
        o = orc.IVF(X.shape[1], OM[metric], nlist)
        assert o.train(T) == 0
        assert o.add_batch(ids, X) == 0
    return g, o


def vs_oracle(g, o, Q, k, nprobes, **kw):
    ids, sc, cnt = g.search_batch(Q, k, nprobes=nprobes, mode=2, threshold=kw.get("threshold", 0.0), document_ids=kw.get("filter_ids", ()))
    for b, q in enumerate(Q):
        n, oi, os_ = o.search(q, k, nprobes, threshold=kw.get("threshold", 0.0), filter_ids=kw.get("filter_ids", ()))
        assert cnt[b] == n, (b, cnt[b], n)
        assert np.array_equal(ids[b, :n], oi), (b, ids[b, :n], oi)
        assert np.array_equal(bits(sc[b, :n]), bits(os_)), (b, sc[b, :n], os_)


@pytest.mark.parametrize("metric", METRICS)
def test_ivf_fast_matches_oracle(ctx, metric):
    n, d, nlist = 6000, 72, 24
    X = clustered(131, n, d, 20)
    Q = clustered(131, 40, d, 20) + np.float32(0.01)
    g, o = build(ctx, metric, X, X[:1500], nlist)
    for nprobes in (1, 4, 24, 0):
        vs_oracle(g, o, Q, 10, nprobes)
    assert g.stat("fast_queries") == len(Q) and g.stat("strict_queries") == 0
    vs_oracle(g, o, Q, 100, 6)
    ref = o.search(Q[0], 40, 6)[2]
    vs_oracle(g, o, Q, 40, 6, threshold=float(ref[15]))
    vs_oracle(g, o, Q, 15, 6, filter_ids=list(range(1, 3000, 2)))
    for i in (5, 6, 1000, 2999, 4000):
        g.remove(i); assert o.remove(i) == 0
    vs_oracle(g, o, Q, 15, 6)
    vs_oracle(g, o, Q, 15, 6, filter_ids=list(range(1, 3000, 3)))
    g.flush(); o.flush()
    vs_oracle(g, o, Q, 15, 6)


@pytest.mark.parametrize("metric", METRICS)
def test_ivf_fast_equals_strict_ragged_lists_and_groups(ctx, metric):
    """lists of length 0, 1, 63, 64, 65, 257 and one long list probed by 150 queries (three query groups, several tiles)"""
    d, nlist = 64, 16
    centers = synth(7, nlist, d) * np.float32(4.0)
    sizes = [0, 1, 63, 64, 65, 257, 2100, 300, 5, 700, 128, 256, 512, 31, 33, 1000]
    T = np.vstack([centers[l] + synth(100 + l, 40, d) * np.float32(0.05) for l in range(nlist)]).astype(np.float32)
    X = np.vstack([centers[l] + synth(200 + l, max(s, 1), d)[:s] * np.float32(0.05) for l, s in enumerate(sizes)]).astype(np.float32)
    g, _ = build(ctx, metric, X, T, nlist, with_oracle=False)
    Q = np.vstack([centers[6] + synth(300, 150, d) * np.float32(0.05), centers + synth(301, nlist, d) * np.float32(0.05)]).astype(np.float32)
    for nprobes in (1, 3, 16):
        for k in (1, 10, 200):
            same(g.search_batch(Q, k, nprobes=nprobes, mode=2), g.search_batch(Q, k, nprobes=nprobes, mode=1), (metric, nprobes, k))
    # auto mode takes the fast path here
    same(g.search_batch(Q, 10, nprobes=3), g.search_batch(Q, 10, nprobes=3, mode=1))
    assert g.stat("strict_queries") == len(Q)      # counters describe the last search (mode 1)


def test_ivf_fast_batches_beyond_one_slice_and_ties(ctx):
    d, nlist, n = 40, 12, 5000
    X = clustered(9, n, d, 12)
    X[100:140] = X[100]                    # 40 identical vectors: tie order = scan position
    X[3000:3010] = X[100]
    g, o = build(ctx, L2_SQUARED, X, X[:1200], nlist)
    Q = np.vstack([clustered(9, 600, d, 12) + np.float32(0.02), X[100:101]]).astype(np.float32)     # 601 queries: three slices
    same(g.search_batch(Q, 20, nprobes=4, mode=2), g.search_batch(Q, 20, nprobes=4, mode=1))
    vs_oracle(g, o, Q[-3:], 60, 4)
    # k beyond the candidates of the probed lists, and more candidates than the post stage holds (-> flagged, re-run exactly)
    same(g.search_batch(Q[:50], 1000, nprobes=1, mode=2), g.search_batch(Q[:50], 1000, nprobes=1, mode=1))
    with pytest.raises(CometError):
        g.search_batch(Q[:4], 2000, nprobes=1, mode=2)      # k > 1024: no fast path
    with pytest.raises(CometError):
        g.search_batch(Q[:4], 0, nprobes=1, mode=2)         # k <= 0 (all candidates): no fast path


def test_ivf_fast_adversarial_overflow(ctx):
    """thousands of rows at the same distance from the query: the candidate list overflows and the query is re-run on the exact kernels"""
    d, nlist = 32, 4
    base = synth(5, nlist, d) * np.float32(3.0)
    X = np.vstack([np.repeat(base[0:1], 6000, axis=0), base[1] + synth(6, 500, d) * np.float32(0.1), base[2] + synth(7, 500, d) * np.float32(0.1),
                   base[3] + synth(8, 500, d) * np.float32(0.1)]).astype(np.float32)
    T = np.vstack([base[l] + synth(20 + l, 30, d) * np.float32(0.01) for l in range(nlist)]).astype(np.float32)
    g, _ = build(ctx, L2_SQUARED, X, T, nlist, with_oracle=False)
    Q = np.vstack([base[0:1] + np.float32(0.001), base[1:2]]).astype(np.float32)
    a = g.search_batch(Q, 10, nprobes=2, mode=2)
    assert g.stat("fast_overflows") >= 1
    same(a, g.search_batch(Q, 10, nprobes=2, mode=1))


@pytest.mark.parametrize("metric", [L2_SQUARED, COSINE])
def test_ivf_fast_config5_shape(ctx, metric):
    """BASELINE configs[4]'s vector leg at its own parameters (d 768, nlist 1024, nprobe 32 and the hybrid default 1, K 10, B 256) on a
    corpus the exact kernels can check"""
    n, d, nlist = 40000, 768, 1024
    X = clustered(77, n, d, 512, 0.2)
    g, _ = build(ctx, metric, X, X[:10240], nlist, with_oracle=False)
    Q = clustered(77, 256, d, 512, 0.2) + np.float32(0.01)
    for nprobes in (1, 32):
        same(g.search_batch(Q, 10, nprobes=nprobes, mode=2), g.search_batch(Q, 10, nprobes=nprobes, mode=1), (metric, nprobes))
