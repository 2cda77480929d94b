"""Queries (and rows) with non-finite components: a caller's bug must come back as rows of results or as an error, never as a dead process. The reference
computes NaN distances and sorts them with an inconsistent comparator (its output for such a query is unspecified), so nothing is compared with the oracle
here except the FINITE queries of the same batch, which must be untouched by their neighbours. Found by round 5's lane test: a query buffer of recycled device
memory (ids of an earlier search: 0xFFFFFFFF words = NaNs whose payload equals the selection kernels' EXCLUDED sentinel) left the probe list of an IVFPQ search
unwritten and the next kernel read list_len[garbage]."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, L2_SQUARED, FlatIndex, HNSWIndex, IVFIndex, IVFPQIndex, PQIndex

pytestmark = pytest.mark.gpu


def patterns(B, d):
    rng = np.random.default_rng(5)
    base = orc.synth(7, 0, B * d).reshape(B, d)
    out = {"nan": np.full((B, d), np.nan, np.float32), "inf": np.full((B, d), np.inf, np.float32), "huge": np.full((B, d), 3e38, np.float32),
           "ones_payload": np.full((B, d), 0xFFFFFFFF, np.uint32).view(np.float32),
           "bits": rng.integers(0, 2**32, (B, d), dtype=np.uint64).astype(np.uint32).view(np.float32)}
    m = base.copy(); m[::4, 5] = np.nan; m[1::4, 7] = np.inf; m[2::4, 9] = np.float32(-3e38); m[2::4, 3] = np.array([0xFFFFFFFF], np.uint32).view(np.float32)[0]
    out["mixed"] = m                         # every 4th query (3 mod 4) is finite
    return base, out


@pytest.mark.parametrize("kind", ["flat", "flat_cos", "ivf", "ivf_big", "pq", "ivfpq", "ivfpq_big", "hnsw"])
def test_nonfinite_queries_do_not_kill_the_process(ctx, kind):
    n, d, B, k = 20000, 64, 64, 5
    X = orc.synth(191, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    kw = {}
    if kind == "flat":
        g = FlatIndex(ctx, d, L2_SQUARED)
    elif kind == "flat_cos":
        g = FlatIndex(ctx, d, COSINE)
    elif kind == "ivf":
        g = IVFIndex(ctx, d, 32, L2_SQUARED); g.train(X[:4000]); kw = {"nprobes": 4}             # nlist < 64: the exact coarse ranking
    elif kind == "ivf_big":
        g = IVFIndex(ctx, d, 256, COSINE); g.train(X[:8000]); kw = {"nprobes": 8}                 # the MFMA coarse ranking (coarse_pick_kernel)
    elif kind == "pq":
        g = PQIndex(ctx, d, L2_SQUARED, 8, 6); g.train(X[:4000])
    elif kind == "ivfpq":
        g = IVFPQIndex(ctx, d, L2_SQUARED, 32, 8, 6); g.train(X[:4000]); kw = {"nprobes": 4}
    elif kind == "ivfpq_big":
        g = IVFPQIndex(ctx, d, L2_SQUARED, 256, 8, 6); g.train(X[:8000]); kw = {"nprobes": 8}
    else:
        g = HNSWIndex(ctx, d, L2_SQUARED, 8, 40, 24); n = 3000
    g.add_batch(ids[:n], X[:n])
    base, pats = patterns(B, d)
    want = g.search_batch(base, k, **kw)
    for name, Q in pats.items():
        try:
            gi, gs, gc = g.search_batch(Q, k, **kw)
        except Exception as e:               # noqa: BLE001 — an error is an acceptable answer (cosine: ErrZeroVector-like), a crash is not
            assert "HIP error" not in str(e), (kind, name, e)
            continue
        assert ((gc <= k) & (gc >= -16)).all(), (kind, name, gc)
        for b in range(B):
            if gc[b] > 0:
                assert (gi[b, :gc[b]] >= 1).all() and (gi[b, :gc[b]] <= n).all(), (kind, name, b, gi[b])
        if name == "mixed":                  # the finite queries of the batch: exactly what they get on their own
            for b in range(3, B, 4):
                assert gc[b] == want[2][b] and np.array_equal(gi[b, :gc[b]], want[0][b, :gc[b]]), (kind, b)
                assert np.array_equal(gs[b, :gc[b]].view(np.uint32), want[1][b, :gc[b]].view(np.uint32)), (kind, b)
    # the context is still alive and exact
    again = g.search_batch(base, k, **kw)
    assert np.array_equal(again[0], want[0]) and np.array_equal(again[2], want[2])
