"""The context's second execution lane (common.hpp: Ctx::alt): every other asynchronous search of an index runs on its own stream with
its own scratch arena. Whatever interleaving of searches, waits, removes and adds the host produces, the results are those of the
synchronous call; and a search_wait makes exactly its search's results final."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, L2_SQUARED, FlatIndex, HNSWIndex, IVFIndex, IVFPQIndex, PQIndex

pytestmark = pytest.mark.gpu


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def same(a, b):
    (i1, s1, c1), (i2, s2, c2) = a, b
    assert np.array_equal(c1, c2)
    for q in range(len(c1)):
        n = c1[q]
        assert np.array_equal(i1[q, :n], i2[q, :n]), (q, i1[q, :n], i2[q, :n])
        assert np.array_equal(np.ascontiguousarray(s1[q, :n]).view(np.uint32), np.ascontiguousarray(s2[q, :n]).view(np.uint32)), q


def make(ctx, kind, d, X):
    ids = np.arange(1, len(X) + 1, dtype=np.uint32)
    if kind == "flat":
        g = FlatIndex(ctx, d, COSINE)
    elif kind == "hnsw":
        g = HNSWIndex(ctx, d, L2_SQUARED, 8, 40, 24)          # graph built by the GPU insert kernel
    elif kind == "ivf":
        g = IVFIndex(ctx, d, 32, L2_SQUARED); g.train(X[:4000])
    elif kind == "ivfpq":
        g = IVFPQIndex(ctx, d, L2_SQUARED, 32, 8, 6); g.train(X[:4000])
    else:
        g = PQIndex(ctx, d, L2_SQUARED, 8, 6); g.train(X[:4000])
    g.add_batch(ids, X)
    return g


@pytest.mark.parametrize("kind", ["flat", "ivf", "ivfpq", "pq", "hnsw"])
def test_alternating_lanes_equal_sync(ctx, kind):
    n, d, B, k, nb = (30000 if kind != "hnsw" else 4000), 64, 96, 7, 6
    centers = synth(61, 50, d)
    X = (centers[np.arange(n) % 50] + synth(62, n, d) * np.float32(0.3)).astype(np.float32)
    g = make(ctx, kind, d, X)
    kw = {} if kind in ("flat", "pq", "hnsw") else {"nprobes": 5}
    batches = [(centers[np.arange(B) % 50] + synth(70 + i, B, d) * np.float32(0.3)).astype(np.float32) for i in range(nb)]
    qd = [ctx.alloc(B * d * 4) for _ in batches]
    outs = [(ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)) for _ in batches]
    for p, Qb in zip(qd, batches):
        ctx.upload(p, Qb)

    def fetch(i):
        return (ctx.download(outs[i][0], (B, k), np.uint32), ctx.download(outs[i][1], (B, k), np.float32), ctx.download(outs[i][2], (B,), np.int32))

    for rnd in range(3):
        want = [g.search_batch(Qb, k, **kw) for Qb in batches]
        # all in flight at once (lanes 1, 0, 1, 0, ...), waited for in a different order
        tickets = [g.search_batch_dev_async(qd[i], B, k, *outs[i], k, **kw) for i in range(nb)]
        for i in (3, 0, 5, 1, 4, 2):
            g.search_wait(tickets[i])
            same(fetch(i), want[i])
        # the bench's pattern: enqueue i + 1, then wait for i
        prev = None
        for i in range(nb):
            t = g.search_batch_dev_async(qd[i], B, k, *outs[i], k, **kw)
            if prev is not None:
                g.search_wait(prev[1]); same(fetch(prev[0]), want[prev[0]])
            prev = (i, t)
        g.search_wait(prev[1]); same(fetch(prev[0]), want[prev[0]])
        # searches in flight on both lanes while the host changes the index: the calls wait for the lanes they must
        t0 = g.search_batch_dev_async(qd[0], B, k, *outs[0], k, **kw)
        t1 = g.search_batch_dev_async(qd[1], B, k, *outs[1], k, **kw)
        for rid in range(1 + rnd * 50, 40 + rnd * 50, 3):
            g.remove(rid)                                   # soft delete: the next search rebuilds the device-side list
        t2 = g.search_batch_dev_async(qd[2], B, k, *outs[2], k, **kw)
        g.add_batch(np.arange(n + 1 + rnd * 100, n + 101 + rnd * 100, dtype=np.uint32), X[rnd * 100:rnd * 100 + 100] * np.float32(1.01))
        t3 = g.search_batch_dev_async(qd[3], B, k, *outs[3], k, **kw)
        for t in (t0, t1, t2, t3):
            g.search_wait(t)
        same(fetch(3), g.search_batch(batches[3], k, **kw))
        same(fetch(0), want[0]); same(fetch(1), want[1])   # enqueued before the changes
    ctx.sync()
    for p in qd + [x for o in outs for x in o]:
        ctx.free(p)


def test_set_lanes_bounds_and_single_lane(ctx):
    from comet_amd import CometError
    for bad in (0, 9, -1):
        with pytest.raises(CometError):
            ctx.set_lanes(bad)
    n, d, B, k = 20000, 48, 64, 5
    X = synth(91, n, d); Q = synth(92, B, d)
    g = make(ctx, "ivfpq", d, X)
    want = g.search_batch(Q, k, nprobes=4)
    qd = ctx.alloc(B * d * 4); ctx.upload(qd, Q)
    outs = [(ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)) for _ in range(4)]
    try:
        for lanes in (1, 3, 4):
            ctx.set_lanes(lanes)
            ts = [g.search_batch_dev_async(qd, B, k, *o, k, nprobes=4) for o in outs]
            for t, o in zip(ts, outs):
                g.search_wait(t)
                same((ctx.download(o[0], (B, k), np.uint32), ctx.download(o[1], (B, k), np.float32), ctx.download(o[2], (B,), np.int32)), want)
    finally:
        ctx.set_lanes(4)
    for p in [qd] + [x for o in outs for x in o]:
        ctx.free(p)


def test_async_search_starts_behind_synth_fill_on_lane0(ctx):
    """The header's own example (include/comet_gpu.h, comet_ctx_fence): a generator call that only ENQUEUES on lane 0 (comet_synth_fill_dev holds no CallGuard)
    followed at once by asynchronous searches that land on lanes 1..3 — the searches must read the finished queries. The buffer is refilled many times with a
    large fill in front (the race needs the generator still running when the search is enqueued); every result is compared with a blocking search of the
    same queries generated on the host."""
    n, d, B, k = 20000, 64, 64, 5
    X = synth(191, n, d)
    g = make(ctx, "ivfpq", d, X)
    big = ctx.alloc(64 << 20)
    qd = ctx.alloc(B * d * 4)
    outs = [(ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)) for _ in range(4)]
    ctx.set_lanes(4)
    try:
        g.search_wait(g.search_batch_dev_async(qd, B, k, *outs[0], k, nprobes=4))        # the context has seen an asynchronous search: lanes exist
        for rnd in range(12):
            seed = 500 + rnd
            want = g.search_batch(synth(seed, B, d), k, nprobes=4)
            ctx.synth_fill(big, 7, 0, (64 << 20) // 4)          # ~10 us .. 100 us of generator work queued on lane 0
            ctx.synth_fill(qd, seed, 0, B * d)                  # then the queries, still only enqueued
            ts = [g.search_batch_dev_async(qd, B, k, *o, k, nprobes=4) for o in outs]    # lanes 1, 2, 3, 0
            for t, o in zip(ts, outs):
                g.search_wait(t)
                same((ctx.download(o[0], (B, k), np.uint32), ctx.download(o[1], (B, k), np.float32), ctx.download(o[2], (B,), np.int32)), want)
    finally:
        ctx.set_lanes(4)
    for p in [qd, big] + [x for o in outs for x in o]:
        ctx.free(p)
