"""The in-library RCCL exchange (comet_comm_* / comet_index_search_sharded_*) on one GPU: a world-size-1 communicator runs
the real ncclAllGather + merge on the exchange stream, pipelined with the next batch's search, and must reproduce the plain
search bit for bit. (Multi-rank merging of per-shard blocks is covered against the unsharded oracle by
test_quant_gpu.py::test_sharded_lists_merge_equals_unsharded / test_merge_topk_packed_blocks; the rank plumbing by the
world-2 gloo CPU test.)"""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, FlatIndex, IVFPQIndex, L2_SQUARED
from comet_amd.dist import Comm

pytestmark = pytest.mark.gpu


def test_world1_sharded_search_equals_plain_search(ctx):
    comm = Comm(ctx, 0, 1, port=29733)
    assert comm.allreduce_max(3.5) == 3.5
    comm.barrier()
    n, d, B, K = 30_000, 96, 32, 20
    X = orc.synth(81, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    flat = FlatIndex(ctx, d, COSINE); flat.add_batch(ids, X)
    pq = IVFPQIndex(ctx, d, L2_SQUARED, 16, 12, 6); pq.train(X[:4000]); pq.add_batch(ids, X)
    Qs = [orc.synth(82 + i, 0, B * d).reshape(B, d) for i in range(5)]
    q_dev = [ctx.alloc(B * d * 4) for _ in Qs]
    for p, q in zip(q_dev, Qs):
        ctx.upload(p, q)
    outs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in Qs]
    for index, kw in ((flat, {}), (pq, {"nprobes": 5})):
        want = [index.search_batch(q, K, **kw) for q in Qs]
        tickets = []
        for i in range(len(Qs)):                       # up to 3 in flight, finished out of step with the enqueues
            tickets.append(comm.search_async(index, q_dev[i], B, K, *outs[i], K, **kw))
            if i >= 2:
                comm.search_wait(index, tickets[i - 2], block=False)
        for t in tickets[-2:]:
            comm.search_wait(index, t, block=True)
        comm.sync()
        for i, (wi, ws, wc) in enumerate(want):
            gi = ctx.download(outs[i][0], (B, K), np.uint32); gs = ctx.download(outs[i][1], (B, K), np.float32); gc = ctx.download(outs[i][2], (B,), np.int32)
            assert np.array_equal(gc, wc) and np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32)), i
    comm.close()


def test_world1_real_rccl_round_trip_at_the_headline_block_size(ctx):
    """The packed block of the headline step (B = 256 queries x K = 100: 2 x 25 600 words + 256 counts) through the REAL ncclAllGather on the exchange stream, four
    batches in flight (the communicator's slots), plus the host all-reduces (max, sum) the bench's timing uses — what the driver's first multi-GPU run issues,
    at world size 1 (the only size this box can give real RCCL)."""
    comm = Comm(ctx, 0, 1, port=29735)
    assert comm.allreduce_max(-2.25) == -2.25
    for _ in range(3):
        comm.barrier()
    n, d, B, K = 60_000, 128, 256, 100
    X = orc.synth(91, 0, n * d).reshape(n, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    flat = FlatIndex(ctx, d, COSINE); flat.add_batch(ids, X)
    Qs = [orc.synth(92 + i, 0, B * d).reshape(B, d) for i in range(6)]
    q_dev = [ctx.alloc(B * d * 4) for _ in Qs]
    for p, q in zip(q_dev, Qs):
        ctx.upload(p, q)
    outs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in Qs]
    want = [flat.search_batch(q, K) for q in Qs]
    tickets = []
    for i in range(len(Qs)):
        tickets.append(comm.search_async(flat, q_dev[i], B, K, *outs[i], K))
        if i >= 3:
            comm.search_wait(flat, tickets[i - 3], block=False)
    for t in tickets[-3:]:
        comm.search_wait(flat, t, block=True)
    comm.sync()
    for i, (wi, ws, wc) in enumerate(want):
        gi = ctx.download(outs[i][0], (B, K), np.uint32); gs = ctx.download(outs[i][1], (B, K), np.float32); gc = ctx.download(outs[i][2], (B,), np.int32)
        assert np.array_equal(gc, wc) and np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32)), i
    comm.close()
