"""One rank of a multi-process sharded search on ONE GPU (test infrastructure; see tests/shm_rccl.cpp and tests/test_comm_multirank_gpu.py).
usage: python tests/comm_worker.py <rank> <world> <port> <case> <outfile.npz>"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as orc                     # deterministic synthetic data only (the checker's generator)
import comet_amd as ca
from comet_amd.dist import Comm

CASES = {
    # name: (kind, n, d, B, K, batches, extra)
    "flat": ("flat", 40_000, 96, 48, 20, 6, {}),
    "flat_bigk": ("flat", 30_000, 64, 3, 2600, 3, {}),            # world x k_cap > 8192 candidates per query: the merge sorts in global memory (its own workspace)
    "ivf": ("ivf", 30_000, 64, 40, 10, 5, {"nprobes": 6}),
    "ivfpq": ("ivfpq", 30_000, 96, 40, 10, 5, {"nprobes": 6}),
    "ivfpq_members": ("ivfpq_members", 30_000, 96, 40, 10, 5, {"nprobes": 6}),
}


def data(case):
    kind, n, d, B, K, nb, kw = CASES[case]
    centers = orc.synth(901, 0, 64 * d).reshape(64, d)
    X = (centers[np.arange(n) % 64] + orc.synth(902, 0, n * d).reshape(n, d) * np.float32(0.2)).astype(np.float32)
    Qs = [(centers[(np.arange(B) * 7 + i) % 64] + orc.synth(903 + i, 0, B * d).reshape(B, d) * np.float32(0.2)).astype(np.float32) for i in range(nb)]
    return X, Qs


def build(ctx, case, X, rank, world):
    kind, n, d, B, K, nb, kw = CASES[case]
    ids = np.arange(1, n + 1, dtype=np.uint32)
    if kind == "flat":
        idx = ca.FlatIndex(ctx, d, ca.COSINE)
        lo, hi = n * rank // world, n * (rank + 1) // world
        idx.add_batch(ids[lo:hi], X[lo:hi])
    elif kind == "ivf":
        idx = ca.IVFIndex(ctx, d, 24, ca.L2_SQUARED)
        idx.train(X[:3000])
        if world > 1:
            idx.set_shard(rank, world)
        idx.add_batch(ids, X)
    else:
        idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, 24, 12, 6)
        idx.train(X[:4000])
        if world > 1 and kind == "ivfpq":
            idx.set_shard(rank, world)
            idx.add_batch(ids, X)
        elif world > 1:                       # member sharding: every rank keeps a round-robin share of every list (the caller adds its share)
            idx.add_batch(ids[rank::world], X[rank::world])
        else:
            idx.add_batch(ids, X)
    return idx


def main():
    rank, world, port, case, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    kind, n, d, B, K, nb, kw = CASES[case]
    ctx = ca.Context(0)
    comm = Comm(ctx, rank, world, port=port)
    assert comm.allreduce_max(float(rank)) == float(world - 1)
    comm.barrier()
    X, Qs = data(case)
    idx = build(ctx, case, X, rank, world)
    q_dev = [ctx.alloc(B * d * 4) for _ in Qs]
    for p, q in zip(q_dev, Qs):
        ctx.upload(p, q)
    outs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in Qs]
    tickets = []
    for i in range(nb):                       # up to 3 in flight, finished out of step with the enqueues
        tickets.append(comm.search_async(idx, q_dev[i], B, K, *outs[i], K, **kw))
        if i >= 2:
            comm.search_wait(idx, tickets[i - 2], block=False)
    for t in tickets[-2:]:
        comm.search_wait(idx, t, block=True)
    comm.sync()
    res = {}
    for i in range(nb):
        res[f"ids{i}"] = ctx.download(outs[i][0], (B, K), np.uint32); res[f"sc{i}"] = ctx.download(outs[i][1], (B, K), np.float32); res[f"cn{i}"] = ctx.download(outs[i][2], (B,), np.int32)
    comm.barrier()
    np.savez(out, **res)
    comm.close()


if __name__ == "__main__":
    main()
