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
    # BASELINE configs[3] with every one of its parameters (d 768, nlist 4096, nprobe 32, M 96, nbits 8, K 10) in its sharded form: lists dealt by length (LPT),
    # stage 1 on the nearest list's owner, bounds all-reduced — checked against the ORACLE by the test (not only against the unsharded HIP index)
    "ivfpq_c3": ("ivfpq", 60_000, 768, 32, 10, 3, {"nprobes": 32}),
    # rank 1's second search throws after the batch was entered: nobody may hang, every rank sees counts = -code for that batch, the others are untouched
    "ivfpq_fail": ("ivfpq", 30_000, 96, 40, 10, 4, {"nprobes": 6}),
    # rank 1 was handed another placement than its peers: the first sharded search must fail on EVERY rank
    "ivfpq_owners": ("ivfpq", 30_000, 96, 40, 10, 1, {"nprobes": 6}),
}
SHAPES = {"ivfpq_c3": (4096, 96, 8, 40_960)}          # nlist, M, nbits, training vectors (the reference's minimum nlist x 10, ivfpq_index.go:185)


def data(case):
    kind, n, d, B, K, nb, kw = CASES[case]
    if case == "ivfpq_c3":                    # the inputs of tests/test_configs_gpu.py::test_config3_* (SplitMix64 rows)
        X = orc.synth(0xC0FFEE + 3, 0, n * d).reshape(n, d)
        return X, [orc.synth(0xBEEF + 3 + i, 0, B * d).reshape(B, d) for i in range(nb)]
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
        nlist, M, nbits, ntrain = SHAPES.get(case, (24, 12, 6, 4000))
        idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, nlist, M, nbits)
        idx.train(X[:ntrain])
        if world > 1 and kind == "ivfpq":
            idx.set_shard(rank, world)
            if case == "ivfpq_owners" and rank == 1:
                own = idx.list_owners(nlist); own[0] = (own[0] + 1) % world
                idx.set_list_owners(own)
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
    res = {}
    if case == "ivfpq_owners":
        try:
            comm.search_async(idx, q_dev[0], B, K, *outs[0], K, **kw)
            res["error"] = np.array("")
        except RuntimeError as e:
            res["error"] = np.array(str(e))
        np.savez(out, **res)
        comm.close()
        return
    errors = []

    def wait(t, block):
        try:
            comm.search_wait(idx, t, block=block)
        except RuntimeError as e:             # ivfpq_fail: the failing rank learns of its error here; its peers read it from the counts
            errors.append(str(e))
    for i in range(nb):                       # up to 3 in flight, finished out of step with the enqueues
        tickets.append(comm.search_async(idx, q_dev[i], B, K, *outs[i], K, **kw))
        if i >= 2:
            wait(tickets[i - 2], False)
    for t in tickets[-2:]:
        wait(t, True)
    comm.sync()
    res["errors"] = np.array("\n".join(errors))
    if world > 1 and kind == "ivfpq":
        res["owners"] = idx.list_owners(SHAPES.get(case, (24,))[0])
    for i in range(nb):
        res[f"ids{i}"] = ctx.download(outs[i][0], (B, K), np.uint32); res[f"sc{i}"] = ctx.download(outs[i][1], (B, K), np.float32); res[f"cn{i}"] = ctx.download(outs[i][2], (B,), np.int32)
    comm.barrier()
    np.savez(out, **res)
    comm.close()


if __name__ == "__main__":
    main()
