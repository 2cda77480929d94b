"""The N > 1 path on CPU: world_size-2 gloo processes exercise the shard bounds, the all-gather exchange and the
merge ordering of comet_amd.dist. The per-shard search is done by the CPU oracle here (this is a test of the
exchange / merge logic; on GPUs the shards are searched by the HIP path and merged by comet_merge_topk_dev, which
the GPU test test_merge_topk covers against the same oracle)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:   # surface the failure instead of letting the parent time out
        q.put((rank, f"{type(e).__name__}: {e}"))


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib as orc
    from comet_amd.dist import TopKExchange, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, B, k = 3001, 24, 5, 12
    X = orc.synth(1, 0, n * d).reshape(n, d)
    X[n - 2] = X[3]; X[n // 2 + 1] = X[3]                       # equal scores across shards
    Q = np.vstack([orc.synth(2, 0, (B - 1) * d).reshape(B - 1, d), X[3:4]])
    ids = np.arange(1, n + 1, dtype=np.uint32)
    lo, hi = shard_bounds(n, rank, world)
    shard = orc.Flat(d, "l2_squared"); shard.add_batch(ids[lo:hi], X[lo:hi])
    ex = TopKExchange(B, k, torch.device("cpu"))
    for b in range(B):
        cnt, oi, os_ = shard.search(Q[b], k)
        ex.counts[b] = cnt
        ex.ids[b, :cnt] = torch.from_numpy(oi.view(np.int32).copy())
        ex.scores[b, :cnt] = torch.from_numpy(os_.copy())
    mi, ms, mc = ex.exchange_and_merge(k)
    full = orc.Flat(d, "l2_squared"); full.add_batch(ids, X)
    ok = True
    for b in range(B):
        cnt, oi, os_ = full.search(Q[b], k)
        ok &= int(mc[b]) == cnt and np.array_equal(mi[b, :cnt].numpy().view(np.uint32), oi) and \
            np.array_equal(ms[b, :cnt].numpy().view(np.uint32), os_.view(np.uint32))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_exchange_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    assert sorted(res) == [(0, True), (1, True)]


def test_merge_host_orders_ties_by_shard_then_position():
    from comet_amd.dist import merge_topk_host, shard_bounds
    ids = np.array([[[10, 11, 12]], [[20, 21, 22]]], np.uint32)
    sc = np.array([[[0.5, 1.0, 2.0]], [[0.5, 1.0, 1.5]]], np.float32)
    cnt = np.array([[3], [2]], np.int32)
    i, s, c = merge_topk_host(ids, sc, cnt, 4)
    assert c[0] == 3 or c[0] == 4
    assert i[0, :3].tolist() == [10, 20, 11] and s[0, :3].tolist() == [0.5, 0.5, 1.0]
    i, s, c = merge_topk_host(ids, sc, cnt, 0)
    assert c[0] == 3                       # k_cap bounds the row
    assert [shard_bounds(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]


def _rdzv_worker(rank, world, port, q):
    import ctypes as C
    from comet_amd.dist import _rendezvous_id

    class FakeLib:                       # stands in for libcomet_hip.so: rank 0 "creates" the 128-byte RCCL id
        @staticmethod
        def comet_comm_unique_id(buf):
            for i in range(128):
                buf[i] = (i * 7 + 3) & 0xFF
            return 0
    q.put((rank, _rendezvous_id(FakeLib, rank, world, "127.0.0.1", port, timeout_s=30.0)))


def test_comm_id_rendezvous_three_ranks():
    """The only host-side step of the in-library RCCL path: rank 0 hands the 128-byte unique id to the other ranks over TCP on
    127.0.0.1 (comet_amd.dist._rendezvous_id, what Comm.from_env does under torch.distributed.run). Late and early joiners."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    port, q, world = _free_port(), ctx.Queue(), 3
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in (1, 0, 2)]   # a client starts before the server
    procs[0].start(); time.sleep(0.5); procs[1].start(); procs[2].start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(30); assert p.exitcode == 0
    want = bytes((i * 7 + 3) & 0xFF for i in range(128))
    assert got == {0: want, 1: want, 2: want}
