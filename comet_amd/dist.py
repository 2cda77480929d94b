"""Multi-GPU exchange for sharded indexes: one process per GPU, every rank searches its own shard for the same query
batch, the per-shard top-K blocks are exchanged with ONE RCCL all-gather per batch and every rank merges them.

On GPUs the whole data path lives behind the C ABI (comet_comm_* / comet_index_search_sharded_*: RCCL over xGMI on the
library's own exchange stream, merge_topk_kernel) — `Comm` / `ShardedSearch` below only carry the 128-byte RCCL id from
rank 0 to the other processes over a plain TCP socket (the control plane a Go host would do with whatever it has) and
forward calls. torch is not involved.

The merge is exact (top-K of a union = top-K of the per-part top-Ks) and keeps the canonical tie order of the
unsharded index as long as shards hold contiguous, ascending row blocks: ties go to the lower shard, then to the
lower position. The reference's analogue is the per-segment fan-out + mergeResults (storage.go:546-626,
storage_merge.go:13-46), which also runs on the host.

On GPUs the merge is the HIP kernel behind comet_merge_topk_dev; with CPU tensors (gloo — used by the CPU test of
the exchange logic) a small numpy merge with the same ordering is used.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_bounds(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous row block of `rank` (what bench.py and the tests use)."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def merge_topk_host(ids: np.ndarray, scores: np.ndarray, counts: np.ndarray, k: int):
    """ids/scores: [R, B, k_cap], counts: [R, B] -> merged ([B, k_cap] ids, scores, [B] counts).
    Order: score ascending, then shard, then position (same as merge_topk_kernel)."""
    R, B, k_cap = ids.shape
    out_ids = np.zeros((B, k_cap), ids.dtype)
    out_sc = np.zeros((B, k_cap), scores.dtype)
    out_cnt = np.zeros(B, np.int32)
    for b in range(B):
        if (counts[:, b] < 0).any():
            out_cnt[b] = counts[:, b].min()
            continue
        rows = [(scores[r, b, j], r, j) for r in range(R) for j in range(min(int(counts[r, b]), k_cap))]
        rows.sort(key=lambda t: (t[0], t[1], t[2]))
        total = len(rows)
        kq = total if (k <= 0 or k > total) else k
        kq = min(kq, k_cap)
        for i in range(kq):
            _, r, j = rows[i]
            out_ids[b, i] = ids[r, b, j]
            out_sc[b, i] = scores[r, b, j]
        out_cnt[b] = kq
    return out_ids, out_sc, out_cnt


def _rendezvous_id(lib, rank: int, world: int, addr: str, port: int, timeout_s: float | None = None) -> bytes:
    """Rank 0 makes the RCCL unique id and hands it to the other ranks over TCP (control plane only).
    The ranks of a fresh node start minutes apart (the first import of the runtime pages the image in, N processes at once): the wait is
    COMET_RDZV_TIMEOUT seconds, default 600."""
    import os
    import socket
    import time
    if timeout_s is None:
        timeout_s = float(os.environ.get("COMET_RDZV_TIMEOUT", "600"))
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        from ._lib import check
        check(lib.comet_comm_unique_id(buf))
        ident = bytes(buf)
        if world > 1:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port)); srv.listen(world)
            srv.settimeout(timeout_s)
            for _ in range(world - 1):
                conn, _a = srv.accept()
                conn.sendall(ident); conn.close()
            srv.close()
        return ident
    deadline = time.time() + timeout_s
    while True:
        try:
            s = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    data = b""
    while len(data) < 128:
        chunk = s.recv(128 - len(data))
        if not chunk:
            raise ConnectionError("rendezvous socket closed early")
        data += chunk
    s.close()
    return data


class Comm:
    """comet_comm: an RCCL communicator owned by the library, bound to a comet_amd.Context (one per process / GPU)."""

    def __init__(self, ctx, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29641):
        from ._lib import check
        self.ctx, self.lib, self.rank, self.world = ctx, ctx.lib, rank, world
        ident = _rendezvous_id(self.lib, rank, world, addr, port)
        buf = (C.c_uint8 * 128).from_buffer_copy(ident)
        self.h = C.c_void_p()
        check(self.lib.comet_comm_create(ctx.h, buf, rank, world, C.byref(self.h)))

    @classmethod
    def from_env(cls, ctx):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run exports them (the id socket uses MASTER_PORT + 101:
        the launcher's own store owns MASTER_PORT)."""
        import os
        port = int(os.environ.get("COMET_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29540")) + 101))
        return cls(ctx, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("MASTER_ADDR", "127.0.0.1"), port)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.comet_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        from ._lib import check
        check(self.lib.comet_comm_barrier(self.h))

    def sync(self):
        from ._lib import check
        check(self.lib.comet_comm_sync(self.h))

    def allreduce_max(self, x: float) -> float:
        from ._lib import check
        v = C.c_double(x)
        check(self.lib.comet_comm_allreduce_f64(self.h, C.byref(v), 0))
        return v.value

    def search_async(self, index, q_dev: int, B: int, k: int, out_ids_dev: int, out_scores_dev: int, out_counts_dev: int, k_cap: int,
                     threshold: float = 0.0, nprobes: int = 0, mode: int = 0) -> int:
        from ._lib import SearchParams, check
        p = SearchParams(k=int(k), threshold=float(threshold), nprobes=int(nprobes), ef_search=0, filter_ids=None, n_filter=0, mode=int(mode))
        t = C.c_uint64()
        check(self.lib.comet_index_search_sharded_async(index.h, self.h, C.c_void_p(q_dev), int(B), C.byref(p), C.c_void_p(out_ids_dev),
                                                        C.c_void_p(out_scores_dev), C.c_void_p(out_counts_dev), int(k_cap), C.byref(t)))
        return t.value

    def search_wait(self, index, ticket: int, block: bool = True) -> None:
        from ._lib import check
        check(self.lib.comet_index_search_sharded_wait(index.h, self.h, C.c_uint64(int(ticket)), 1 if block else 0))


class TopKExchange:
    """CPU (gloo) model of the exchange, used by the world-2 CPU test of the sharding bookkeeping: the same packed block
    [B*k_cap ids | B*k_cap scores | B counts] per rank, ONE all-gather, the same (score, shard, position) merge order — in
    numpy. The GPU data path does not come through here (Comm above)."""

    def __init__(self, B: int, k_cap: int, device, ctx=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.ctx = torch, dist, group, ctx
        self.world = dist.get_world_size(group)
        self.B, self.k_cap = B, k_cap
        self.block = 2 * B * k_cap + B                      # 32-bit words per rank
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self.pack = mk((self.block,), torch.int32)
        self.g_pack = mk((self.world, self.block), torch.int32)
        n = B * k_cap
        # typed views into this rank's block (what a search fills) and into the gathered blocks
        self.ids, self.scores, self.counts = self.pack[:n].view(B, k_cap), self.pack[n:2 * n].view(torch.float32).view(B, k_cap), self.pack[2 * n:]
        self.g_ids = self.g_pack[:, :n].view(torch.int32)
        self.m_ids, self.m_scores, self.m_counts = mk((B, k_cap), torch.int32), mk((B, k_cap), torch.float32), mk((B,), torch.int32)

    def local_ptrs(self):
        base = self.pack.data_ptr()
        n = self.B * self.k_cap
        return base, base + 4 * n, base + 8 * n

    def exchange_and_merge(self, k: int):
        """self.ids/scores/counts hold this rank's rows; returns the merged (ids, scores, counts) tensors."""
        d = self.dist
        W, B, K = self.world, self.B, self.k_cap
        d.all_gather_into_tensor(self.g_pack.view(W * self.block), self.pack, group=self.group)
        if self.pack.is_cuda:
            raise RuntimeError("TopKExchange is the CPU model of the exchange; on GPUs use comet_amd.dist.Comm (RCCL inside libcomet_hip.so)")
        else:
            g = self.g_pack.numpy()
            n = B * K
            gi = np.ascontiguousarray(g[:, :n]).view(np.uint32).reshape(W, B, K)
            gs = np.ascontiguousarray(g[:, n:2 * n]).view(np.float32).reshape(W, B, K)
            gc = np.ascontiguousarray(g[:, 2 * n:]).reshape(W, B)
            i, s, c = merge_topk_host(gi, gs, gc, k)
            self.m_ids.copy_(self.torch.from_numpy(i.view(np.int32)))
            self.m_scores.copy_(self.torch.from_numpy(s))
            self.m_counts.copy_(self.torch.from_numpy(c))
        return self.m_ids, self.m_scores, self.m_counts
