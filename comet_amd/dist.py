"""Multi-GPU exchange for row-sharded indexes: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm), every rank searches its own shard for the same query batch, the per-shard top-K rows are
exchanged with ONE all-gather per array per batch, and every rank merges them.

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


class TopKExchange:
    """Pre-allocated buffers + ONE all-gather per batch. A rank's result block is one packed int32 buffer
    [B*k_cap ids | B*k_cap scores | B counts]; the search writes its three outputs straight into it (local_ptrs), the
    all-gather stacks the W blocks, and comet_merge_topk_packed_dev merges them. `ctx` is a comet_amd.Context for CUDA
    tensors, None for the CPU/gloo path."""

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
            from ._lib import check
            self.torch.cuda.current_stream().synchronize()      # the merge runs on the library's own stream
            check(self.ctx.lib.comet_merge_topk_packed_dev(self.ctx.h, C.c_void_p(self.g_pack.data_ptr()), self.block, W, B, K, int(k),
                                                           C.c_void_p(self.m_ids.data_ptr()), C.c_void_p(self.m_scores.data_ptr()),
                                                           C.c_void_p(self.m_counts.data_ptr())))
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
