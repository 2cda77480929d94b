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
    """Pre-allocated buffers + one all-gather per array per batch. `ctx` is a comet_amd.Context for CUDA tensors,
    None for the CPU/gloo path."""

    def __init__(self, B: int, k_cap: int, device, ctx=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.ctx = torch, dist, group, ctx
        self.world = dist.get_world_size(group)
        self.B, self.k_cap = B, k_cap
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self.ids, self.scores, self.counts = mk((B, k_cap), torch.int32), mk((B, k_cap), torch.float32), mk((B,), torch.int32)
        self.g_ids, self.g_scores, self.g_counts = mk((self.world, B, k_cap), torch.int32), mk((self.world, B, k_cap), torch.float32), mk((self.world, B), torch.int32)
        self.m_ids, self.m_scores, self.m_counts = mk((B, k_cap), torch.int32), mk((B, k_cap), torch.float32), mk((B,), torch.int32)

    def local_ptrs(self):
        return self.ids.data_ptr(), self.scores.data_ptr(), self.counts.data_ptr()

    def exchange_and_merge(self, k: int):
        """self.ids/scores/counts hold this rank's rows; returns the merged (ids, scores, counts) tensors."""
        d = self.dist
        W, B, K = self.world, self.B, self.k_cap     # flattened [W*B, K] views: both RCCL and gloo accept them
        d.all_gather_into_tensor(self.g_ids.view(W * B, K), self.ids, group=self.group)
        d.all_gather_into_tensor(self.g_scores.view(W * B, K), self.scores, group=self.group)
        d.all_gather_into_tensor(self.g_counts.view(W * B), self.counts, group=self.group)
        if self.ids.is_cuda:
            from ._lib import check
            self.torch.cuda.current_stream().synchronize()      # the merge runs on the library's own stream
            check(self.ctx.lib.comet_merge_topk_dev(self.ctx.h, C.c_void_p(self.g_ids.data_ptr()), C.c_void_p(self.g_scores.data_ptr()),
                                                    C.c_void_p(self.g_counts.data_ptr()), self.world, self.B, self.k_cap, int(k),
                                                    C.c_void_p(self.m_ids.data_ptr()), C.c_void_p(self.m_scores.data_ptr()),
                                                    C.c_void_p(self.m_counts.data_ptr())))
        else:
            i, s, c = merge_topk_host(self.g_ids.numpy().view(np.uint32), self.g_scores.numpy(), self.g_counts.numpy(), k)
            self.m_ids.copy_(self.torch.from_numpy(i.view(np.int32)))
            self.m_scores.copy_(self.torch.from_numpy(s))
            self.m_counts.copy_(self.torch.from_numpy(c))
        return self.m_ids, self.m_scores, self.m_counts
