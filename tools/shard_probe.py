#!/usr/bin/env python3
"""shard_probe.py — single-GPU measurement of what ONE rank of an N-GPU job does per batch, for the scaling model in DESIGN.md.
For N in 1, 2, 4, 8 it builds rank 0's shard of the bench corpus (Flat: rows/N contiguous rows; IVFPQ: the lists l % N == 0 of
the 1M-row index), runs the bench loop through the in-library RCCL path at world size 1 (the all-gather then moves one block;
its N-rank cost is modelled separately from the block size) and prints ms/step with the per-kernel breakdown."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (generators / constants only)
import comet_amd as ca  # noqa: E402
from comet_amd._lib import check  # noqa: E402
from comet_amd.dist import Comm  # noqa: E402

ctx = ca.Context(0)
comm = Comm(ctx, 0, 1, port=29777)
dim, B, rows = 768, 256, 1_000_000
steps, regions = 20, 5
out = {"flat": {}, "ivfpq_lists": {}, "ivfpq_members": {}}


def loop(idx, q_dev, K, **kw):
    ptrs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(3)]

    def run(n):
        prev = None
        for i in range(n):
            t = comm.search_async(idx, q_dev, B, K, *ptrs[i % 3], K, **kw)
            if prev is not None:
                comm.search_wait(idx, prev, block=False)
            prev = t
        comm.search_wait(idx, prev, block=True)
    run(3); comm.sync()
    ctx.profile(True); ctx.profile_reset()
    times = []
    for _ in range(regions):
        comm.sync(); t0 = time.perf_counter(); run(steps); comm.sync(); times.append(time.perf_counter() - t0)
    prof = ctx.profile_dump(); ctx.profile(False)
    for b in ptrs:
        for p in b:
            ctx.free(p)
    tot = steps * regions
    return {"ms_per_step": sorted(times)[len(times) // 2] / steps * 1e3, "kernels_ms_per_step": {k: round(v[0] / tot, 4) for k, v in sorted(prof.items())}}


q_dev = ctx.alloc(B * dim * 4)
for N in (1, 2, 4, 8):
    idx = ca.FlatIndex(ctx, dim, ca.COSINE)
    bench.add_rows(ctx, idx, 0, rows // N, dim, lambda buf, r0, m: ctx.synth_fill(buf, bench.CORPUS_SEED, r0 * dim, m * dim))
    ctx.synth_fill(q_dev, bench.QUERY_SEED, 0, B * dim)
    out["flat"][N] = loop(idx, q_dev, 100)
    print("flat", N, out["flat"][N], flush=True)
    idx.close()
mix = lambda buf, r0, m: ctx.synth_mixture(buf, bench.MIX_SEED, bench.MIX_CENTERS, bench.MIX_SIGMA, bench.MIX_SUB, bench.MIX_NOISE, r0, m, dim)
ctx.synth_mixture(q_dev, bench.MIX_SEED, bench.MIX_CENTERS, bench.MIX_SIGMA, bench.MIX_SUB, bench.MIX_NOISE, rows + 7, B, dim)
ntrain = 102400
tbuf = ctx.alloc(ntrain * dim * 4)
mix(tbuf, 0, ntrain)
for policy in ("lists", "members"):
    for N in (1, 2, 4, 8):
        idx = ca.IVFPQIndex(ctx, dim, ca.L2_SQUARED, 1024, 96, 8)
        check(ctx.lib.comet_index_train_dev(idx.h, C.c_void_p(tbuf), ntrain))
        if policy == "lists":
            if N > 1:
                idx.set_shard(0, N)
            bench.add_rows(ctx, idx, 0, rows, dim, mix)
        else:
            # rank 0's round-robin share of the members: rows 0, N, 2N, ... generated one chunk at a time
            chunk = 65536
            buf = ctx.alloc(chunk * dim * 4); sel = ctx.alloc(chunk // N * dim * 4 + 4096)
            for lo in range(0, rows, chunk):
                m = min(chunk, rows - lo)
                mix(buf, lo, m); ctx.sync()
                X = ctx.download(buf, (m, dim), np.float32)[(-lo) % N::N]
                idx.add_batch(np.arange(lo + 1, lo + m + 1, dtype=np.uint32)[(-lo) % N::N], X)
            ctx.free(buf); ctx.free(sel)
        out[f"ivfpq_{policy}"][N] = loop(idx, q_dev, 10, nprobes=32)
        out[f"ivfpq_{policy}"][N]["rows_on_rank"] = len(idx)
        print("ivfpq", policy, N, out[f"ivfpq_{policy}"][N], flush=True)
        idx.close()
print(json.dumps(out))
