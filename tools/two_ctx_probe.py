"""What a second execution lane could buy: two contexts (two HIP streams) on the one GPU, each with its own copy of the index, searched
alternately from one host thread — against one context alone. Usage: python tools/two_ctx_probe.py [ivf|ivfpq|flat|hnsw] [contexts = 2]
(COMET_LANES=1 in the environment isolates the effect: each context then is exactly one stream)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import comet_amd as ca

kind = sys.argv[1] if len(sys.argv) > 1 else "ivf"
NCTX = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n, d, B, K, nlist, steps = 1_000_000, 768, 256, 10, 1024, 200
if kind == "hnsw":
    n, d = 50_000, 384


def build(ctx):
    buf = ctx.alloc(131072 * d * 4)
    if kind == "flat":
        idx = ca.FlatIndex(ctx, d, ca.COSINE)
    elif kind == "hnsw":
        idx = ca.HNSWIndex(ctx, d, ca.L2_SQUARED, 16, 200, 128)
    else:
        idx = ca.IVFIndex(ctx, d, nlist, ca.COSINE) if kind == "ivf" else ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, nlist, 96, 8)
        ctx.synth_mixture(buf, 0xC0FFEE + 7, 2048, 0.15, 65536, 0.02, 0, nlist * 100, d)
        idx.train_dev(buf, nlist * 100)
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        ctx.synth_mixture(buf, 0xC0FFEE + 7, 2048, 0.15, 65536, 0.02, lo, hi - lo, d)
        idx.add_batch_dev(np.arange(lo + 1, hi + 1, dtype=np.uint32), buf, hi - lo)
    q = ctx.alloc(B * d * 4)
    ctx.synth_mixture(q, 0xC0FFEE + 7, 2048, 0.15, 65536, 0.02, 5_000_000, B, d)
    ctx.free(buf)
    bufs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(2)]
    return idx, q, bufs


kw = {} if kind in ("flat", "hnsw") else {"nprobes": 32}
ctxs = [ca.Context(0) for _ in range(NCTX)]
sets = [build(c) for c in ctxs]


def run(active, nsteps):
    prev = [None] * len(active)
    for i in range(nsteps):
        for a, s in enumerate(active):
            idx, q, bufs = sets[s]
            t = idx.search_batch_dev_async(q, B, K, *bufs[i & 1], K, **kw)
            if prev[a] is not None:
                idx.search_wait(prev[a])
            prev[a] = t
    for a, s in enumerate(active):
        sets[s][0].search_wait(prev[a])
    for c in ctxs:
        c.sync()


out = {}
for name, active in [("one_context", [0])] + [(f"{m}_contexts", list(range(m))) for m in range(2, NCTX + 1)] + [("one_context_again", [NCTX - 1])]:
    run(active, 10)
    t0 = time.perf_counter(); run(active, steps); el = time.perf_counter() - t0
    out[name] = {"batches_per_s": round(len(active) * steps / el), "qps": round(len(active) * steps * B / el), "ms_per_batch": round(el / (len(active) * steps) * 1e3, 4)}
print(json.dumps({"kind": kind, **out}))
