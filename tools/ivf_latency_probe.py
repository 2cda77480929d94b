"""IVF search latency per call at small batches (the reference runs ONE query per Execute()): fast path vs exact kernels, host buffers,
1M x 768 mixture corpus and a small 50k x 128 index. usage: python tools/ivf_latency_probe.py"""
import json, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import comet_amd as ca, bench
ctx = ca.Context(0)
out = {}
for (n, d, nlist) in ((1_000_000, 768, 1024), (50_000, 128, 64)):
    ivf = ca.IVFIndex(ctx, d, nlist, ca.COSINE)
    nt = nlist * 100 if n >= nlist * 100 else n
    buf = ctx.alloc(max(nt, 65536) * d * 4)
    bench.mix_fill(ctx, d)(buf, 0, nt); ivf.train_dev(buf, nt); ctx.free(buf)
    bench.add_rows(ctx, ivf, 0, n, d, bench.mix_fill(ctx, d))
    qb = ctx.alloc(64 * d * 4); ctx.synth_mixture(qb, bench.MIX_SEED, bench.MIX_CENTERS, bench.MIX_SIGMA, bench.MIX_SUB, bench.MIX_NOISE, n + 7, 64, d)
    Q = ctx.download(qb, (64, d), np.float32)
    for B in (1, 8, 64):
        for npb in (1, 8, 32):
            r = {}
            for mode, name in ((1, "exact"), (2, "fast")):
                ivf.search_batch(Q[:B], 10, nprobes=npb, mode=mode)
                t0 = time.perf_counter()
                for i in range(30):
                    ivf.search_batch(Q[:B], 10, nprobes=npb, mode=mode)
                r[name] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
            out[f"{n}x{d} nlist{nlist} B{B} nprobe{npb}"] = r
    ivf.close()
print(json.dumps(out, indent=0))
