"""PQ (no coarse quantiser) search at scale: 1M x 768, M = 96, 8 bits — every query scans every code (pq_index_search.go)."""
import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import comet_amd as ca
import oracle_lib as orc
ctx = ca.Context(0)
n, d, M, B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768, 96, 256, 10
idx = ca.PQIndex(ctx, d, ca.L2_SQUARED, M, 8)
rows = lambda lo, hi: orc.synth(0xC0FFEE + 9, lo * d, (hi - lo) * d).reshape(hi - lo, d)
t0 = time.time(); idx.train(rows(0, 25600)); train_s = time.time() - t0
t0 = time.time()
for lo in range(0, n, 131072):
    hi = min(n, lo + 131072)
    idx.add_batch(np.arange(lo + 1, hi + 1, dtype=np.uint32), rows(lo, hi))
add_s = time.time() - t0
Q = orc.synth(0xBEEF + 9, 0, B * d).reshape(B, d)
q_dev = ctx.alloc(B * d * 4); ctx.upload(q_dev, Q)
oi, os_, oc = ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)
idx.search_batch_dev(q_dev, B, K, oi, os_, oc, K); ctx.sync()
ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    idx.search_batch_dev(q_dev, B, K, oi, os_, oc, K)
ctx.sync()
el = (time.perf_counter() - t0) / steps
prof = {k: round(v[0] / steps, 4) for k, v in ctx.profile_dump().items()}
adc = prof.get("adc_scan", 0.0)
print(json.dumps({"workload": f"PQ L2^2 {n}x{d}, M={M} nbits=8, batch={B}, K={K} (GPU train {train_s:.1f}s, add {add_s:.1f}s)", "qps": B / el, "ms_per_batch": el * 1e3,
                  "kernels_ms_per_batch": prof, "adc_code_bytes": n * M * B, "adc_GBps": n * M * B / (adc * 1e-3) / 1e9 if adc else None}))
