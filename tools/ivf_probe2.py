"""debug: wall time of the IVF fast path at nprobe 32 with and without per-kernel profiling, host vs device buffers"""
import json, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import comet_amd as ca
import oracle_lib as orc
n, d, nlist, B, K = 1_000_000, 768, 1024, 256, 10
ctx = ca.Context(0)
centers = orc.synth(0x5EED, 0, 2048 * d).reshape(2048, d)
def rows(lo, hi):
    noise = orc.synth(0xC0FFEE + 4, lo * d, (hi - lo) * d).reshape(hi - lo, d)
    blob = ((np.arange(lo, hi, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(2048)
    return (centers[blob.astype(np.int64)] + noise * np.float32(0.15)).astype(np.float32)
ivf = ca.IVFIndex(ctx, d, nlist, ca.COSINE)
ivf.train(rows(0, nlist * 100))
for lo in range(0, n, 131072):
    hi = min(n, lo + 131072)
    ivf.add_batch(np.arange(lo + 1, hi + 1, dtype=np.uint32), rows(lo, hi))
qr = (np.arange(B) * 7919) % n
Q = np.vstack([rows(int(r), int(r) + 1) for r in qr]) + orc.synth(0xBEEF + 4, 0, B * d).reshape(B, d) * np.float32(0.05)
out = {}
for npb in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16,24,32").split(",")]:
    ivf.search_batch(Q, K, nprobes=npb)
    t0 = time.perf_counter()
    for _ in range(20):
        ivf.search_batch(Q, K, nprobes=npb)
    out[f"np{npb}_host_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ivf.search_batch(Q, K, nprobes=npb); ts.append((time.perf_counter() - t0) * 1e3)
    out[f"np{npb}_single_ms"] = ts
print(json.dumps(out))
