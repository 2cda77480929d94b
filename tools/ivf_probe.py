"""IVF list scan, fast path vs exact kernels at configs[4]'s vector-leg shape (1M x 768, nlist 1024, B 256, K 10): ms per batch with
device-resident queries / results and two batches in flight (what bench.py measures), the per-kernel breakdown, and a bit-for-bit
comparison of the two result sets. Usage: python tools/ivf_probe.py [rows] [dim] [cosine|l2sq] [blobs|mixture] [nprobes,...]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import comet_amd as ca
import oracle_lib as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = ca.COSINE if (len(sys.argv) <= 3 or sys.argv[3] == "cosine") else ca.L2_SQUARED
corpus = sys.argv[4] if len(sys.argv) > 4 else "blobs"
probes = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "1,8,32").split(",")]
nlist, B, K, steps = 1024, 256, 10, 40
ctx = ca.Context(0)
centers = orc.synth(0x5EED, 0, 2048 * d).reshape(2048, d)


def rows(lo, hi):
    noise = orc.synth(0xC0FFEE + 4, lo * d, (hi - lo) * d).reshape(hi - lo, d)
    blob = ((np.arange(lo, hi, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(2048)
    return (centers[blob.astype(np.int64)] + noise * np.float32(0.15)).astype(np.float32)


ivf = ca.IVFIndex(ctx, d, nlist, metric)
if corpus == "mixture":       # the Flat leg's two-level mixture (bench.py): 2048 centres, 65536 sub-centres at 0.15, noise 0.02, generated on the device
    buf = ctx.alloc(131072 * d * 4)
    ctx.synth_mixture(buf, 0xC0FFEE + 7, 2048, 0.15, 65536, 0.02, 0, nlist * 100, d)
    t0 = time.time(); ivf.train_dev(buf, nlist * 100); train_s = time.time() - t0
    t0 = time.time()
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        ctx.synth_mixture(buf, 0xC0FFEE + 7, 2048, 0.15, 65536, 0.02, lo, hi - lo, d)
        ivf.add_batch_dev(np.arange(lo + 1, hi + 1, dtype=np.uint32), buf, hi - lo)
    add_s = time.time() - t0
    ctx.synth_mixture(buf, 0xC0FFEE + 7, 2048, 0.15, 65536, 0.02, 5_000_000, B, d)       # queries: fresh rows of the same mixture
    Q = ctx.download(buf, (B, d), np.float32)
    ctx.free(buf)
else:
    t0 = time.time(); ivf.train(rows(0, nlist * 100)); train_s = time.time() - t0
    t0 = time.time()
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        ivf.add_batch(np.arange(lo + 1, hi + 1, dtype=np.uint32), rows(lo, hi))
    add_s = time.time() - t0
    qr = (np.arange(B) * 7919) % n
    Q = np.vstack([rows(int(r), int(r) + 1) for r in qr]) + orc.synth(0xBEEF + 4, 0, B * d).reshape(B, d) * np.float32(0.05)
out = {"workload": f"IVF {'cosine' if metric == ca.COSINE else 'l2sq'} {n}x{d} nlist {nlist} B {B} K {K} corpus {corpus}", "train_s": round(train_s, 2), "add_s": round(add_s, 2),
       "max_list_len": ivf.stat("max_list_len")}
q_dev = ctx.alloc(B * d * 4); ctx.upload(q_dev, Q)
bufs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(2)]


def run(npb, mode, nsteps):
    prev = None
    for i in range(nsteps):
        w = i & 1
        t = ivf.search_batch_dev_async(q_dev, B, K, bufs[w][0], bufs[w][1], bufs[w][2], K, nprobes=npb, mode=mode)
        if prev is not None:
            ivf.search_wait(prev)
        prev = t
    ivf.search_wait(prev)
    ctx.sync()


for npb in probes:
    res = {}
    ref = None
    for mode, name in ((1, "strict"), (0, "auto")):
        run(npb, mode, 4)
        t0 = time.perf_counter(); run(npb, mode, steps); el = (time.perf_counter() - t0) / steps
        ctx.profile(True); ctx.profile_reset(); run(npb, mode, 10); prof = ctx.profile_dump(); ctx.profile(False)
        r = ivf.search_batch(Q, K, nprobes=npb, mode=mode)
        res[name] = {"ms_per_batch": round(el * 1e3, 4), "qps": round(B / el), "kernels_ms": {k: round(v[0] / 10, 4) for k, v in sorted(prof.items())}}
        if mode == 0:
            res[name]["fast_queries"] = ivf.stat("fast_queries"); res[name]["candidates_per_query"] = ivf.stat("fast_candidates") / B
            res[name]["overflows"] = ivf.stat("fast_overflows"); res[name]["scan_rows"] = ivf.stat("ivf_scan_rows")
            sk = "ivf_scan_i8" if "ivf_scan_i8" in res[name]["kernels_ms"] else "ivf_scan_f16"
            if res[name]["scan_rows"] and sk in res[name]["kernels_ms"]:
                by = res[name]["scan_rows"] * ((d + 127) // 128 * 128 if sk == "ivf_scan_i8" else (d + 63) // 64 * 64 * 2)
                res[name]["scan_bytes"] = by; res[name]["scan_TBps"] = round(by / (res[name]["kernels_ms"][sk] * 1e-3) / 1e12, 3)
            res["identical"] = bool(np.array_equal(r[2], ref[2]) and np.array_equal(r[0], ref[0]) and np.array_equal(r[1].view(np.uint32), ref[1].view(np.uint32)))
        else:
            ref = r
    out[f"nprobe{npb}"] = res
print(json.dumps(out))
