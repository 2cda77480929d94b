#!/usr/bin/env python3
"""adc_trace.py — per-item phase trace of adc_scan_kernel (needs a library built with -DADC_TRACE: comet_debug_adc_trace). IVFPQ 1M x 768 on UNIFORM rows
(lists of ~1000 codes: the short-list regime), every-candidate search; prints shader-clock intervals of workgroup 8's first items.
usage: adc_trace.py [rows] [nlist]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import comet_amd as ca  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d, B = 768, (int(sys.argv[3]) if len(sys.argv) > 3 else 256)
ctx = ca.Context(0)
idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, nlist, 96, 8)
fill = lambda buf, lo, m: ctx.synth_fill(buf, 0xC0FFEE + 3, lo * d, m * d)
nt = nlist * 100
tb = ctx.alloc(nt * d * 4); fill(tb, 0, nt); idx.train_dev(tb, nt); ctx.free(tb)
bench.add_rows(ctx, idx, 0, rows, d, fill)
q = ctx.alloc(B * d * 4); ctx.synth_fill(q, 0xBEEF + 3, 0, B * d)
o = (ctx.alloc(B * 40), ctx.alloc(B * 40), ctx.alloc(B * 4))
for _ in range(3):
    idx.search_batch_dev(q, B, 10, *o, 10, nprobes=32, mode=1)
ctx.sync()
ctx.profile(True); ctx.profile_reset()
idx.search_batch_dev(q, B, 10, *o, 10, nprobes=32, mode=1)
ctx.sync()
print("adc_scan of the traced launch:", {k: v for k, v in ctx.profile_dump().items() if k == "adc_scan"}, "ms")
buf = (C.c_ulonglong * 512)()
ctx.lib.comet_debug_adc_trace.restype = C.c_int
assert ctx.lib.comet_debug_adc_trace(buf, 512) == 0
t = np.array(buf[:], dtype=np.uint64).reshape(32, 16)
# tables built in LDS (adc_scan_kernel<DSUB>): wave 0 builds its share of the next slab AFTER its gathers (rows 0..15), wave 1 BEFORE them (rows 16..31); slots 7 + ph = build done
for wv, base in (("wave 0 (gathers, then build)", 0), ("wave 1 (build, then gathers)", 16)):
    print(wv)
    print("item | barrier ph0 | gathers ph0 | build | barrier ph1 | gathers ph1 | build | barrier ph2 | gathers ph2 | build | epilogue | total   (s_memtime ticks)")
    for i in range(0, 14):
        r = t[base + i].astype(np.int64)
        if r[0] == 0 or r[15] == 0:
            break
        late = base == 0
        def seg(ph):
            bar = r[1 + 2 * ph] - (r[0] if ph == 0 else max(r[2 * ph], r[6 + ph]))
            if late:
                return bar, r[2 + 2 * ph] - r[1 + 2 * ph], (r[7 + ph] - r[2 + 2 * ph]) if r[7 + ph] else 0
            return bar, r[2 + 2 * ph] - max(r[7 + ph], r[1 + 2 * ph]), (r[7 + ph] - r[1 + 2 * ph]) if r[7 + ph] else 0
        a, b, c_ = seg(0), seg(1), seg(2)
        if i == 0:
            print(f"   (item 0 starts at tick {r[0]}, the last traced item of the wave ends at {max(t[base + j][15] for j in range(16))}: {max(int(t[base + j][15]) for j in range(16)) - int(r[0])} ticks)")
        e0 = max(r[6], r[9])
        print(f"       epilogue: bound A arrived +{r[10]-e0}, keys A +{r[11]-r[10]}, rest of A +{(r[12]-r[11]) if r[12] else 0}, bound B.. keys B +{(r[13]-r[12]) if r[13] and r[12] else 0}, rest +{r[15]-max(r[13], r[11])}")
        print(f"{i:4d} | {a[0]:7d} | {a[1]:7d} | {a[2]:6d} | {b[0]:7d} | {b[1]:7d} | {b[2]:6d} | {c_[0]:7d} | {c_[1]:7d} | {c_[2]:6d} | {r[15]-max(r[6], r[9]):7d} | {r[15]-r[0]:7d}")
