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
d, B = 768, 256
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
buf = (C.c_ulonglong * 512)()
ctx.lib.comet_debug_adc_trace.restype = C.c_int
assert ctx.lib.comet_debug_adc_trace(buf, 512) == 0
t = np.array(buf[:], dtype=np.uint64).reshape(32, 16)
print("item | wait+barrier ph0 | gathers ph0 | wait+barrier ph1 | gathers ph1 | wait+barrier ph2 | gathers ph2 | epilogue | total   (shader clocks)")
for i in range(1, 14):
    r = t[i].astype(np.int64)
    if r[0] == 0 or r[15] == 0:
        break
    print(f"{i:4d} | {r[1]-r[0]:7d} | {r[2]-r[1]:7d} | {r[3]-r[2]:7d} | {r[4]-r[3]:7d} | {r[5]-r[4]:7d} | {r[6]-r[5]:7d} | {r[15]-r[6]:7d} | {r[15]-r[0]:7d}")
