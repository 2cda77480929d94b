#!/usr/bin/env python3
"""a2_trace.py — per-step trace of adc_scan2_kernel (needs a library built with -DA2_TRACE: comet_debug_a2_trace). IVFPQ on UNIFORM rows, every-candidate search;
prints shader-clock (s_memtime, 100 MHz) intervals of workgroup 8's first batch for an early and a late wave.
usage: a2_trace.py [rows] [nlist] [B]"""
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
buf = (C.c_ulonglong * 1024)()
ctx.lib.comet_debug_a2_trace.restype = C.c_int
assert ctx.lib.comet_debug_a2_trace(buf, 1024) == 0
t = np.array(buf[:], dtype=np.int64).reshape(2, 64, 8)
for w, name in ((0, "wave 0 (early: barrier, build next, gather)"), (1, "wave 4 (late: build, barrier, gather)")):
    r = t[w]
    print(name)
    print(f"  batch: prologue {r[63][1]-r[63][0]}, loop {r[63][2]-r[63][1]}, park {r[63][3]-r[63][2]}, epilogues {r[63][4]-r[63][3]}, total {r[63][4]-r[63][0]} ticks (x clock/100MHz shader clocks)")
    print(f"  prologue: top barrier {r[61][0]-r[63][0]}, records {r[61][1]-r[61][0]}, residuals {r[61][2]-r[61][1]}, bounds+codewords {r[61][3]-r[61][2]}, barrier {r[61][4]-r[61][3]}, first build {r[63][1]-r[61][4]}; epilogue: staging {r[61][5]-r[63][3]}, flush {r[63][4]-r[61][5]}")
    print("  chains of this wave per item:", [int(x) & 0xFFFF for x in r[62][:4]], "items in the batch:", int(r[62][0]) >> 32)
    print("  step | barrier+cw | build | late barrier | gathers | total")
    for k in range(0, 24):
        s = r[k]
        if s[0] == 0:
            break
        print(f"  {k:4d} | {s[1]-s[0]:6d} | {s[2]-s[1]:6d} | {s[3]-s[2]:6d} | {s[4]-s[3]:6d} | {s[4]-s[0]:6d}")
