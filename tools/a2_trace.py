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
# stamps (kernels_adc2.inc.hpp, -DA2_TRACE, workgroup 8, batch A2_TRACE_BATCH): row 63 = [batch start (behind the previous batch's epilogue), first table built, loop done, sums parked,
# next loop-top]; row 61 = [next batch's requests issued, epilogue staged, flushed, residuals + bounds formed, barrier passed] (61[0..2] belong to the NEXT iteration's top: they are written
# while `trace_on` still holds the traced batch); rows 0..23: per step [start, barrier + code-word requests, build, late barrier, gathers]
for w, name in ((0, "wave 0 (early: barrier, build next, gather)"), (1, "wave 4 (late: build, barrier, gather)")):
    r = t[w]
    print(name)
    print(f"  batch: residuals + bounds + first table {r[63][1]-r[63][0]}, step loop {r[63][2]-r[63][1]}, sums parked {r[63][3]-r[63][2]}; then (next iteration's top) requests for the next batch "
          f"{r[61][0]-r[63][4]}, this batch's epilogue staged {r[61][1]-r[61][0]}, flushed {r[61][2]-r[61][1]}  [shader clocks]")
    print("  chains of this wave per item:", [int(x) & 0xFFFF for x in r[62][:4]], "items in the batch:", int(r[62][0]) >> 32)
    print("  step | barrier+cw | build | late barrier | gathers | total")
    tot = 0
    for k in range(0, 24):
        s_ = r[k]
        if s_[0] == 0:
            break
        tot += s_[4] - s_[0]
        print(f"  {k:4d} | {s_[1]-s_[0]:6d} | {s_[2]-s_[1]:6d} | {s_[3]-s_[2]:6d} | {s_[4]-s_[3]:6d} | {s_[4]-s_[0]:6d}")
    ch = sum(int(x) & 0xFFFF for x in r[62][:4])
    print(f"  LDS-pipe time of the batch's gathers at 7 clocks per wave-gather: ~{ch * 8 * 96 * 7} clocks (if every wave held this wave's {ch} chains) against {tot} in the step loop")
