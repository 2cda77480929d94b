#!/usr/bin/env python3
"""adc_scaling.py — adc_scan_kernel's time against the batch size (every-candidate search, mode 1) on UNIFORM rows: what part of a launch is fixed cost.
usage: adc_scaling.py [rows] [nlist]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import comet_amd as ca  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = 768
ctx = ca.Context(0)
ctx.set_lanes(1)
idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, nlist, 96, 8)
fill = lambda buf, lo, m: ctx.synth_fill(buf, 0xC0FFEE + 3, lo * d, m * d)
nt = nlist * 100
tb = ctx.alloc(nt * d * 4); fill(tb, 0, nt); idx.train_dev(tb, nt); ctx.free(tb)
bench.add_rows(ctx, idx, 0, rows, d, fill)
Bmax = 1024
q = ctx.alloc(Bmax * d * 4); ctx.synth_fill(q, 0xBEEF + 3, 0, Bmax * d)
o = (ctx.alloc(Bmax * 40), ctx.alloc(Bmax * 40), ctx.alloc(Bmax * 4))
for mode in (1, 0):
    for B in (8, 16, 32, 64, 128, 256, 512, 1024):
        for _ in range(3):
            idx.search_batch_dev(q, B, 10, *o, 10, nprobes=32, mode=mode)
        ctx.sync(); ctx.profile(True); ctx.profile_reset()
        n = 10
        for _ in range(n):
            idx.search_batch_dev(q, B, 10, *o, 10, nprobes=32, mode=mode)
        ctx.sync()
        p = ctx.profile_dump(); ctx.profile(False)
        print(f"mode {mode} B {B:5d}: " + "  ".join(f"{k} {v[0] / n:.4f} ms ({v[1] // n}x)" for k, v in sorted(p.items()) if k in ("adc_scan", "pq_lut", "adc_order", "pq_bound", "coarse_pick", "select_composites", "sel_composites")), flush=True)
