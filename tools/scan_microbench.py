#!/usr/bin/env python3
"""Small driver for profiling the Flat fast-path kernels under rocprofv3 --pmc (few launches, short build).
usage: scan_microbench.py [rows] [batch] [iters] [mode]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import comet_amd as ca
import ctypes as C
from comet_amd._lib import check

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dim, K = 768, 10
ctx = ca.Context(0)
idx = ca.FlatIndex(ctx, dim, ca.COSINE)
chunk = 65536
buf = ctx.alloc(chunk * dim * 4); idb = ctx.alloc(chunk * 4)
for lo in range(0, rows, chunk):
    m = min(chunk, rows - lo)
    ctx.synth_fill(buf, 0xC0FFEE, lo * dim, m * dim)
    ctx.upload(idb, np.arange(lo + 1, lo + m + 1, dtype=np.uint32))
    added = C.c_int64()
    check(ctx.lib.comet_index_add_dev(idx.h, C.c_void_p(idb), C.c_void_p(buf), m, C.byref(added)))
q = ctx.alloc(B * dim * 4); ctx.synth_fill(q, 0xBEEF, 0, B * dim)
oi, os_, oc = ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)
ctx.sync()
ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(iters):
    idx.search_batch_dev(q, B, K, oi, os_, oc, K, mode=mode)
ctx.sync()
el = time.perf_counter() - t0
print("ms/iter", el / iters * 1e3, {k: round(v[0] / v[1], 4) for k, v in ctx.profile_dump().items()})
