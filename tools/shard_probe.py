import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, ctypes as C
import comet_amd as ca
from comet_amd._lib import check
ctx = ca.Context(0)
dim, K, B = 768, 100, 256
for rows in (500000, 250000, 125000):
    idx = ca.FlatIndex(ctx, dim, ca.COSINE)
    chunk = 62500
    buf = ctx.alloc(chunk * dim * 4); idb = ctx.alloc(chunk * 4)
    for lo in range(0, rows, chunk):
        m = min(chunk, rows - lo)
        ctx.synth_fill(buf, 0xC0FFEE, lo * dim, m * dim)
        ctx.upload(idb, np.arange(lo + 1, lo + m + 1, dtype=np.uint32))
        added = C.c_int64()
        check(ctx.lib.comet_index_add_dev(idx.h, C.c_void_p(idb), C.c_void_p(buf), m, C.byref(added)))
    q = ctx.alloc(B * dim * 4); ctx.synth_fill(q, 0xBEEF, 0, B * dim)
    oi, os_, oc = ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)
    for mode in (2, 1, 0):
        idx.search_batch_dev(q, B, K, oi, os_, oc, K, mode=mode); ctx.sync()
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(10):
            idx.search_batch_dev(q, B, K, oi, os_, oc, K, mode=mode)
        ctx.sync()
        el = (time.perf_counter() - t0) / 10
        prof = {k: round(v[0] / 10, 4) for k, v in ctx.profile_dump().items()}; ctx.profile(False)
        st = {k: idx.stat(k) for k in ("fast_queries", "strict_queries", "fast_candidates", "fast_expansions", "fast_overflows")}
        print(rows, "mode", mode, "ms/step %.3f" % (el * 1e3), prof, st, flush=True)
    for p in (buf, idb, q, oi, os_, oc): ctx.free(p)
    del idx
