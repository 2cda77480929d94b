"""What a batch of non-finite / garbage queries does to every index kind (a caller's bug must come back as results or an error, never as a dead process).
usage: python tools/nan_query_probe.py [kind ...]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as orc
import comet_amd as ca

ctx = ca.Context(0)
n, d, B, k = 20000, 64, 64, 5
X = orc.synth(191, 0, n * d).reshape(n, d)
ids = np.arange(1, n + 1, dtype=np.uint32)
kinds = sys.argv[1:] or ["flat", "ivf", "pq", "ivfpq", "hnsw"]
rng = np.random.default_rng(5)
pats = {"nan": np.full((B, d), np.nan, np.float32), "inf": np.full((B, d), np.inf, np.float32), "huge": np.full((B, d), 3e38, np.float32),
        "bits": rng.integers(0, 2**32, (B, d), dtype=np.uint64).astype(np.uint32).view(np.float32), "mixed": None}
m = orc.synth(7, 0, B * d).reshape(B, d).copy(); m[::3, 5] = np.nan; m[1::3, 7] = np.inf; m[2::3, 9] = -3e38
pats["mixed"] = m
for kind in kinds:
    if kind == "flat":
        g = ca.FlatIndex(ctx, d, ca.COSINE)
    elif kind == "ivf":
        g = ca.IVFIndex(ctx, d, 32, ca.L2_SQUARED); g.train(X[:4000])
    elif kind == "pq":
        g = ca.PQIndex(ctx, d, ca.L2_SQUARED, 8, 6); g.train(X[:4000])
    elif kind == "ivfpq":
        g = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, 32, 8, 6); g.train(X[:4000])
    else:
        g = ca.HNSWIndex(ctx, d, ca.L2_SQUARED, 8, 40, 24); n = 3000
    g.add_batch(ids[:n], X[:n])
    kw = {"nprobes": 4} if kind in ("ivf", "ivfpq") else {}
    for name, Q in pats.items():
        print(kind, name, "...", flush=True)
        try:
            i, s, c = g.search_batch(Q, k, **kw)
            print(kind, name, "counts", c[:6].tolist(), flush=True)
        except Exception as e:          # noqa: BLE001
            print(kind, name, "raised", type(e).__name__, str(e)[:120], flush=True)
        qd = ctx.alloc(B * d * 4); ctx.upload(qd, Q)
        outs = [(ctx.alloc(B * k * 4), ctx.alloc(B * k * 4), ctx.alloc(B * 4)) for _ in range(4)]
        ts = [g.search_batch_dev_async(qd, B, k, *o, k, **kw) for o in outs]
        for t in ts:
            g.search_wait(t)
        ctx.sync()
        print(kind, name, "async ok", flush=True)
print("done")
