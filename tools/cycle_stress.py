#!/usr/bin/env python3
"""cycle_stress.py — create / fill / search / destroy small indexes in a tight loop (no oracle): finds what only breaks after thousands of index lifetimes or
searches on one context (leaked events, streams, pinned slots, handle tables). usage: cycle_stress.py [cycles] [kinds, e.g. pq,flat]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import comet_amd as ca  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["flat", "ivf", "pq", "ivfpq"]
trace = open(os.environ.get("CYCLE_TRACE", "/dev/null"), "w")
ctx = ca.Context(0)
rng = np.random.default_rng(5)
d, n = 16, 2000
X = rng.standard_normal((n, d)).astype(np.float32)
Q = rng.standard_normal((8, d)).astype(np.float32)
ids = np.arange(1, n + 1, dtype=np.uint32)
flt = [int(i) for i in ids[::3]]
for i in range(N):
    kind = KINDS[i % len(KINDS)]
    trace.seek(0); trace.write(f"cycle {i} {kind}            \n"); trace.flush()
    if kind == "flat":
        g = ca.FlatIndex(ctx, d, ca.L2_SQUARED)
    elif kind == "ivf":
        g = ca.IVFIndex(ctx, d, 8, ca.L2_SQUARED); g.train(X[:500])
    elif kind == "pq":
        g = ca.PQIndex(ctx, d, ca.L2_SQUARED, 4, 6); g.train(X[:500])
    else:
        g = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, 4, 4, 8); g.train(X[:1200])
    g.add_batch(ids, X)
    kw = {"nprobes": 2} if kind in ("ivf", "ivfpq") else {}
    g.search_batch(Q, 5, **kw)
    g.search_batch(Q, 65, document_ids=flt, **kw)
    g.remove(7)
    g.search_batch(Q, 5, threshold=3.0, **kw)
    g.close()
    if i % 2000 == 0:
        print("cycle", i, flush=True)
print(f"cycle_stress OK: {N} index lifetimes ({KINDS}), 3 searches each")
