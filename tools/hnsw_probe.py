import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, comet_amd as ca, oracle_lib as orc
ctx = ca.Context(0)
n, d, B = 100000, 384, 256
X = orc.synth(0x48, 0, n * d).reshape(n, d)
o = orc.HNSW(d, "l2", 16, 200, 128, seed=7); o.add_batch(np.arange(1, n + 1), X)
ids, levels, vecs, eoff, edges = o.export()
g = ca.HNSWIndex(ctx, d, ca.EUCLIDEAN, 16, 200, 128); g.load_graph(ids, levels, vecs, eoff, edges, o.entry(), o.max_level())
Q = orc.synth(0x49, 0, B * d).reshape(B, d)
g.search_batch(Q, 10, ef_search=128)
ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(10): g.search_batch(Q, 10, ef_search=128)
el = (time.perf_counter() - t0) / 10
print("ms/batch %.3f" % (el * 1e3), {k: round(v[0] / 10, 4) for k, v in ctx.profile_dump().items()})
