# tools/hnsw_graphs_probe.py [rows] — hnsw_search on two graphs over the same clustered rows: one built by the GPU insert kernel (the reference's
# construction: searches stop after ~33 expansions) and a navigable layer-0 graph (16 exact neighbours + 16 random edges per node, ~150 expansions).
# Prints the kernel's time per batch of 256 and the counters; with a library built with -DHN_TRACE (index_hnsw.hip) the per-expansion phase trace too.
# HN_RESULTS=<tag> saves the results to /tmp/hn_<graph>_<tag>.npy; a later run with another tag compares against every saved tag (A/B of two builds).
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, comet_amd as ca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d, B, K = 384, 256, 10
tag = os.environ.get("HN_RESULTS", "")
ctx = ca.Context(0)
rng = np.random.default_rng(5)
cent = rng.standard_normal((max(64, n // 15), d), dtype=np.float32)
X = (cent[rng.integers(0, len(cent), n)] + 0.15 * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
Q = (cent[rng.integers(0, len(cent), B)] + 0.15 * rng.standard_normal((B, d), dtype=np.float32)).astype(np.float32)
ids_n = np.arange(1, n + 1, dtype=np.uint32)
def run(g, name):
    ids, sc, cnt = g.search_batch(Q, K, ef_search=128)
    print(name, "evals/q", g.stat("hnsw_distance_evals") / B, "exp/q", g.stat("hnsw_expansions") / B)
    mine = np.concatenate([ids.astype(np.int64).ravel(), sc.view(np.uint32).astype(np.int64).ravel(), cnt.astype(np.int64).ravel()])
    if tag: np.save("/tmp/hn_%s_%s.npy" % (name, tag), mine)
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(10): g.search_batch(Q, K, ef_search=128)
    el = (time.perf_counter() - t0) / 10
    print("  ms/batch %.3f" % (el * 1e3), {k: round(v[0] / 10, 4) for k, v in ctx.profile_dump().items() if "hnsw" in k})
    ctx.profile(False)
    import glob
    for f in sorted(glob.glob("/tmp/hn_%s_*.npy" % name)):
        if tag and f.endswith("_%s.npy" % tag): continue
        o = np.load(f); print("  identical to %s:" % f, bool(o.shape == mine.shape and (o == mine).all()))
g = ca.HNSWIndex(ctx, d, ca.EUCLIDEAN, 16, 200, 128); g.set_level_seed(7)
t0 = time.time(); g.add_batch(ids_n, X); print("build %.1fs" % (time.time() - t0))
run(g, "built")
fl = ca.FlatIndex(ctx, d, ca.EUCLIDEAN); fl.add_batch(ids_n, X)
knn = np.zeros((n, 17), np.uint32)
for lo in range(0, n, 4096): knn[lo:lo + 4096] = fl.search_batch(X[lo:lo + 4096], 17)[0]
fl.close()
edges = np.concatenate([knn[:, 1:17], np.random.default_rng(5).integers(1, n + 1, (n, 16), dtype=np.uint32)], axis=1)
gn = ca.HNSWIndex(ctx, d, ca.EUCLIDEAN, 16, 200, 128)
gn.load_graph(ids_n, np.zeros(n, np.int32), X, np.arange(0, (n + 1) * 32, 32, dtype=np.int64), edges.reshape(-1), 1, 0)
run(gn, "navigable")
