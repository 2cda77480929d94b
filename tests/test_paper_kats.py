"""Hand-computed known-answer cases (tests/golden/paper_kats.json: every number follows from the reference's source by arithmetic written out in the file) for the
parts of the path the reference's own tests hold no numbers for — PQ / IVFPQ training, codes, distances and tie order, BM25 scores, the shape of an HNSW graph
(nine nodes, every searchLayer / prune step written out; the GPU insert kernel's graphs are compared with the oracle's byte for byte in tests/test_hnsw_gpu.py). Asserted on the CPU oracle here
(no GPU) and on the GPU through the C ABI (-m gpu): with these the oracle is pinned for PQ / IVFPQ / BM25 by something other than itself.
Reference: clustering.go:119-243, pq_index.go:189-260,439-471, pq_index_search.go:243-306, ivfpq_index.go:176-260,467-500, ivfpq_index_search.go:231-390,
bm25_index_search.go:278-397, hnsw_index.go:228-288,493-694, hnsw_index_search.go:248-354."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc

K = json.loads((Path(__file__).parent / "golden" / "paper_kats.json").read_text())


def root32(s):
    return np.float32(np.sqrt(np.float64(s)))                      # float32(math.Sqrt(float64(sum)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def pq_case():
    c = K["pq_dim4_m2_nbits1"]
    ids = np.array(sorted(int(i) for i in c["vectors"]), np.uint32)
    X = np.array([c["vectors"][str(i)] for i in ids], np.float32)
    return c, np.array(c["train"], np.float32), ids, X


def ivfpq_case():
    c = K["ivfpq_dim4_nlist2_m2_nbits1"]
    t = c["train"]
    T = np.array([t[name] for name in t["order"] for _ in range(t["copies_each"])], np.float32)
    ids = np.array(sorted(int(i) for i in c["vectors"]), np.uint32)
    X = np.array([c["vectors"][str(i)] for i in ids], np.float32)
    return c, T, ids, X


def check_pq(codebooks, c, ids, search):
    assert np.array_equal(np.asarray(codebooks, np.float32).reshape(2, 2, 2), np.array(c["codebooks"], np.float32))
    for q in c["queries"]:
        n, gi, gs = search(np.array(q["q"], np.float32), q["k"])
        assert n == len(q["ids"]) and list(gi[:n]) == q["ids"], (q, gi[:n])
        assert np.array_equal(bits(gs[:n]), bits(np.array([root32(s) for s in q["sums"]]))), (q, gs[:n])


def test_pq_paper_case_on_the_oracle():
    c, T, ids, X = pq_case()
    o = orc.PQ(c["dim"], "l2_squared", c["M"], c["nbits"])
    assert o.train(T) == 0 and o.add_batch(ids, X) == 0
    assert np.array_equal(o.codes(), np.array([c["codes"][str(i)] for i in ids], np.uint8))
    check_pq(o.codebooks(), c, ids, lambda q, k: o.search(q, k))


def test_ivfpq_paper_case_on_the_oracle():
    c, T, ids, X = ivfpq_case()
    o = orc.IVFPQ(c["dim"], "l2_squared", c["nlist"], c["M"], c["nbits"])
    assert o.train(T) == 0 and o.add_batch(ids, X) == 0
    assert np.array_equal(np.asarray(o.centroids(), np.float32).reshape(2, 4), np.array(c["centroids"], np.float32))
    assert np.array_equal(np.asarray(o.codebooks(), np.float32).reshape(2, 2, 2), np.array(c["codebooks"], np.float32))
    for l in ("0", "1"):
        assert list(o.list_ids(int(l))) == c["lists"][l]["ids"] and np.array_equal(o.list_codes(int(l)), np.array(c["lists"][l]["codes"], np.uint8)), l
    for q in c["queries"]:
        n, gi, gs = o.search(np.array(q["q"], np.float32), q["k"], q["nprobes"])
        assert n == len(q["ids"]) and list(gi[:n]) == q["ids"], (q, gi[:n])
        assert np.array_equal(bits(gs[:n]), bits(np.array([root32(s) for s in q["sums"]]))), (q, gs[:n])


def bm25_expected(c, q):
    """the file's formula, evaluated in float64 in the reference's order (bm25_index_search.go:306-325)"""
    docs = c["documents"]
    N = float(len(docs)); avg = sum(len(t) for t in docs.values()) / len(docs)
    idf = {"ln1.6": math.log((N - 2 + 0.5) / (2 + 0.5) + 1.0), "ln8/3": math.log((N - 1 + 0.5) / (1 + 0.5) + 1.0)}
    out = []
    for d in q["ids"]:
        s = 0.0
        for name, tf, ln in q["terms"][str(d)]:
            tfv, dl = float(tf), float(ln)
            s += idf[name] * (tfv * (1.2 + 1)) / (tfv + 1.2 * (1 - 0.75 + 0.75 * (dl / avg)))
        out.append(s)
    return out


def check_bm25(c, search):
    for q in c["queries"]:
        n, gi, gs64 = search(np.array(q["tokens"], np.uint32), q["k"])
        assert n == len(q["ids"]) and list(gi[:n]) == q["ids"], (q, gi[:n])
        want = bm25_expected(c, q)
        for a, b, approx in zip(gs64[:n], want, q["approx"]):
            assert abs(a - b) <= 4.5e-16 * abs(b), (a, b)              # 2 ulp: Go's math.Log vs libm
            assert abs(a - approx) < 2e-6, (a, approx)                 # and the digits written in the file


def test_bm25_paper_case_on_the_oracle():
    c = K["bm25_three_documents"]
    o = orc.BM25()
    for d, toks in c["documents"].items():
        o.add(int(d), np.array(toks, np.uint32))

    def search(q, k):
        n, ids, _s32, s64 = o.search(q, k)
        return n, ids, s64
    check_bm25(c, search)


def hnsw_case(metric):
    c = K["hnsw_dim2_m2_efc3"]
    o = orc.HNSW(c["dim"], metric, c["M"], c["efConstruction"], c["efSearch"])
    for i in c["insert_order"]:
        assert o.add(i, np.array(c["vectors"][str(i)], np.float32), c["levels"][str(i)]) == 0
    return c, o


@pytest.mark.parametrize("metric", ["l2_squared", "l2"])
def test_hnsw_paper_case_on_the_oracle(metric):
    """insertNode / searchLayer / selectNeighbors / pruneConnections and the search, on nine points whose every step is written out in the fixture:
    edge lists with their order, entry point, maxLevel, the node no search can reach, result ids and distances. Euclidean takes the root of the same
    (distinct) squared distances: the same graph, scores float32(math.Sqrt(float64(d2))) (distance.go:120)."""
    c, o = hnsw_case(metric)
    ids, levels, vecs, eoff, edges = o.export()
    assert list(ids) == c["insert_order"] and [int(x) for x in levels] == [c["levels"][str(i)] for i in c["insert_order"]]
    assert o.entry() == c["entry"] and o.max_level() == c["max_level"]
    slot = 0
    for i, lv in zip(ids, levels):
        for layer in range(int(lv) + 1):
            got = [int(x) for x in edges[eoff[slot]:eoff[slot + 1]]]
            assert got == c["graph"][str(int(i))][layer], (int(i), layer, got)
            slot += 1
    inbound = {int(x) for x in edges}
    assert 7 not in inbound and inbound == {1, 2, 3, 4, 5, 6, 8, 9}            # the prune quirk (hnsw_index.go:283-284,673-676): node 7 is linked from nowhere
    for q in c["queries"]:
        n, gi, gs = o.search(np.array(q["q"], np.float32), q["k"], q["ef"])
        want = np.array(q["d2"], np.float32) if metric == "l2_squared" else np.array([root32(x) for x in q["d2"]], np.float32)
        assert n == len(q["ids"]) and [int(x) for x in gi[:n]] == q["ids"], (q, gi[:n])
        assert np.array_equal(bits(gs[:n]), bits(want)), (q, gs[:n])


def test_hnsw_paper_case_trace_printer_agrees():
    """tools/hnsw_paper_sim.py — the plain-Python restatement the fixture's trace was checked with (no code shared with the oracle) — still gives the fixture's graph and results"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hnsw_paper_sim", Path(__file__).resolve().parent.parent / "tools" / "hnsw_paper_sim.py")
    sim = importlib.util.module_from_spec(spec); spec.loader.exec_module(sim)
    c = K["hnsw_dim2_m2_efc3"]
    g = sim.build_case({k: c[k] for k in ("M", "efConstruction", "efSearch", "insert_order", "vectors", "levels")})
    assert {str(i): n["e"] for i, n in g.nodes.items()} == c["graph"] and g.entry == c["entry"] and g.maxLevel == c["max_level"]
    for q in c["queries"]:
        r = g.search(list(q["q"]), q["k"], q["ef"])
        assert [b for _a, b in r] == q["ids"] and [a for a, _b in r] == q["d2"], q


# ---------------------------------------------------------------------------------------------- the same cases on the GPU, through the C ABI
@pytest.mark.gpu
def test_pq_paper_case_on_the_gpu(ctx):
    from comet_amd import L2_SQUARED, PQIndex
    c, T, ids, X = pq_case()
    g = PQIndex(ctx, c["dim"], L2_SQUARED, c["M"], c["nbits"])
    g.train(T); g.add_batch(ids, X)
    gi, gcodes, _ = g.list_read(0, codes_width=c["M"])
    assert np.array_equal(gi, ids) and np.array_equal(gcodes, np.array([c["codes"][str(i)] for i in ids], np.uint8))

    def search(q, k):
        i, s, n = g.search_batch(q[None, :], k)
        return int(n[0]), i[0], s[0]
    check_pq(g.codebooks(c["M"], 2, 2), c, ids, search)


@pytest.mark.gpu
def test_ivfpq_paper_case_on_the_gpu(ctx):
    from comet_amd import IVFPQIndex, L2_SQUARED
    c, T, ids, X = ivfpq_case()
    g = IVFPQIndex(ctx, c["dim"], L2_SQUARED, c["nlist"], c["M"], c["nbits"])
    g.train(T); g.add_batch(ids, X)
    assert np.array_equal(np.asarray(g.centroids(2), np.float32).reshape(2, 4), np.array(c["centroids"], np.float32))
    assert np.array_equal(np.asarray(g.codebooks(2, 2, 2), np.float32).reshape(2, 2, 2), np.array(c["codebooks"], np.float32))
    for l in ("0", "1"):
        gi, gc, _ = g.list_read(int(l), codes_width=c["M"])
        assert list(gi) == c["lists"][l]["ids"] and np.array_equal(gc, np.array(c["lists"][l]["codes"], np.uint8)), l
    for q in c["queries"]:
        for mode in (0, 1):
            i, s, n = g.search_batch(np.array([q["q"]], np.float32), q["k"], nprobes=q["nprobes"], mode=mode)
            nn = int(n[0])
            assert nn == len(q["ids"]) and list(i[0, :nn]) == q["ids"], (q, mode, i[0, :nn])
            assert np.array_equal(bits(s[0, :nn]), bits(np.array([root32(x) for x in q["sums"]]))), (q, mode, s[0, :nn])


@pytest.mark.gpu
def test_bm25_paper_case_on_the_gpu(ctx):
    from comet_amd import BM25SearchIndex
    c = K["bm25_three_documents"]
    g = BM25SearchIndex(ctx)
    for d, toks in c["documents"].items():
        g.add(int(d), np.array(toks, np.uint32))

    def search(q, k):
        ids, _sc, sc64, cnt = g.search_batch([q], k)
        return int(cnt[0]), ids[0], sc64[0]
    check_bm25(c, search)
