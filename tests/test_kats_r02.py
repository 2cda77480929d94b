"""The rest of SURVEY §8(c)'s known-answer list (tests/golden/reference_kats_r02.json, transcribed from the reference's
*_test.go): asserted on the CPU oracle and the host mirror here, and on the GPU path in the `gpu`-marked tests below."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd.hybrid import reciprocal_rank_fusion
from comet_amd.index import TextResult, VectorResult, aggregate, aggregate_text, sanitize_k

K2 = json.loads((Path(__file__).parent / "golden" / "reference_kats_r02.json").read_text())
CENTROID_BLOCKS = ["nearest_centroid_basic", "nearest_centroid_single", "nearest_centroid_negative"]


def test_sanitize_k_table():
    for c in K2["sanitize_k"]["cases"]:
        assert sanitize_k(c["k"], c["max"]) == c["want"] == orc.lib().orc_sanitize_k(c["k"], c["max"]), c


def test_vector_aggregation_sum_max_mean():
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for c in K2["vector_aggregation"]["cases"]:
        ids = np.asarray([r[0] for r in c["results"]], np.uint32); sc = np.asarray([r[1] for r in c["results"]], np.float32)
        oi, os_ = np.zeros(len(ids), np.uint32), np.zeros(len(ids), np.float32)
        n = orc.lib().orc_aggregate({"sum": 0, "max": 1, "mean": 2}[c["kind"]], p(ids), p(sc), len(ids), p(oi), p(os_))
        assert n == c["unique"]
        got = dict(zip(oi[:n].tolist(), os_[:n].tolist()))
        assert np.float32(got[c["node"]]) == np.float32(c["expected_float32"]), c          # the reference compares with ==
        assert all(os_[i] <= os_[i + 1] for i in range(n - 1))
        mirror = aggregate([VectorResult(int(i), np.float32(s)) for i, s in zip(ids, sc)], c["kind"])
        assert len(mirror) == c["unique"] and {r.id: r.score for r in mirror}[c["node"]] == np.float32(c["expected_float32"])


def test_text_aggregation_sum_max_mean():
    for c in K2["text_aggregation"]["cases"]:
        got = aggregate_text([TextResult(i, np.float32(s)) for i, s in c["results"]], c["kind"])
        assert {str(r.id): float(r.score) for r in got} == c["expected"], c
        assert all(got[i].score >= got[i + 1].score for i in range(len(got) - 1))


def _centroid_cases():
    for name in CENTROID_BLOCKS:
        b = K2[name]
        for v, want in b["cases"]:
            yield b["metric"], b["centroids"], v, [want]
    b = K2["nearest_centroid_metrics"]
    for m in b["metrics"]:
        for v, want in b["cases"]:
            yield m, b["centroids"], v, [want]
    b = K2["nearest_centroid_high_dim"]
    cen = [[float(i * b["scale"])] * b["dim"] for i in range(b["n"])]
    for t in range(b["n"]):
        yield "l2_squared", cen, [np.float32(t * b["scale"]) + np.float32(b["offset"])] * b["dim"], [t]
    b = K2["nearest_centroid_many"]
    cen = [[float(i)] * b["dim"] for i in range(b["n"])]
    for t in b["targets"]:
        yield "l2_squared", cen, [np.float32(t) + np.float32(b["offset"])] * b["dim"], [t]
    b = K2["nearest_centroid_boundary"]
    yield b["metric"], b["centroids"], b["vector"], [b["source_picks"]]
    b = K2["nearest_centroid_normalized"]
    for v, accepted in b["cases"]:
        yield b["metric"], b["centroids"], orc.preprocess("cosine", v)[0], accepted


def test_nearest_centroid_tables_oracle():
    for metric, cen, v, accepted in _centroid_cases():
        assert orc.nearest_centroid(v, cen, metric) in accepted, (metric, v)
    b = K2["nearest_centroid_consistency"]
    cent, assign = orc.kmeans(b["vectors"], b["k"], b["metric"], 20)
    for v, a in zip(b["vectors"], assign):
        assert orc.nearest_centroid(v, cent, b["metric"]) == a


def test_rrf_custom_k():
    b = K2["rrf_custom_k"]
    got = reciprocal_rank_fusion({int(k): v for k, v in b["vector"].items()}, {}, b["K"])
    assert abs(got[1] - b["expected"]["1"]) <= b["tolerance"]
    vid, vs = np.asarray([1], np.uint32), np.asarray([0.1], np.float64)
    oi, os_ = np.zeros(2, np.uint32), np.zeros(2, np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert orc.lib().orc_rrf(C.c_double(b["K"]), p(vid), p(vs), 1, None, None, 0, p(oi), p(os_)) == 1 and abs(os_[0] - 0.1) <= b["tolerance"]


def test_ivf_multiple_clusters_oracle():
    b = K2["ivf_multiple_clusters"]
    o = orc.IVF(b["dim"], b["metric"], b["nlist"])
    assert o.train(np.asarray(b["training"], np.float32)) == 0
    for i, v in enumerate(b["vectors"]):
        assert o.add(i + 1, v) == 0
    assert sum(1 for s in o.list_sizes() if s > 0) >= b["min_non_empty_lists"]


def _hnsw_recall_data(b):
    X = np.array([[(i * 10 + j) % 100 for j in range(b["dim"])] for i in range(b["n"])], np.float32)
    q = np.array([j % 100 for j in range(b["dim"])], np.float32)
    return X, q


def test_hnsw_recall_shape_oracle():
    b = K2["hnsw_recall_shape"]
    X, q = _hnsw_recall_data(b)
    o = orc.HNSW(b["dim"], b["metric"], b["m"], b["efc"], b["efs"], seed=11)
    assert o.add_batch(np.arange(1, b["n"] + 1), X) == 0
    n, ids, sc = o.search(q, b["k"])
    assert n == b["expected_results"] and all(s <= b["max_distance"] for s in sc)


def _bm25_docs(b):
    docs = {}
    for did, spec in b["docs_as_term_counts"].items():
        toks = [1] * spec["cat"] + list(range(100 + int(did) * 50, 100 + int(did) * 50 + spec["len"] - spec["cat"]))
        docs[int(did)] = toks
    return docs


def _check_bm25_properties(b, ids, scores):
    assert b["must_not_contain"] not in ids
    assert b["doc_in_top2"] in ids[:2]
    assert all(scores[i] >= scores[i + 1] for i in range(len(scores) - 1)) and all(s > 0 for s in scores)


def test_bm25_ranking_properties_oracle():
    b = K2["bm25_ranking_properties"]
    o = orc.BM25()
    for did, toks in _bm25_docs(b).items():
        assert o.add(did, toks) == 0
    n, ids, sc, _ = o.search([1], b["k"])
    _check_bm25_properties(b, ids.tolist(), sc.tolist())


# ---------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_nearest_centroid_tables_gpu(ctx):
    for metric, cen, v, accepted in _centroid_cases():
        got = int(ctx.nearest_centroid(np.asarray(v, np.float32), np.asarray(cen, np.float32), metric)[0])
        assert got in accepted and got == orc.nearest_centroid(v, cen, metric), (metric, v)
    b = K2["nearest_centroid_consistency"]
    cent, assign = ctx.kmeans(np.asarray(b["vectors"], np.float32), b["k"], b["metric"], 20)
    assert ctx.nearest_centroid(np.asarray(b["vectors"], np.float32), cent, b["metric"]).tolist() == assign.tolist()


@pytest.mark.gpu
def test_ivf_multiple_clusters_gpu(ctx):
    from comet_amd import IVFIndex
    b = K2["ivf_multiple_clusters"]
    g = IVFIndex(ctx, b["dim"], b["nlist"], b["metric"])                  # NewIVFIndex(3, 3, Euclidean)
    g.train(np.asarray(b["training"], np.float32))
    for i, v in enumerate(b["vectors"]):
        g.add(i + 1, np.asarray(v, np.float32))
    assert sum(1 for l in range(b["nlist"]) if g.list_size(l) > 0) >= b["min_non_empty_lists"]
    o = orc.IVF(b["dim"], b["metric"], b["nlist"]); o.train(np.asarray(b["training"], np.float32))
    for i, v in enumerate(b["vectors"]):
        o.add(i + 1, v)
    assert [g.list_size(l) for l in range(b["nlist"])] == o.list_sizes()


@pytest.mark.gpu
def test_hnsw_recall_shape_gpu(ctx):
    from comet_amd import HNSWIndex
    b = K2["hnsw_recall_shape"]
    X, q = _hnsw_recall_data(b)
    o = orc.HNSW(b["dim"], b["metric"], b["m"], b["efc"], b["efs"], seed=11)
    assert o.add_batch(np.arange(1, b["n"] + 1), X) == 0
    g = HNSWIndex(ctx, b["dim"], b["metric"], b["m"], b["efc"], b["efs"])
    ids, levels, vecs, eoff, edges = o.export()
    g.load_graph(ids, levels, vecs, eoff, edges, o.entry(), o.max_level())
    res = g.new_search().with_query(q).with_k(b["k"]).execute()
    assert len(res) == b["expected_results"] and all(r.score <= b["max_distance"] for r in res)
    assert [r.id for r in res] == o.search(q, b["k"])[1].tolist()


@pytest.mark.gpu
def test_bm25_ranking_properties_gpu(ctx):
    from comet_amd import BM25SearchIndex
    b = K2["bm25_ranking_properties"]
    g = BM25SearchIndex(ctx)
    for did, toks in _bm25_docs(b).items():
        g.add(did, toks)
    res = g.new_search().with_query([1]).with_k(b["k"]).execute()
    _check_bm25_properties(b, [r.id for r in res], [float(r.score) for r in res])
