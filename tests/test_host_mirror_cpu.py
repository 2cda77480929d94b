"""CPU-side tests of the host mirror's own logic (no GPU, no HIP calls): the pieces that stay above the C ABI in the
reference too — sanitizeK / LimitResults / Autocut (limiter.go), multi-query aggregation (aggregation.go), argument
validation of the fluent builders — checked against the reference's known-answer tests and against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd.index import (MAX_AGGREGATION, MEAN_AGGREGATION, SUM_AGGREGATION, TextResult, VectorResult, aggregate,
                             aggregate_text, autocut, default_nprobes, sanitize_k, _metric_code, UnknownDistanceKind, _validate_pq)

KATS = json.loads((Path(__file__).parent / "golden" / "reference_kats.json").read_text())


def test_sanitize_k_matches_reference_and_oracle():
    # flat_index_search_test.go:348-389: k = 0 / -1 / 3 / 5 / 100 / 1 over 5 results
    for c in KATS["flat_search_k_bounds"]["cases"]:
        assert sanitize_k(c["k"], 5) == c["len"] == orc.lib().orc_sanitize_k(c["k"], 5)


def test_autocut_kats_and_oracle():
    for c in KATS["autocut"]["cases"]:
        assert autocut(c["scores"], c["cutoff"]) == c["expected"], c
    rng = np.random.default_rng(0)
    for _ in range(200):
        y = np.sort(rng.random(int(rng.integers(2, 30)))).astype(np.float32)
        cut = int(rng.integers(1, 4))
        assert autocut(y, cut) == orc.autocut(y, cut)


def test_vector_aggregation_kats():
    b = KATS["sum_aggregation"]
    res = [VectorResult(i, np.float32(s)) for i, s in zip(b["ids"], b["scores"])]
    agg = aggregate(res, SUM_AGGREGATION)
    assert len(agg) == b["unique"]
    assert {r.id: r.score for r in agg}[1] == np.float32(b["node1_expected"])          # the reference compares with ==
    assert all(agg[i].score <= agg[i + 1].score for i in range(len(agg) - 1))
    # max / mean (aggregation_test.go:54-115)
    res = [VectorResult(1, np.float32(0.1)), VectorResult(2, np.float32(0.2)), VectorResult(1, np.float32(0.3))]
    assert {r.id: r.score for r in aggregate(res, MAX_AGGREGATION)} == {1: np.float32(0.3), 2: np.float32(0.2)}
    assert {r.id: r.score for r in aggregate(res, MEAN_AGGREGATION)}[1] == np.float32((np.float32(0.1) + np.float32(0.3)) / np.float32(2))
    with pytest.raises(ValueError):
        aggregate(res, "median")
    assert aggregate([], SUM_AGGREGATION) == []


def test_aggregation_matches_oracle_on_random_lists():
    import ctypes as C
    rng = np.random.default_rng(1)
    for kind_i, kind in enumerate((SUM_AGGREGATION, MAX_AGGREGATION, MEAN_AGGREGATION)):
        ids = rng.integers(1, 20, 60).astype(np.uint32)
        sc = rng.random(60).astype(np.float32)
        got = aggregate([VectorResult(int(i), s) for i, s in zip(ids, sc)], kind)
        oi, os_ = np.zeros(60, np.uint32), np.zeros(60, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        n = orc.lib().orc_aggregate(kind_i, p(ids), p(sc), 60, p(oi), p(os_))
        assert [r.id for r in got] == oi[:n].tolist()
        assert np.array_equal(np.array([r.score for r in got], np.float32).view(np.uint32), os_[:n].view(np.uint32))


def test_text_aggregation_is_descending():
    res = [TextResult(1, np.float32(2.0)), TextResult(2, np.float32(3.0)), TextResult(1, np.float32(0.5))]
    agg = aggregate_text(res, SUM_AGGREGATION)          # bm25_index_search_test.go:449-563: 2.5 / 3.0
    assert [(r.id, float(r.score)) for r in agg] == [(2, 3.0), (1, 2.5)]
    assert [(r.id, float(r.score)) for r in aggregate_text(res, MAX_AGGREGATION)] == [(2, 3.0), (1, 2.0)]


def test_constructor_validation_and_defaults():
    assert default_nprobes(4096) == 64 and default_nprobes(10) == 3            # int(sqrt(nlist)) ivf_index.go:406-413
    with pytest.raises(UnknownDistanceKind):
        _metric_code("manhattan")                                               # ErrUnknownDistanceKind distance.go:9
    for bad in ((0, 4, 8), (8, 0, 8), (10, 4, 8), (8, 4, 0), (8, 4, 17)):
        with pytest.raises(ValueError):
            _validate_pq(*bad)                                                  # pq_index.go:135-155
    _validate_pq(8, 4, 16)


def test_rrf_batch_equals_per_query_form():
    """the vectorised batch form of RRF + sort + cut returns what the per-query dict form (the restatement of fusion.go:174-243) returns, bit for bit"""
    import numpy as np
    from comet_amd.hybrid import reciprocal_rank_fusion, reciprocal_rank_fusion_batch
    rng = np.random.default_rng(4)
    B, k = 64, 10
    v_ids = np.stack([rng.choice(40, k, replace=False) for _ in range(B)]).astype(np.uint32) + 1
    t_ids = np.stack([rng.choice(40, k, replace=False) for _ in range(B)]).astype(np.uint32) + 1
    v_cnt = rng.integers(0, k + 1, B); t_cnt = rng.integers(0, k + 1, B)
    v_cnt[0] = 0; t_cnt[1] = 0; v_cnt[2] = t_cnt[2] = k
    ids, sc, cnt = reciprocal_rank_fusion_batch(v_ids, v_cnt, t_ids, t_cnt, k)
    for b in range(B):
        v = {int(i): float(r) for r, i in enumerate(v_ids[b, :v_cnt[b]])}          # any ascending scores: only the order matters
        t = {int(i): float(-r) for r, i in enumerate(t_ids[b, :t_cnt[b]])}
        f = reciprocal_rank_fusion(v, t)
        want = sorted(f.items(), key=lambda kv: -kv[1])[:k]
        assert cnt[b] == len(want)
        assert [int(x) for x in ids[b, :cnt[b]]] == [d for d, _ in want]
        assert np.array_equal(sc[b, :cnt[b]].view(np.uint64), np.array([s for _, s in want], np.float64).view(np.uint64))


def test_bench_compact_line_is_driver_sized():
    """The driver parses bench.py's LAST stdout line from a bounded tail: the compact record built from a full record of every leg (the committed
    profiles/r05_bench_legs_f.json: all default legs present) stays under 6 KB and keeps the contract's keys, `roofline` and `cpu_baseline`."""
    import importlib.util
    import json
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_for_test", root / "bench.py")
        bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    full = json.loads((root / "profiles" / "r05_bench_legs_f.json").read_text())
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 6 * 1024, len(line)
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["config"]["workload"] and rec["roofline"]["kernel"] and rec["roofline"]["frac"] > 0 and rec["cpu_baseline"]["value"] > 0
    for leg in ("c1", "flat_l2", "ivfpq", "ivfpq_uniform", "ivfpq10m", "hnsw_reference_graph_parity", "hnsw_navigable", "hybrid"):
        assert leg in rec["legs"], leg


def test_rrf_batch_with_an_empty_leg():
    """round-4 advisor: a leg with zero columns (k = 0 on that side) used to raise in argmax over an empty axis"""
    import numpy as np
    from comet_amd.hybrid import reciprocal_rank_fusion_batch
    v_ids = np.array([[5, 9, 2], [7, 1, 0]], np.uint32); v_cnt = np.array([3, 2])
    none = np.zeros((2, 0), np.uint32); zero = np.zeros(2, np.int64)
    ids, sc, cnt = reciprocal_rank_fusion_batch(v_ids, v_cnt, none, zero, 2)
    assert cnt.tolist() == [2, 2] and ids[0].tolist() == [5, 9] and ids[1].tolist() == [7, 1] and np.allclose(sc[0], [1 / 60, 1 / 61])
    ids, sc, cnt = reciprocal_rank_fusion_batch(none, zero, v_ids, v_cnt, 3)
    assert cnt.tolist() == [3, 2] and ids[0].tolist() == [5, 9, 2]
    ids, sc, cnt = reciprocal_rank_fusion_batch(none, zero, none, zero, 3)
    assert cnt.tolist() == [0, 0] and ids.shape == (2, 0)
