"""The reference's own document-filter tests (WithDocumentIDs — document_filter.go; flat / ivf / pq / ivfpq / hnsw / bm25 _index_document_filter_test.go), restated on the
CPU oracle from the hand-transcribed tables in tests/golden/reference_kats_r06.json: the same indexes, rows, queries, filters, thresholds, nprobes / efSearch values and
expected id sets. No GPU: this pins the oracle's pre-filter (ids skipped before the distance, flat_index_search.go:254-262 and the same lines of the other kinds) by the
reference's tests; the GPU's filter path is compared with the oracle's bit for bit in the -m gpu suites (test_flat_gpu.py, test_configs_gpu.py, test_fuzz_gpu.py …)."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc

K = json.loads((Path(__file__).parent / "golden" / "reference_kats_r06.json").read_text())


def ramp(i):
    v = np.zeros(3, np.float32); v[i % 3] = np.float32(i)                   # the tests' `vec[i%3] = float32(i)`
    return v


def slope():
    return np.array([[i + j for j in range(8)] for i in range(20)], np.float32)          # `vec[j] = float32(i + j)`, i = 0..19


def rows_of(c):
    if "vectors" in c:
        ids = sorted(int(i) for i in c["vectors"])
        return ids, np.array([c["vectors"][str(i)] for i in ids], np.float32)
    lo, hi = c["ramp_ids"]
    ids = list(range(lo, hi + 1))
    return ids, np.array([ramp(i) for i in ids], np.float32)


def check_ids(c, got, want, exact):
    got = [int(x) for x in got]
    assert len(set(got)) == len(got), (c["source"], got)
    if exact:
        assert sorted(got) == want, (c["source"], got, want)
    else:
        assert set(got) <= set(want), (c["source"], got, want)
        if not want:
            assert not got, (c["source"], got)


def cases_of(c):
    return c["cases"] if "cases" in c else [{"name": c["source"], "filter": c["filter"], "ids": c["ids"]}]


def test_flat_document_filter_tables():
    for key in ("flat", "flat_multi_query", "flat_threshold"):
        c = K[key]
        ids, X = rows_of(c)
        o = orc.Flat(c["dim"], c["metric"]); assert o.add_batch(np.array(ids, np.uint32), X) == 0
        for q in c.get("queries", [c.get("query")]):
            for case in cases_of(c):
                n, gi, gs = o.search(np.array(q, np.float32), c["k"], threshold=c.get("threshold", 0.0), filter_ids=case["filter"])
                check_ids(c, gi[:n], case["ids"], c["exact"])
                assert np.all(np.diff(gs[:n]) >= 0)
    # the threshold case's distances are the ones its comments name: 0, 1 (2 and 9 are cut by the filter / the threshold)
    c = K["flat_threshold"]; ids, X = rows_of(c)
    o = orc.Flat(3, "l2"); o.add_batch(np.array(ids, np.uint32), X)
    n, gi, gs = o.search(np.array(c["query"], np.float32), 10, threshold=1.5, filter_ids=[1, 2, 3])
    assert list(gi[:n]) == [1, 2] and list(gs[:n]) == [0.0, 1.0]


def test_ivf_document_filter_tables():
    c = K["ivf"]; ids, X = rows_of(c)
    o = orc.IVF(c["dim"], c["metric"], c["nlist"]); assert o.train(np.array(c["train"], np.float32)) == 0 and o.add_batch(np.array(ids, np.uint32), X) == 0
    for case in c["cases"]:
        n, gi, _ = o.search(np.array(c["query"], np.float32), c["k"], c["nprobes"], filter_ids=case["filter"])
        check_ids(c, gi[:n], case["ids"], False)
        if not case["filter"]:
            assert sorted(int(x) for x in gi[:n]) == case["ids"]                       # every list probed, nothing filtered: all six rows come back
    c = K["ivf_nprobes"]; ids, X = rows_of(c)
    lo, hi = c["train_ramp"]
    o = orc.IVF(c["dim"], c["metric"], c["nlist"]); assert o.train(np.array([ramp(i) for i in range(lo, hi + 1)], np.float32)) == 0 and o.add_batch(np.array(ids, np.uint32), X) == 0
    seen = []
    for npb in c["nprobes"]:
        n, gi, _ = o.search(np.array(c["query"], np.float32), c["k"], npb, filter_ids=c["filter"])
        check_ids(c, gi[:n], c["ids"], False); seen.append(set(int(x) for x in gi[:n]))
    assert seen[0] <= seen[1] <= seen[2] == set(c["ids"])                              # more lists probed -> a superset; all four lists -> every filtered row


def test_pq_and_ivfpq_document_filter_tables():
    c = K["pq"]; ids, X = rows_of(c)
    o = orc.PQ(c["dim"], c["metric"], c["M"], c["nbits"]); assert o.train(slope()) == 0 and o.add_batch(np.array(ids, np.uint32), X) == 0
    for case in c["cases"]:
        n, gi, _ = o.search(np.array(c["query"], np.float32), c["k"], filter_ids=case["filter"])
        check_ids(c, gi[:n], case["ids"], False)
        assert n == len(case["ids"])                                                   # PQ scans every code: exactly the filtered rows
    c = K["ivfpq"]; ids, X = rows_of(c)
    o = orc.IVFPQ(c["dim"], c["metric"], c["nlist"], c["M"], c["nbits"]); assert o.train(slope()) == 0 and o.add_batch(np.array(ids, np.uint32), X) == 0
    for case in c["cases"]:
        n, gi, _ = o.search(np.array(c["query"], np.float32), c["k"], c["nprobes"], filter_ids=case["filter"])
        check_ids(c, gi[:n], case["ids"], False)
        assert n == len(case["ids"])                                                   # nprobes = nlist


@pytest.mark.parametrize("seed", [1, 12345, 987654321])
def test_hnsw_document_filter_tables(seed):
    """levels come from the oracle's seeded generator (the reference's from math/rand): with M 16 and at most twenty rows nothing is ever pruned, every node is
    linked both ways on layer 0 and reachable from the first node whatever the levels are — the expected id sets do not depend on them"""
    for key in ("hnsw", "hnsw_ef_search", "hnsw_after_deletion"):
        c = K[key]; ids, X = rows_of(c)
        o = orc.HNSW(c["dim"], c["metric"], c["M"], c["efConstruction"], c["efSearch"], seed=seed)
        assert o.add_batch(ids, X) == 0
        for r in c.get("remove", []):
            assert o.remove(r) == 0
        for case in cases_of(c):
            n, gi, gs = o.search(np.array(c["query"], np.float32), c["k"], c.get("ef", 0), filter_ids=case["filter"])
            check_ids(c, gi[:n], case["ids"], c["exact"])
            assert n <= c["k"] and np.all(np.diff(gs[:n]) >= 0)


def test_bm25_document_filter_table():
    c = K["bm25"]
    vocab = {}
    tok = lambda text: [vocab.setdefault(w, len(vocab) + 1) for w in text.split(" ")]
    o = orc.BM25()
    for d in sorted(c["documents"], key=int):
        o.add(int(d), tok(c["documents"][d]))
    for case in c["cases"]:
        n, gi, _s32, s64 = o.search(tok(case["query"]), 10, filter_ids=case["filter"])
        check_ids(c, gi[:n], case["ids"], True)
        assert np.all(np.diff(s64[:n]) <= 0) and np.all(s64[:n] > 0)


@pytest.mark.parametrize("seed", [1, 12345])
def test_hnsw_search_behaviours_of_the_reference_tests(seed):
    """hnsw_index_search_test.go:123-330 (Simple, ExactMatch, KGreaterThanSize, WithThreshold, ThresholdStrictFiltering), :646-852 (zero-vector cosine query, empty index,
    all rows deleted + Flush, single node, the three metrics, cosine nearest). The tests' NewVectorNode draws ids from a process-wide counter (node.go:55-61); here the rows go in with id 0
    and take the index's own ids (hnsw_index.go:259-262: nextID, from 0 — the first node's id 0 is also the 'no entry point' value of :268) — none of the expectations depends on the ids."""
    c = K["hnsw_search"]
    for case in c["cases"]:
        o = orc.HNSW(c["dim"], case["metric"], c["M"], c["efConstruction"], c["efSearch"], seed=seed)
        for v in case["vectors"]:
            assert o.add(0, np.array(v, np.float32)) == 0, case["name"]
        ids, _lv, vecs, _eo, _ed = o.export() if case["vectors"] else (np.zeros(0, np.uint32), None, np.zeros((0, 3), np.float32), None, None)
        assert list(ids) == list(range(len(case["vectors"]))), case["name"]            # auto ids 0, 1, 2, …
        if case.get("remove_all_and_flush"):
            for i in ids:
                assert o.remove(int(i)) == 0
            o.flush()
        n, gi, gs = o.search(np.array(case["query"], np.float32), case["k"], 0, threshold=case.get("threshold", 0.0), cap=16)
        if case.get("error"):
            assert n < 0, case["name"]                                                  # ErrZeroVector from Preprocess (distance.go: cosine)
            continue
        if "count" in case:
            assert n == case["count"], (case["name"], n)
        if "min_count" in case:
            assert n >= case["min_count"], (case["name"], n)
        byid = {int(i): v for i, v in zip(ids, vecs)}
        if "first" in case:
            assert np.allclose(byid[int(gi[0])], _unit(case["first"]) if case["metric"] == "cosine" else case["first"], atol=1e-3), (case["name"], gi[:n])
        if "max_distance" in case:
            for i in gi[:n]:
                assert np.sqrt(((byid[int(i)] - np.array(case["query"], np.float32)) ** 2).sum()) <= case["max_distance"], case["name"]
        assert np.all(np.diff(gs[:n]) >= 0)


def _unit(v):
    v = np.array(v, np.float32)
    return v / np.sqrt((v * v).sum())
