"""Index lifecycle behaviours of the reference's *_index_test.go files — training preconditions, Add before Train, zero vectors under cosine, soft delete / Flush
bookkeeping, searches that skip soft-deleted rows, HNSW's Flush of the entry point and of every node — restated on the CPU oracle (error CODES stand where the
reference returns an error value; the messages are the GPU library's and are compared in the -m gpu suites). Sources: flat_index_test.go:188-313,343-435;
ivf_index_test.go:97-132,204-378,716-818; pq_index_test.go:159-227,319-531; ivfpq_index_test.go:150-204,297-506; hnsw_index_test.go:443-672. No GPU."""
import numpy as np
import pytest

import oracle_lib as orc

E8 = [1, 0, 0, 0, 0, 0, 0, 0]


def slope8(n):
    return np.array([[(i * 8 + j) % 10 for j in range(8)] for i in range(n)], np.float32)


def ramp8(n):
    return np.array([[i * 8 + j for j in range(8)] for i in range(n)], np.float32)


def test_training_preconditions():
    # five rows for ten lists (ivf_index_test.go:97-117, ivfpq_index_test.go:150-170); 100 rows for Ksub 256 (pq_index_test.go:159-184)
    five3 = np.array([[i, 0, 0] for i in range(5)], np.float32)
    assert orc.IVF(3, "l2", 10).train(five3) == orc.ERR_TRAIN_DATA
    five8 = np.array([[i, 0, 0, 0, 0, 0, 0, 0] for i in range(5)], np.float32)
    assert orc.IVFPQ(8, "l2", 10, 4, 4).train(five8) == orc.ERR_TRAIN_DATA
    assert orc.PQ(8, "l2", 4, 8).train(np.array([[i] * 8 for i in range(100)], np.float32)) == orc.ERR_TRAIN_DATA
    # enough rows: 300 for Ksub 256 (pq_index_test.go:111-157), nlist * 10 (ivfpq_index_test.go:89-148)
    assert orc.PQ(8, "l2", 4, 8).train(slope8(300)) == 0
    assert orc.IVFPQ(8, "l2", 2, 4, 4).train(ramp8(100)) == 0
    # Add before Train (ivf_index_test.go:119-132, pq_index_test.go:214-227, ivfpq_index_test.go:191-204)
    assert orc.IVF(3, "l2", 2).add(1, [1, 0, 0]) == orc.ERR_NOT_TRAINED
    assert orc.PQ(8, "l2", 4, 6).add(1, E8) == orc.ERR_NOT_TRAINED
    assert orc.IVFPQ(8, "l2", 2, 4, 4).add(1, E8) == orc.ERR_NOT_TRAINED
    # and a search before Train (ivf_index_search_test.go:296-311, pq_index_search_test.go:338-354, ivfpq_index_search_test.go:577-594)
    assert orc.IVF(3, "l2", 2).search(np.array([1, 0, 0], np.float32), 5, 1, cap=4)[0] == orc.ERR_NOT_TRAINED
    assert orc.PQ(8, "l2", 4, 6).search(np.array(E8, np.float32), 5, cap=4)[0] == orc.ERR_NOT_TRAINED
    assert orc.IVFPQ(8, "l2", 2, 4, 4).search(np.array(E8, np.float32), 5, 1, cap=4)[0] == orc.ERR_NOT_TRAINED


def test_zero_vector_under_cosine():
    """Add of a zero vector fails for cosine and is fine for Euclidean (ivf_index_test.go:204-225, pq_index_test.go:319-348, ivfpq_index_test.go:297-327; flat: flat_index_test.go:88-159)"""
    f = orc.Flat(3, "cosine"); assert f.add(1, [0, 0, 0]) == orc.ERR_ZERO_VECTOR and f.add(2, [1, 0, 0]) == 0 and f.capacity() == 1
    assert orc.Flat(3, "l2").add(1, [0, 0, 0]) == 0
    i = orc.IVF(3, "cosine", 2); assert i.train(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0]], np.float32)) == 0
    assert i.add(1, [0, 0, 0]) == orc.ERR_ZERO_VECTOR and i.add(2, [0, 1, 0]) == 0
    p = orc.PQ(8, "cosine", 4, 6); assert p.train(slope8(100) + 1) == 0 and p.add(1, [0] * 8) == orc.ERR_ZERO_VECTOR and p.add(2, E8) == 0
    q = orc.IVFPQ(8, "cosine", 2, 4, 4); assert q.train(ramp8(100) + 1) == 0 and q.add(1, [0] * 8) == orc.ERR_ZERO_VECTOR and q.add(2, E8) == 0
    h = orc.HNSW(3, "cosine", 16, 200, 200); assert h.add(1, [0, 0, 0]) == orc.ERR_ZERO_VECTOR and h.add(2, [1, 2, 3]) == 0


def make(kind):
    if kind == "flat":
        return orc.Flat(8, "l2"), {}
    if kind == "ivf":
        o = orc.IVF(8, "l2", 2); assert o.train(ramp8(100)) == 0
        return o, {"nprobes": 2}
    if kind == "pq":
        o = orc.PQ(8, "l2", 4, 6); assert o.train(slope8(100)) == 0
        return o, {}
    if kind == "ivfpq":
        o = orc.IVFPQ(8, "l2", 2, 4, 4); assert o.train(ramp8(100)) == 0
        return o, {"nprobes": 2}
    return orc.HNSW(8, "l2", 16, 200, 200), {"ef": 0}


def stored(o):
    """rows in storage, soft-deleted ones included (`len(idx.vectors)` / the lists' lengths; the IVF wrappers' capacity() is a host-side add counter)"""
    return o._count() if hasattr(o, "_count") else o.capacity()


def search_ids(o, kw, q, k, **more):
    if "nprobes" in kw:
        n, gi, _ = o.search(np.array(q, np.float32), k, kw["nprobes"], cap=16, **more)
    elif "ef" in kw:
        n, gi, _ = o.search(np.array(q, np.float32), k, 0, cap=16, **more)
    else:
        n, gi, _ = o.search(np.array(q, np.float32), k, cap=16, **more)
    return [int(x) for x in gi[:n]]


@pytest.mark.parametrize("kind", ["flat", "ivf", "pq", "ivfpq", "hnsw"])
def test_remove_flush_and_soft_deleted_rows_in_searches(kind):
    """Remove soft-deletes (storage keeps the row), a second Remove and an unknown id fail, Flush drops the rows and empties the bitmap, searches skip deleted rows
    before and after Flush (flat_index_test.go:188-313,343-435 and the same tests of the other kinds)"""
    o, kw = make(kind)
    rows = {11: [1, 0, 0, 0, 0, 0, 0, 0], 12: [2, 0, 0, 0, 0, 0, 0, 0], 13: [3, 0, 0, 0, 0, 0, 0, 0], 14: [4, 0, 0, 0, 0, 0, 0, 0]}
    for i, v in rows.items():
        assert o.add(i, v) == 0
    q = [1.5, 0, 0, 0, 0, 0, 0, 0]
    assert sorted(search_ids(o, kw, q, 10)) == [11, 12, 13, 14]
    assert o.remove(12) == 0 and o.remove(13) == 0
    assert stored(o) == 4                                                   # soft delete: storage untouched (`len(idx.vectors)`)
    assert o.remove(12) == orc.ERR_ALREADY_DELETED and o.remove(9999) == orc.ERR_NOT_FOUND
    assert sorted(search_ids(o, kw, q, 10)) == [11, 14]
    assert sorted(search_ids(o, kw, q, 10, filter_ids=[11, 12, 13])) == [11]   # the filter does not bring a deleted row back (hnsw_index_document_filter_test.go:127-178)
    o.flush()
    assert stored(o) == 2 and sorted(search_ids(o, kw, q, 10)) == [11, 14]
    assert o.remove(12) == orc.ERR_NOT_FOUND                                   # gone for good after Flush
    o.flush()                                                                  # a second Flush with nothing to do
    assert stored(o) == 2
    assert o.remove(11) == 0 and o.remove(14) == 0
    o.flush()
    assert stored(o) == 0 and search_ids(o, kw, q, 10) == []


def test_hnsw_flush_of_the_entry_point_and_of_everything():
    """hnsw_index_test.go:586-629: the entry point is removed and flushed -> another node takes over; :631-672: everything flushed -> entry 0, maxLevel -1"""
    o = orc.HNSW(3, "l2", 16, 200, 200, seed=3)
    for i in range(1, 6):
        assert o.add(i, [i - 1, 0, 0]) == 0
    entry = o.entry(); assert entry == 1
    assert o.remove(entry) == 0
    o.flush()
    assert o.entry() != entry and o.entry() != 0 and o.capacity() == 4
    n, gi, _ = o.search(np.array([0, 0, 0], np.float32), 10, 0, cap=8)
    assert sorted(int(x) for x in gi[:n]) == [2, 3, 4, 5]
    o = orc.HNSW(3, "l2", 16, 200, 200)
    assert o.add(1, [1, 2, 3]) == 0 and o.add(2, [4, 5, 6]) == 0 and o.remove(1) == 0 and o.remove(2) == 0
    o.flush()
    assert o.capacity() == 0 and o.entry() == 0 and o.max_level() == -1
    assert o.search(np.array([1, 2, 3], np.float32), 10, 0, cap=4)[0] == 0
