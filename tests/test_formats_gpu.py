"""On-disk formats through the C ABI (comet_index_write_to / comet_index_read_from): the GPU indexes emit the
reference's FLAT / IVFX / PQIX / IVPQ / HNSW layouts byte for byte (compared with the oracle's restatement of
WriteTo on the same content), load what the oracle wrote, and answer searches identically afterwards.
Error cases mirror the reference's tests (ivf_index_test.go:1358 invalid magic; dimension / kind / parameter mismatch;
truncated streams leave the receiving index untouched)."""
import io

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, CometError, FlatIndex, HNSWIndex, IVFIndex, IVFPQIndex, PQIndex
from comet_amd._lib import ERR_FORMAT, ERR_IO

pytestmark = pytest.mark.gpu
METRICS = [EUCLIDEAN, L2_SQUARED, COSINE]


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_results(g, o, Q, k, **kw):
    gi, gs, gc = g.search_batch(Q, k, **{("nprobes" if a == "nprobes" else a): v for a, v in kw.items()})
    for b, q in enumerate(Q):
        args = [kw["nprobes"]] if "nprobes" in kw else ([kw.get("ef_search", 0)] if isinstance(o, orc.HNSW) else [])
        n, oi, os_ = o.search(q, k, *args)
        assert gc[b] == n and np.array_equal(gi[b, :n], oi) and np.array_equal(bits(gs[b, :n]), bits(os_)), b


@pytest.mark.parametrize("metric", METRICS)
def test_flat_write_read(ctx, metric):
    n, d = 700, 40
    X = synth(31, n, d); ids = np.arange(1, n + 1, dtype=np.uint32) * 3
    g = FlatIndex(ctx, d, metric); g.add_batch(ids, X)
    o = orc.Flat(d, metric); assert o.add_batch(ids, X) == 0
    for i in (3, 300, 2100):
        g.remove(i); assert o.remove(i) == 0
    gb = g.to_bytes()                                       # WriteTo flushes first
    assert gb == o.to_bytes() and len(g) == n - 3
    g2 = FlatIndex(ctx, d, metric)
    assert g2.from_bytes(gb + b"xyz") == len(gb)            # consumes exactly its own bytes
    assert len(g2) == n - 3 and g2.to_bytes() == gb
    Q = synth(32, 9, d)
    same_results(g2, o, Q, 10)
    # stream API with a short-reading reader
    class Dribble(io.RawIOBase):
        def __init__(self, b): self.b, self.o = b, 0
        def read(self, n=-1):
            n = min(n, 7); c = self.b[self.o:self.o + n]; self.o += len(c); return c
    g3 = FlatIndex(ctx, d, metric)
    assert g3.read_from(Dribble(gb)) == len(gb) and g3.to_bytes() == gb
    # a file written by the (restated) reference loads, and then appends like any index
    g3.add(999999, X[0].copy()); assert o.add(999999, X[0]) == 0
    assert g3.to_bytes() == o.to_bytes()


def test_flat_read_errors_leave_the_index_untouched(ctx):
    d = 8
    g = FlatIndex(ctx, d, COSINE); g.add_batch(np.arange(1, 21, dtype=np.uint32), synth(33, 20, d))
    b = g.to_bytes()
    victim = FlatIndex(ctx, d, COSINE); victim.add_batch(np.array([5, 6], np.uint32), synth(34, 2, d))
    before = victim.to_bytes()
    for bad, code, text in ((b"XXXX" + b[4:], ERR_FORMAT, "invalid magic number: expected 'FLAT', got 'XXXX'"),
                            (b[:4] + (7).to_bytes(4, "little") + b[8:], ERR_FORMAT, "unsupported version: 7"),
                            (b[:50], ERR_IO, "failed to read"), (b[:-1], ERR_IO, "failed to read bitmap data"),
                            # a zero-length soft-delete tail: roaring's UnmarshalBinary fails on it in the reference (flat_index.go:605-607)
                            (b[:-12] + (0).to_bytes(4, "little"), ERR_FORMAT, "failed to deserialize deleted nodes bitmap")):
        with pytest.raises(CometError) as e:
            victim.from_bytes(bad)
        assert e.value.code == code and text in str(e.value), str(e.value)
        assert victim.to_bytes() == before
    with pytest.raises(CometError, match="dimension mismatch: index has dim=9, serialized data has dim=8"):
        FlatIndex(ctx, 9, COSINE).from_bytes(b)
    with pytest.raises(CometError, match="distance kind mismatch: index uses 'l2', serialized data uses 'cosine'"):
        FlatIndex(ctx, d, EUCLIDEAN).from_bytes(b)
    with pytest.raises(CometError, match="invalid magic number: expected 'IVFX', got 'FLAT'"):   # ivf_index_test.go:1358
        IVFIndex(ctx, d, 2, COSINE).from_bytes(b)


@pytest.mark.parametrize("metric", METRICS)
def test_ivf_write_read(ctx, metric):
    n, d, nlist = 900, 24, 7
    X = synth(35, n, d); ids = np.arange(1, n + 1, dtype=np.uint32)
    g = IVFIndex(ctx, d, nlist, metric)
    empty = g.to_bytes()
    assert empty == orc.IVF(d, metric, nlist).to_bytes()    # untrained, empty
    g.train(X[:300]); g.add_batch(ids, X)
    o = orc.IVF(d, metric, nlist); assert o.train(X[:300]) == 0 and o.add_batch(ids, X) == 0
    for i in (1, 450, 900):
        g.remove(i); assert o.remove(i) == 0
    gb = g.to_bytes()
    assert gb == o.to_bytes()
    g2 = IVFIndex(ctx, d, nlist, metric)
    assert g2.from_bytes(gb) == len(gb) and g2.trained() and len(g2) == n - 3 and g2.to_bytes() == gb
    Q = synth(36, 6, d)
    same_results(g2, o, Q, 10, nprobes=3)
    with pytest.raises(CometError, match="nlist mismatch"):
        IVFIndex(ctx, d, nlist + 1, metric).from_bytes(gb)
    g3 = IVFIndex(ctx, d, nlist, metric)
    assert g3.from_bytes(empty) == len(empty) and not g3.trained() and len(g3) == 0


def test_pq_and_ivfpq_write_read(ctx):
    d, M, nbits = 32, 8, 4
    X = synth(37, 600, d); ids = np.arange(1, 601, dtype=np.uint32)
    g = PQIndex(ctx, d, L2_SQUARED, M, nbits); g.train(X[:200]); g.add_batch(ids, X)
    o = orc.PQ(d, "l2_squared", M, nbits); assert o.train(X[:200]) == 0 and o.add_batch(ids, X) == 0
    g.remove(77); assert o.remove(77) == 0
    gb = g.to_bytes()
    assert gb == o.to_bytes()
    g2 = PQIndex(ctx, d, L2_SQUARED, M, nbits)
    assert g2.from_bytes(gb) == len(gb) and g2.to_bytes() == gb
    same_results(g2, o, synth(38, 5, d), 10)
    with pytest.raises(CometError, match="parameter M mismatch: index has M=4, serialized data has M=8"):
        PQIndex(ctx, d, L2_SQUARED, 4, nbits).from_bytes(gb)

    nlist = 5
    for metric in (L2_SQUARED, COSINE):
        gi = IVFPQIndex(ctx, d, metric, nlist, M, nbits); gi.train(X[:300]); gi.add_batch(ids, X)
        oi = orc.IVFPQ(d, metric, nlist, M, nbits); assert oi.train(X[:300]) == 0 and oi.add_batch(ids, X) == 0
        for i in (2, 333):
            gi.remove(i); assert oi.remove(i) == 0
        gb = gi.to_bytes()
        assert gb == oi.to_bytes()
        g3 = IVFPQIndex(ctx, d, metric, nlist, M, nbits)
        assert g3.from_bytes(gb) == len(gb) and g3.to_bytes() == gb and len(g3) == 598
        same_results(g3, oi, synth(39, 5, d), 10, nprobes=2)
        with pytest.raises(CometError) as e:
            IVFPQIndex(ctx, d, metric, nlist, M, nbits).from_bytes(gb[:len(gb) // 2])
        assert e.value.code == ERR_IO


@pytest.mark.parametrize("metric", METRICS)
def test_hnsw_write_read_flush(ctx, metric):
    n, d, m = 800, 20, 6
    X = synth(41, n, d)
    o = orc.HNSW(d, metric, m, 40, 32, seed=5)
    assert o.add_batch(np.arange(1, n + 1), X) == 0
    ob = o.to_bytes()
    g = HNSWIndex(ctx, d, metric, m, 40, 32)
    assert g.from_bytes(ob) == len(ob) and len(g) == n        # a reference-format graph loads straight into the GPU index
    assert g.to_bytes() == ob
    Q = synth(42, 12, d)
    same_results(g, o, Q, 10, ef_search=32)
    # Flush on the GPU index = the reference's graph repair (hnsw_index.go:348-431), incl. a deleted entry point
    victims = [o.entry(), 5, 6, 7, 400]
    for i in victims:
        g.remove(i); assert o.remove(i) == 0
    same_results(g, o, Q, 10, ef_search=32)                   # soft-deleted
    g.flush(); o.flush()
    assert len(g) == n - len(victims)
    same_results(g, o, Q, 10, ef_search=32)
    assert g.to_bytes() == o.to_bytes()
    with pytest.raises(CometError, match="parameter M mismatch"):
        HNSWIndex(ctx, d, metric, m + 1, 40, 32).from_bytes(ob)
    e = HNSWIndex(ctx, d, metric)
    assert e.to_bytes() == orc.HNSW(d, metric).to_bytes()


def test_hnsw_duplicate_neighbours_are_harmless(ctx):
    """An edge list that names a neighbour twice (possible when a node id is re-added, hnsw_index.go:281) must give the
    reference's results: the second occurrence is already visited (hnsw_index.go:604). The oracle walks the duplicated
    graph as the reference would; the GPU kernel claims `visited` in parallel, so the loader removes the repeats."""
    n, d = 600, 16
    X = synth(43, n, d)
    o = orc.HNSW(d, "l2_squared", 5, 30, 24, seed=8)
    assert o.add_batch(np.arange(1, n + 1), X) == 0
    ids, levels, vecs, eoff, edges = o.export()
    # duplicate every third edge list's first and last entries in the middle of the list
    new_edges, new_off = [], [0]
    for s in range(len(eoff) - 1):
        e = edges[eoff[s]:eoff[s + 1]].tolist()
        if s % 3 == 0 and len(e) >= 2:
            e = e[:1] + [e[-1]] + e[1:] + [e[0]]
        new_edges += e; new_off.append(len(new_edges))
    dup = orc.HNSW(d, "l2_squared", 5, 30, 24)
    import test_formats_cpu as fc
    nodes, s = {}, 0
    for i in range(n):
        lay = []
        for _ in range(levels[i] + 1):
            lay.append(np.array(new_edges[new_off[s]:new_off[s + 1]], np.uint32)); s += 1
        nodes[int(ids[i])] = (int(levels[i]), vecs[i], lay)
    blob = fc.pack_hnsw(d, "l2_squared", 5, 30, 24, o.max_level(), o.entry(), nodes)
    assert dup.from_bytes(blob) == len(blob)
    g = HNSWIndex(ctx, d, L2_SQUARED, 5, 30, 24)
    g.load_graph(ids, levels, vecs, np.array(new_off, np.int64), np.array(new_edges, np.uint32), o.entry(), o.max_level())
    Q = synth(44, 32, d)
    same_results(g, dup, Q, 10, ef_search=24)
    same_results(g, o, Q, 10, ef_search=24)                   # and duplicates do not change the reference's answer either
    assert g.to_bytes() == blob                               # the stored graph keeps the edge lists as loaded
