"""On-disk formats: the oracle's WriteTo / ReadFrom restatement against byte images packed here,
field by field, from the reference's format comments (flat_index.go:348-360, ivf_index.go:441-462,
pq_index.go:480-505, ivfpq_index.go:507-535, hnsw_index.go:701-727) — a third, independent statement of the
layouts, so the oracle is pinned by the documented format and not only by itself. Error behaviour mirrors
ivf_index_test.go:1358 (invalid magic), flat_index_test.go (dimension / distance kind mismatch, truncation)."""
import struct

import numpy as np
import pytest

import oracle_lib as orc

EMPTY_ROARING = struct.pack("<II", 12346, 0)     # roaring portable format, no containers (SERIAL_COOKIE_NO_RUNCONTAINER)


def u32(x): return struct.pack("<I", x)
def f32s(a): return np.ascontiguousarray(a, np.float32).tobytes()
def kind(k): return u32(len(k)) + k.encode()
def tail(): return u32(len(EMPTY_ROARING)) + EMPTY_ROARING


def pack_flat(dim, metric, ids, vecs):
    b = b"FLAT" + u32(1) + u32(dim) + kind(metric) + u32(len(ids))
    for i, v in zip(ids, vecs):
        b += u32(int(i)) + u32(dim) + f32s(v)
    return b + tail()


def pack_ivf(dim, metric, nlist, centroids, lists):
    b = b"IVFX" + u32(1) + u32(dim) + kind(metric) + u32(nlist) + bytes([1 if centroids is not None else 0])
    if centroids is not None:
        for c in centroids:
            b += u32(dim) + f32s(c)
    b += u32(nlist)
    for ids, vecs in lists:
        b += u32(len(ids))
        for i, v in zip(ids, vecs):
            b += u32(int(i)) + f32s(v)
    return b + tail()


def pack_pq(dim, metric, M, nbits, codebooks, ids, codes):
    ksub, dsub = 1 << nbits, dim // M
    b = b"PQIX" + u32(1) + u32(dim) + kind(metric) + u32(M) + u32(nbits) + u32(ksub) + u32(dsub) + bytes([1 if codebooks is not None else 0])
    if codebooks is not None:
        for m in range(M):
            b += u32(ksub * dsub) + f32s(codebooks[m])
    b += u32(len(ids))
    for i, c in zip(ids, codes):
        b += u32(int(i)) + bytes(bytearray(c))
    return b + tail()


def pack_ivfpq(dim, metric, nlist, M, nbits, centroids, codebooks, lists):
    ksub, dsub = 1 << nbits, dim // M
    b = b"IVPQ" + u32(1) + u32(dim) + kind(metric) + u32(nlist) + u32(M) + u32(nbits) + u32(ksub) + u32(dsub) + bytes([1 if centroids is not None else 0])
    if centroids is not None:
        for c in centroids:
            b += u32(dim) + f32s(c)
        for m in range(M):
            b += u32(ksub * dsub) + f32s(codebooks[m])
    b += u32(nlist)
    for ids, codes in lists:
        b += u32(len(ids))
        for i, c in zip(ids, codes):
            b += u32(int(i)) + bytes(bytearray(c))
    return b + tail()


def pack_hnsw(dim, metric, M, efc, efs, max_level, entry, nodes):
    """nodes: {id: (level, vec, [edges per layer])}, written in ascending id order (the canonical order both sides use)."""
    b = b"HNSW" + u32(1) + u32(dim) + kind(metric) + u32(M) + u32(efc) + u32(efs)
    b += struct.pack("<d", 1.0 / orc.lib().orc_go_log(float(M))) + struct.pack("<i", max_level) + u32(entry) + u32(len(nodes))
    for i in sorted(nodes):
        level, vec, edges = nodes[i]
        b += u32(i) + struct.pack("<i", level) + u32(dim) + f32s(vec) + u32(len(edges))
        for e in edges:
            b += u32(len(e)) + np.ascontiguousarray(e, np.uint32).tobytes()
    return b + tail()


def synth(seed, n, d):
    return orc.synth(seed, 0, n * d).reshape(n, d)


@pytest.mark.parametrize("metric", ["l2", "l2_squared", "cosine"])
def test_flat_bytes_match_the_documented_layout(metric):
    X = synth(1, 7, 6)
    ids = [3, 10, 11, 12, 500, 7, 1]
    o = orc.Flat(6, metric)
    assert o.add_batch(ids, X) == 0
    stored = orc.preprocess_rows(metric, X)               # cosine stores normalised vectors (flat_index.go:182)
    assert o.to_bytes() == pack_flat(6, metric, ids, stored)
    assert o.remove(10) == 0 and o.remove(7) == 0           # WriteTo flushes first (flat_index.go:368)
    keep = [i for i in range(7) if ids[i] not in (10, 7)]
    b = o.to_bytes()
    assert b == pack_flat(6, metric, [ids[i] for i in keep], stored[keep])
    o2 = orc.Flat(6, metric)
    assert o2.from_bytes(b) == len(b) and o2.to_bytes() == b
    assert orc.Flat(6, metric).from_bytes(b + b"trailing") == len(b)      # ReadFrom consumes exactly its own bytes
    q = synth(2, 1, 6)[0]
    assert o2.search(q, 3)[1].tolist() == o.search(q, 3)[1].tolist()


def test_flat_read_errors():
    o = orc.Flat(4, "cosine"); o.add_batch([1, 2], synth(3, 2, 4))
    b = o.to_bytes()
    assert orc.Flat(4, "cosine").from_bytes(b"XXXX" + b[4:]) == -5          # invalid magic number (flat_index.go:516)
    assert orc.Flat(4, "cosine").from_bytes(b[:4] + u32(2) + b[8:]) == -5   # unsupported version
    assert orc.Flat(5, "cosine").from_bytes(b) == -2                        # dimension mismatch (:536)
    assert orc.Flat(4, "l2").from_bytes(b) == -5                            # distance kind mismatch (:553)
    for cut in (2, 9, 30, len(b) - 1):
        assert orc.Flat(4, "cosine").from_bytes(b[:cut]) == -8              # truncated stream
    victim = orc.Flat(4, "cosine"); victim.add_batch([9], synth(4, 1, 4))
    before = victim.to_bytes()
    assert victim.from_bytes(b[:40]) == -8 and victim.to_bytes() == before  # a failed ReadFrom leaves the index untouched


def test_ivf_bytes_match_the_documented_layout():
    d, nlist = 4, 3
    X = synth(5, 30, d)
    o = orc.IVF(d, "l2_squared", nlist)
    assert o.to_bytes() == pack_ivf(d, "l2_squared", nlist, None, [([], [])] * nlist)     # untrained: no centroids, empty lists
    assert o.train(X) == 0 and o.add_batch(range(1, 31), X) == 0
    assign = [orc.nearest_centroid(x, o.centroids(), "l2_squared") for x in X]
    lists = [([i + 1 for i in range(30) if assign[i] == l], X[[i for i in range(30) if assign[i] == l]]) for l in range(nlist)]
    b = o.to_bytes()
    assert b == pack_ivf(d, "l2_squared", nlist, o.centroids(), lists)
    o2 = orc.IVF(d, "l2_squared", nlist)
    assert o2.from_bytes(b) == len(b) and o2.to_bytes() == b
    assert orc.IVF(d, "l2_squared", 4).from_bytes(b) == -5                  # nlist mismatch (ivf_index.go:676)
    assert orc.IVF(d, "l2_squared", nlist).from_bytes(b"IVFY" + b[4:]) == -5
    q = synth(6, 1, d)[0]
    assert o2.search(q, 5, 2)[1].tolist() == o.search(q, 5, 2)[1].tolist()


def test_pq_and_ivfpq_bytes_match_the_documented_layout():
    d, M, nbits = 8, 4, 2
    X = synth(7, 64, d)
    o = orc.PQ(d, "l2_squared", M, nbits)
    assert o.train(X) == 0 and o.add_batch(range(1, 21), X[:20]) == 0
    assert o.remove(4) == 0
    b = o.to_bytes()                                                        # flushes id 4 away
    ids = [i for i in range(1, 21) if i != 4]
    assert b == pack_pq(d, "l2_squared", M, nbits, o.codebooks(), ids, o.codes())
    o2 = orc.PQ(d, "l2_squared", M, nbits)
    assert o2.from_bytes(b) == len(b) and o2.to_bytes() == b
    assert orc.PQ(d, "l2_squared", 2, nbits).from_bytes(b) == -5            # parameter M mismatch (pq_index.go:737)
    q = synth(8, 1, d)[0]
    assert np.array_equal(o2.search(q, 5)[2].view(np.uint32), o.search(q, 5)[2].view(np.uint32))

    nlist = 2
    T = np.arange(100 * d, dtype=np.float32).reshape(100, d) * 0.01           # the ramp of ivfpq_index_search_test.go:9-72
    p = orc.IVFPQ(d, "l2_squared", nlist, M, nbits)
    assert p.train(T) == 0 and p.add_batch(range(1, 41), T[:40]) == 0
    lists = [(p.list_ids(l), p.list_codes(l)) for l in range(nlist)]
    b = p.to_bytes()
    assert b == pack_ivfpq(d, "l2_squared", nlist, M, nbits, p.centroids(), p.codebooks(), lists)
    p2 = orc.IVFPQ(d, "l2_squared", nlist, M, nbits)
    assert p2.from_bytes(b) == len(b) and p2.to_bytes() == b
    assert p2.search(T[3], 5, 2)[1].tolist() == p.search(T[3], 5, 2)[1].tolist()
    assert orc.IVFPQ(d, "l2_squared", nlist, M, 3).from_bytes(b) == -5
    assert orc.IVFPQ(d, "l2_squared", nlist, M, nbits).from_bytes(b[:-5]) == -8


def test_hnsw_bytes_flush_and_round_trip():
    d, n = 6, 60
    X = synth(9, n, d)
    o = orc.HNSW(d, "l2", 4, 20, 16, seed=3)
    assert o.add_batch(range(1, n + 1), X) == 0
    ids, levels, vecs, eoff, edges = o.export()
    nodes, s = {}, 0
    for i in range(n):
        lay = []
        for _ in range(levels[i] + 1):
            lay.append(edges[eoff[s]:eoff[s + 1]]); s += 1
        nodes[int(ids[i])] = (int(levels[i]), vecs[i], lay)
    b = o.to_bytes()
    assert b == pack_hnsw(d, "l2", 4, 20, 16, o.max_level(), o.entry(), nodes)
    o2 = orc.HNSW(d, "l2", 4, 20, 16)
    assert o2.from_bytes(b) == len(b) and o2.to_bytes() == b
    Q = synth(10, 5, d)
    for q in Q:
        a, c = o.search(q, 5, 16), o2.search(q, 5, 16)
        assert a[1].tolist() == c[1].tolist() and np.array_equal(a[2].view(np.uint32), c[2].view(np.uint32))
    assert orc.HNSW(d, "l2", 5, 20, 16).from_bytes(b) == -5                 # parameter mismatch (hnsw_index.go:975)
    # Flush (hnsw_index.go:348-431): edges to deleted nodes disappear, a deleted entry point is re-seated
    entry = o.entry()
    for i in (entry, 7, 8):
        assert o.remove(i) == 0
    o.flush()
    ids2, levels2, _, eoff2, edges2 = o.export()
    assert set(ids2.tolist()) == set(range(1, n + 1)) - {entry, 7, 8} and not (set(edges2.tolist()) & {entry, 7, 8})
    assert o.entry() in ids2.tolist() and o.entry() != entry
    assert levels2[ids2.tolist().index(o.entry())] == o.max_level()
    b2 = o.to_bytes()
    o3 = orc.HNSW(d, "l2", 4, 20, 16)
    assert o3.from_bytes(b2) == len(b2) and o3.search(Q[0], 5, 16)[1].tolist() == o.search(Q[0], 5, 16)[1].tolist()
    # empty graph
    e = orc.HNSW(d, "cosine")
    be = e.to_bytes()
    assert be == pack_hnsw(d, "cosine", 16, 200, 200, -1, 0, {})
    assert orc.HNSW(d, "cosine").from_bytes(be) == len(be)
