"""adc_scan2_kernel (round 6: batched, phase-major ADC scan with the codeword slices in registers; kernels_adc2.inc.hpp) forced on
(COMET_ADC_KERNEL=2) over every shape it takes — 8-bit codebooks with 4 or 8 dimensions per subspace — against the CPU oracle, bit for bit:
ids, scores, counts. The library picks it by itself only for single-stage launches on lists of a few thousand codes; here it also runs the
stages of the two-stage search, PQ's single list (many segments), ragged phases (M not a multiple of 16), duos with holes, batches smaller
than four items, thresholds, filters, soft deletes and mass ties. Reference: pq_index_search.go:243-306, ivfpq_index_search.go:285-321,350-390."""
import os

import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import COSINE, EUCLIDEAN, L2_SQUARED, IVFPQIndex, PQIndex
from test_quant_gpu import bits, build_ivfpq, check_search, clustered, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_scan2():
    old = os.environ.get("COMET_ADC_KERNEL")
    os.environ["COMET_ADC_KERNEL"] = "2"
    yield
    if old is None:
        os.environ.pop("COMET_ADC_KERNEL", None)
    else:
        os.environ["COMET_ADC_KERNEL"] = old


@pytest.mark.parametrize("metric", [EUCLIDEAN, L2_SQUARED, COSINE])
@pytest.mark.parametrize("d,M", [(32, 8), (64, 16), (160, 20), (136, 34), (768, 96), (384, 96)])
def test_pq_scan2_matches_oracle(ctx, metric, d, M):
    """PQ: ONE list of 7000 codes = three segments of the same duo; M = 8 (half a phase), 16 (one), 20 / 34 (ragged last phase), 96 x dsub 8 / 4."""
    n, nbits = 7000, 8
    X = clustered(41, n, d, 30)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = PQIndex(ctx, d, metric, M, nbits); o = orc.PQ(d, metric, M, nbits)
    g.train(X[:1500]); assert o.train(X[:1500]) == 0
    g.add_batch(ids, X); assert o.add_batch(ids, X) == 0
    for B in (1, 2, 5, 9):                                  # one item, a full duo, duos + a hole, more than two batches of four
        Q = clustered(43, B, d, 30) + np.float32(0.02)
        check_search(g, o, Q, 10, None)
    Q = clustered(44, 5, d, 30) + np.float32(0.02)
    check_search(g, o, Q, 0, None)                          # every candidate comes back: the distance-matrix form of the epilogue
    check_search(g, o, Q, 100, None)                        # K beyond the fused filter
    ref = o.search(Q[0], 50)[2]
    check_search(g, o, Q, 50, None, threshold=float(ref[10]))
    check_search(g, o, Q, 9, None, filter_ids=list(range(100, 5000, 3)))
    for i in (1, 2, 777, 6999):
        g.remove(i); assert o.remove(i) == 0
    check_search(g, o, Q, 9, None)


def test_scan2_mass_ties_and_duplicates(ctx):
    n, d, M, nbits = 9000, 32, 8, 8
    X = clustered(61, n, d, 40)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    Y = np.repeat(X[:1], n, axis=0)
    g = PQIndex(ctx, d, L2_SQUARED, M, nbits); o = orc.PQ(d, L2_SQUARED, M, nbits)
    g.train(X[:2000]); assert o.train(X[:2000]) == 0
    g.add_batch(ids, Y); assert o.add_batch(ids, Y) == 0
    Q = clustered(63, 3, d, 40)
    check_search(g, o, Q, 10, None)
    check_search(g, o, Q, 64, None)
    X[5000:5300] = X[10:310]
    g2 = PQIndex(ctx, d, L2_SQUARED, M, nbits); o2 = orc.PQ(d, L2_SQUARED, M, nbits)
    g2.train(X[:2000]); assert o2.train(X[:2000]) == 0
    g2.add_batch(ids, X); assert o2.add_batch(ids, X) == 0
    Q2 = np.vstack([clustered(62, 4, d, 40) + np.float32(0.01), X[10:11]])
    for k in (1, 10, 64, 65):
        check_search(g2, o2, Q2, k, None)


@pytest.mark.parametrize("metric", [L2_SQUARED, COSINE])
def test_ivfpq_scan2_two_stage_and_single_stage(ctx, metric):
    """IVFPQ, M 96 x dsub 8 (the headline code shape): lists of uneven length (some longer than a 2560... 3072-code segment, some empty), every nprobe
    from one list to all of them, the default two-stage search (stages 1 and 2 through the forced kernel) and the every-candidate mode."""
    n, d, nlist, M, nbits = 12000, 768, 6, 96, 8
    X = clustered(71, n, d, 5, sigma=0.35)                  # five blobs over six lists: one list stays (nearly) empty, the others hold ~2400 codes, one > 3072
    X[:4000] = clustered(72, 4000, d, 1, sigma=0.2)
    Q = clustered(71, 9, d, 5, sigma=0.35) + np.float32(0.01)
    g, o = build_ivfpq(ctx, metric, X, X[:600], nlist, M, nbits)
    for nprobe in (1, 2, 3, 6):
        check_search(g, o, Q, 10, nprobe)
    check_search(g, o, Q[:1], 10, 6)
    check_search(g, o, Q, 64, 4)
    check_search(g, o, Q, 100, 4)                           # beyond the fused filter: distance matrix + radix selection
    ids, sc, cnt = g.search_batch(Q, 10, nprobes=4, mode=1)  # every candidate in one stage
    for b, q in enumerate(Q):
        nn, oi, os_ = o.search(q, 10, 4)
        assert cnt[b] == nn and np.array_equal(ids[b, :nn], oi) and np.array_equal(bits(sc[b, :nn]), bits(os_))
    check_search(g, o, Q, 7, 4, filter_ids=list(range(50, 9000, 7)))
    for i in (3, 5, 4001, 11999):
        g.remove(i); assert o.remove(i) == 0
    check_search(g, o, Q, 7, 4)


def test_ivfpq_scan2_dsub4_many_lists(ctx):
    """dsub 4 (d 64, M 16), 64 lists, a batch of 40 queries x 16 probes: hundreds of items, every queue, work stealing and the guided batch sizes."""
    n, d, nlist, M, nbits = 20000, 64, 64, 16, 8
    X = synth(0xC0FFEE + 9, n, d)
    Q = synth(0xBEEF + 9, 40, d)
    g, o = build_ivfpq(ctx, L2_SQUARED, X, X[:2560], nlist, M, nbits)
    check_search(g, o, Q, 10, 16)
    check_search(g, o, Q, 10, 64)
    ids, sc, cnt = g.search_batch(Q, 10, nprobes=16, mode=1)
    for b, q in enumerate(Q):
        nn, oi, os_ = o.search(q, 10, 16)
        assert cnt[b] == nn and np.array_equal(ids[b, :nn], oi) and np.array_equal(bits(sc[b, :nn]), bits(os_))
