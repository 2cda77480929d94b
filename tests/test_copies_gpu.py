"""Host <-> device copies of the library (common.hpp: Ctx::h2d / Ctx::d2h) go through a pinned bounce buffer in 8 MiB pieces (round 6, DESIGN.md 5.1: the HIP runtime's own
staging of pageable hipMemcpyAsync delivered stale pieces under HSA_ENABLE_SDMA=0). Round trips across the piece boundaries, odd sizes, and a Flat index whose rows
span several pieces searched against the oracle."""
import numpy as np
import pytest

import oracle_lib as orc
from comet_amd import FlatIndex, L2_SQUARED

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nbytes", [1, 4096, (8 << 20) - 4, 8 << 20, (8 << 20) + 4, 3 * (8 << 20) + 12345 * 4])
def test_upload_download_round_trip_across_bounce_pieces(ctx, nbytes):
    n = max(1, nbytes // 4)
    a = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) ^ np.uint32(0x9E3779B9)
    d = ctx.alloc(n * 4)
    try:
        ctx.upload(d, a)
        b = ctx.download(d, (n,), np.uint32)
        assert np.array_equal(a, b)
        # a sub-range (source offsets inside the device buffer)
        if n > 1000:
            c = ctx.download(d + 400, (n - 100,), np.uint32)
            assert np.array_equal(a[100:], c)
    finally:
        ctx.free(d)


def test_flat_rows_spanning_several_bounce_pieces_match_the_oracle(ctx):
    n, d, B, k = 30_000, 200, 16, 10                       # 24 MB of rows: three pieces
    X = orc.synth(0xB0B0, 0, n * d).reshape(n, d)
    Q = orc.synth(0xB0B1, 0, B * d).reshape(B, d)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    g = FlatIndex(ctx, d, L2_SQUARED); g.add_batch(ids, X)
    o = orc.Flat(d, L2_SQUARED); assert o.add_batch(ids, X) == 0
    gi, gs, gc = g.search_batch(Q, k)
    for b in range(B):
        cnt, oi, os_ = o.search(Q[b], k)
        assert gc[b] == cnt and np.array_equal(gi[b, :cnt], oi) and np.array_equal(gs[b, :cnt].view(np.uint32), os_.view(np.uint32)), b
