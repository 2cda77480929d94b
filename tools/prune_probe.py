"""How much of the IVFPQ scan could a LOWER BOUND per (query, probed list) remove?  (experiment behind the fused filter's pruning)

For every (query, probed list) pair of the bench's IVFPQ leg: lb = sum_m min_k LUT[m][k] (no candidate of the list can score below it)
and the same for the subspaces behind each table phase (32 subspaces); compared with the query's final K-th best distance."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench as bn  # noqa: E402
import comet_amd as ca  # noqa: E402
from comet_amd._lib import check  # noqa: E402


def main():
    n, d, nlist, M, nbits, B, K, nprobe = 1_000_000, 768, 1024, 96, 8, 256, 10, 32
    ctx = ca.Context(0)
    idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, nlist, M, nbits)
    ntrain = nlist * 100
    tbuf = ctx.alloc(ntrain * d * 4)
    ctx.synth_mixture(tbuf, bn.MIX_SEED, bn.MIX_CENTERS, bn.MIX_SIGMA, bn.MIX_SUB, bn.MIX_NOISE, 0, ntrain, d)
    check(ctx.lib.comet_index_train_dev(idx.h, C.c_void_p(tbuf), ntrain)); ctx.free(tbuf)
    bn.add_rows(ctx, idx, 0, n, d, lambda buf, lo, m: ctx.synth_mixture(buf, bn.MIX_SEED, bn.MIX_CENTERS, bn.MIX_SIGMA, bn.MIX_SUB, bn.MIX_NOISE, lo, m, d))
    q_dev = ctx.alloc(B * d * 4)
    ctx.synth_mixture(q_dev, bn.MIX_SEED, bn.MIX_CENTERS, bn.MIX_SIGMA, bn.MIX_SUB, bn.MIX_NOISE, n + 7, B, d)
    ctx.sync()
    Q = ctx.download(q_dev, (B, d), np.float32)
    ids, sc, cn = idx.search_batch(Q, K, nprobes=nprobe)
    kth = sc[:, K - 1].astype(np.float64)                                   # distances are sqrt(sum): compare sums
    kth_sum = kth ** 2
    cent = idx.centroids(nlist).astype(np.float64)
    dsub = d // M
    cb = idx.codebooks(M, 256, dsub).astype(np.float64)                     # [M][256][dsub]
    _, e_lists, e_codes = idx.export(codes_width=M)
    list_len = np.bincount(e_lists, minlength=nlist)
    # geometric bound: a candidate of list L is c_L + rho with |rho|^2 = sum_m |cb[m][code_m]|^2 (orthogonal subspaces), so its ADC distance
    # to q is |(q - c_L) - rho| >= |q - c_L| - R_L with R_L = the largest |rho| among the list's codes — known at add time, free at query time
    cbn2 = (idx.codebooks(M, 256, d // M).astype(np.float64) ** 2).sum(2)   # [M][256]
    rho2 = cbn2[np.arange(M)[None, :], e_codes.astype(np.int64)].sum(1)
    R = np.zeros(nlist); np.maximum.at(R, e_lists, np.sqrt(rho2))
    q64 = Q.astype(np.float64)
    d2 = (q64 ** 2).sum(1)[:, None] + (cent ** 2).sum(1)[None, :] - 2.0 * (q64 @ cent.T)
    probed = np.argsort(d2, axis=1, kind="stable")[:, :nprobe]
    cb_n2 = (cb ** 2).sum(2)                                                # [M][256]
    tot = np.zeros(4); cand = 0
    death_hist = np.zeros(13, dtype=np.int64)
    geo_dead = 0; geo_or_lb = 0
    by_rank = np.zeros((nprobe, 4)); cand_rank = np.zeros(nprobe)
    for b in range(B):
        r = q64[b][None, :] - cent[probed[b]]                               # [np][d]
        rm = r.reshape(nprobe, M, dsub)
        # LUT[p][m][k] = |r_m|^2 + |cb|^2 - 2 r_m . cb
        lut = (rm ** 2).sum(2)[:, :, None] + cb_n2[None, :, :] - 2.0 * np.einsum("pmd,mkd->pmk", rm, cb)
        mins = lut.min(2)                                                   # [np][M]
        suffix = np.stack([mins[:, s:].sum(1) for s in (0, 32, 64)], 1)     # lower bound of what phases >= s add
        typical = np.stack([np.median(lut[:, :s, :], axis=2).sum(1) for s in (32, 64)], 1)   # a typical candidate's partial sum after 1 / 2 phases
        w = list_len[probed[b]].astype(np.float64)
        dco = np.sqrt(np.maximum(d2[b, probed[b]], 0.0))
        geo = np.maximum(dco - R[probed[b]], 0.0) ** 2 * (1.0 - 1e-4)
        geo_dead += int(((geo > kth_sum[b]) & (np.arange(nprobe) > 0)).sum())
        csum = np.cumsum(mins, axis=1)                                      # partial lower bound after m + 1 subspaces
        first = np.where(csum > kth_sum[b], np.arange(M)[None, :], M).min(1)   # subspaces needed until the pair is provably dead (M: never)
        for p_ in range(1, nprobe):
            death_hist[min(int(first[p_]) // 8, 12)] += 1
        dead0 = suffix[:, 0] > kth_sum[b]                                   # the whole item can be skipped for this query
        dead1 = typical[:, 0] + suffix[:, 1] > kth_sum[b]                   # typical candidate dead after phase 1 with the rest's lower bound
        dead2 = typical[:, 1] + suffix[:, 2] > kth_sum[b]
        dead1_plain = typical[:, 0] > kth_sum[b]; dead2_plain = typical[:, 1] > kth_sum[b]
        for i, v in enumerate((dead0, dead1, dead2)):
            tot[i] += (w * v).sum(); by_rank[:, i] += w * v
        tot[3] += (w * dead2_plain).sum(); by_rank[:, 3] += w * dead2_plain
        cand += w.sum(); cand_rank += w
    print(json.dumps({"candidates": int(cand),
                      "frac_pairs_dead_before_any_gather (sum of row minima > K-th sum)": tot[0] / cand,
                      "frac_typical_dead_after_phase1_with_min_rest": tot[1] / cand,
                      "frac_typical_dead_after_phase2_with_min_rest": tot[2] / cand,
                      "frac_typical_dead_after_phase2_plain_partial_sum": tot[3] / cand,
                      "pairs_behind_probe0_by_subspaces_until_dead (bins of 8; last = never)": death_hist.tolist(),
                      "pairs_behind_probe0_dead_by_the_geometric_bound": [geo_dead, B * (nprobe - 1)],
                      "dead_before_any_gather_by_probe_rank": [round(x, 3) for x in (by_rank[:, 0] / np.maximum(cand_rank, 1)).tolist()]}))


if __name__ == "__main__":
    main()
