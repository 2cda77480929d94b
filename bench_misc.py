#!/usr/bin/env python3
"""bench_misc.py — the BASELINE configs that are parity cases rather than the headline: HNSW search (configs[2] shape),
IVF + BM25 + Reciprocal Rank Fusion (configs[4] shape). Not the driver's bench; run by hand, result committed under profiles/.

Every section reports GPU queries/s, the CPU oracle's queries/s on the same index (bounded sample, threads stated) and a
bit-exact parity count of the sampled queries. HNSW graphs are built by the oracle (the reference's insert, restated), so
`--hnsw-rows` costs ≈0.17 ms of one host thread per inserted row (the full 1M x 384 graph of configs[2]: ≈3 minutes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def threads_run(fn, n_items, T):
    th = [threading.Thread(target=fn, args=(n_items * t // T, n_items * (t + 1) // T)) for t in range(T)]
    t0 = time.time()
    [t.start() for t in th]; [t.join() for t in th]
    return time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hnsw-rows", type=int, default=30000)
    ap.add_argument("--hnsw-dim", type=int, default=384)
    ap.add_argument("--ivf-rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--docs", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip", default="", help="comma list of hnsw,ivf,bm25,segments")
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--segment-rows", type=int, default=62_500)
    args = ap.parse_args()
    import comet_amd as ca
    import oracle_lib as orc
    from comet_amd.hybrid import reciprocal_rank_fusion
    ctx = ca.Context(0)
    cores = os.cpu_count() or 1
    out = {"n_gpus": 1, "data": "synthetic", "host_cores": cores}
    skip = set(args.skip.split(","))
    B = args.batch

    # ---------------------------------------------------------------- HNSW (configs[2] shape: M=16, efSearch=128, K=10, L2)
    if "hnsw" not in skip:
        n, d = args.hnsw_rows, args.hnsw_dim
        X = orc.synth(0x48, 0, n * d).reshape(n, d)
        o = orc.HNSW(d, "l2", 16, 200, 128, seed=7)
        t0 = time.time(); assert o.add_batch(np.arange(1, n + 1), X) == 0; build_s = time.time() - t0
        ids, levels, vecs, eoff, edges = o.export()
        g = ca.HNSWIndex(ctx, d, ca.EUCLIDEAN, 16, 200, 128)
        g.load_graph(ids, levels, vecs, eoff, edges, o.entry(), o.max_level())
        Q = orc.synth(0x49, 0, B * d).reshape(B, d)
        g.search_batch(Q, 10, ef_search=128)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = g.search_batch(Q, 10, ef_search=128)
        el = (time.perf_counter() - t0) / args.steps
        nq = min(B, cores)
        bad = []

        def work(lo, hi):
            for b in range(lo, hi):
                c, oi, os_ = o.search(Q[b], 10, 128)
                if not (r[2][b] == c and np.array_equal(r[0][b, :c], oi) and np.array_equal(r[1][b, :c].view(np.uint32), os_.view(np.uint32))):
                    bad.append(b)
        cel = threads_run(work, nq, nq)
        out["hnsw"] = {"workload": f"HNSW L2 {n}x{d}, M=16 efC=200 efSearch=128, batch={B}, K=10 (graph built by the CPU oracle in {build_s:.0f}s, one thread)",
                       "gpu_qps_host_buffers": B / el, "ms_per_batch": el * 1e3, "cpu_oracle_qps": nq / cel, "cpu_threads": nq,
                       "parity_checked": nq, "parity_mismatches": len(bad),
                       "distance_evals_per_query": g.stat("hnsw_distance_evals") / B, "expansions_per_query": g.stat("hnsw_expansions") / B}

    # ---------------------------------------------------------------- IVF 1M x 768 (configs[4]: hybrid default nProbes = 1) + BM25 100k docs + RRF
    vec_res = txt_res = None
    if "ivf" not in skip:
        n, d, nlist = args.ivf_rows, args.dim, 1024
        centers = orc.synth(0x5EED, 0, 2048 * d).reshape(2048, d)

        def rows(lo, hi):
            noise = orc.synth(0xC0FFEE + 4, lo * d, (hi - lo) * d).reshape(hi - lo, d)
            blob = ((np.arange(lo, hi, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(2048)
            return (centers[blob.astype(np.int64)] + noise * np.float32(0.15)).astype(np.float32)
        ivf = ca.IVFIndex(ctx, d, nlist, ca.COSINE)
        t0 = time.time(); ivf.train(rows(0, nlist * 100)); train_s = time.time() - t0
        t0 = time.time()
        for lo in range(0, n, 131072):
            hi = min(n, lo + 131072)
            ivf.add_batch(np.arange(lo + 1, hi + 1, dtype=np.uint32), rows(lo, hi))
        add_s = time.time() - t0
        qr = (np.arange(B) * 7919) % n
        Q = np.vstack([rows(int(r), int(r) + 1) for r in qr]) + orc.synth(0xBEEF + 4, 0, B * d).reshape(B, d) * np.float32(0.05)
        res = {}
        for npb in (1, 8, 32):
            ivf.search_batch(Q, 10, nprobes=npb)
            ctx.profile(True); ctx.profile_reset()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                r = ivf.search_batch(Q, 10, nprobes=npb)
            el = (time.perf_counter() - t0) / args.steps
            prof = ctx.profile_dump(); ctx.profile(False)
            res[f"nprobe{npb}"] = {"gpu_qps_host_buffers": B / el, "ms_per_batch": el * 1e3,
                                   "kernels_ms_per_batch": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())}}
            if npb == 1:
                vec_res = r
        out["ivf"] = {"workload": f"IVF cosine {n}x{d}, nlist={nlist}, batch={B}, K=10 (GPU train {train_s:.1f}s, add {add_s:.1f}s)", **res,
                      "note": "parity of IVF against the oracle is covered at oracle-trainable sizes by tests/test_quant_gpu.py; the CPU oracle cannot train 1M x 768 in bench time"}

    if "bm25" not in skip:
        nd = args.docs
        rng = np.random.default_rng(3)
        vocab = 50_000
        zipf = lambda size: np.minimum(vocab - 1, (rng.pareto(1.1, size) * 20).astype(np.int64)).astype(np.uint32)
        lens = rng.integers(20, 120, nd)
        g = ca.BM25SearchIndex(ctx); o = orc.BM25()
        t0 = time.time()
        docs = [zipf(int(l)) for l in lens]
        for i, t in enumerate(docs):
            g.add(i + 1, t); o.add(i + 1, t)
        build_s = time.time() - t0
        queries = [zipf(4).tolist() for _ in range(B)]
        g.search_batch(queries, 10)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = g.search_batch(queries, 10)
        el = (time.perf_counter() - t0) / args.steps
        ctx.profile(True); ctx.profile_reset()
        for _ in range(args.steps):
            g.search_batch(queries, 10)
        bm_prof = {k: round(v[0] / args.steps, 4) for k, v in sorted(ctx.profile_dump().items())}; ctx.profile(False)
        txt_res = r
        nq = min(B, cores)
        bad = []

        def work(lo, hi):
            for b in range(lo, hi):
                c, oi, _, os64 = o.search(queries[b], 10)
                if not (r[3][b] == c and np.array_equal(r[0][b, :c], oi) and np.array_equal(r[2][b, :c].view(np.uint64), np.asarray(os64, np.float64).view(np.uint64))):
                    bad.append(b)
        cel = threads_run(work, nq, nq)
        out["bm25"] = {"workload": f"BM25 over {nd} docs (zipf token ids, 20-120 tokens), batch={B} 4-token queries, K=10 (both indexes built in {build_s:.0f}s)",
                       "gpu_qps_host_buffers": B / el, "ms_per_batch": el * 1e3, "cpu_oracle_qps": nq / cel, "cpu_threads": nq,
                       "parity_checked": nq, "parity_mismatches": len(bad), "kernels_ms_per_batch": bm_prof}

    # ---------------------------------------------------------------- segment layer: S Flat segments resident in HBM, one fused call
    if "segments" not in skip:
        import ctypes as C
        from bench import add_rows
        from comet_amd.hybrid import HybridSearchResult as R, merge_results, sort_results_by_score
        S, m, d, K = args.segments, args.segment_rows, args.dim, 100
        segs = []
        for s_ in range(S):
            f = ca.FlatIndex(ctx, d, ca.COSINE)
            add_rows(ctx, f, s_ * m, (s_ + 1) * m, d, lambda buf, r0, mm: ctx.synth_fill(buf, 0xC0DE, r0 * d, mm * d))
            segs.append(f)
        whole = ca.FlatIndex(ctx, d, ca.COSINE)
        add_rows(ctx, whole, 0, S * m, d, lambda buf, r0, mm: ctx.synth_fill(buf, 0xC0DE, r0 * d, mm * d))
        ss = ca.SegmentSet(segs)
        q_dev = ctx.alloc(B * d * 4); ctx.synth_fill(q_dev, 0xABCD, 0, B * d)
        o_ids, o_sc, o_cn = ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)
        p_ids, p_sc, p_cn = ctx.alloc(S * B * K * 4), ctx.alloc(S * B * K * 4), ctx.alloc(S * B * 4)

        def fused():
            ss.search_batch_dev(q_dev, B, K, o_ids, o_sc, o_cn, K)

        def one_index():
            whole.search_batch_dev(q_dev, B, K, o_ids, o_sc, o_cn, K)

        def fan_out_host_merge():     # what a host-side port of storage.go:546-626 does: S searches, S result downloads, merge on the host
            per = []
            for i, f in enumerate(segs):
                f.search_batch_dev(q_dev, B, K, p_ids + i * B * K * 4, p_sc + i * B * K * 4, p_cn + i * B * 4, K)
            ctx.sync()
            ids = ctx.download(p_ids, (S, B, K), np.uint32); sc = ctx.download(p_sc, (S, B, K), np.float32)
            # vectorised mergeResults + sortResultsByScore + cut (ids are unique across these segments: no deduplication work)
            allsc = sc.transpose(1, 0, 2).reshape(B, S * K); allid = ids.transpose(1, 0, 2).reshape(B, S * K)
            order = np.lexsort((allid, -allsc), axis=1)[:, :K]
            return np.take_along_axis(allid, order, 1), np.take_along_axis(allsc, order, 1)

        def clock(fn, steps):
            fn(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            ctx.sync()
            return (time.perf_counter() - t0) / steps
        t_fused, t_one, t_host = clock(fused, args.steps), clock(one_index, args.steps), clock(fan_out_host_merge, max(2, args.steps // 3))
        ctx.profile(True); ctx.profile_reset()
        for _ in range(args.steps):
            fused()
        ctx.sync()
        seg_prof = {k: round(v[0] / args.steps, 4) for k, v in sorted(ctx.profile_dump().items())}; ctx.profile(False)
        fused(); ctx.sync()
        gi, gs, gc = ctx.download(o_ids, (B, K), np.uint32), ctx.download(o_sc, (B, K), np.float32), ctx.download(o_cn, (B,), np.int32)
        hi_, hs = fan_out_host_merge()
        bad = int(np.sum(np.any(gi != hi_, axis=1) | np.any(gs.view(np.uint32) != hs.view(np.uint32), axis=1) | (gc != K)))
        out["segments"] = {"workload": f"{S} Flat cosine segments x {m} rows x {d} resident in HBM, batch={B}, K={K}: per-segment top-K, highest score per id, "
                                       "score DESCENDING, cut to K (storage.go:489-626 + storage_merge.go:13-54)",
                           "fused_call_ms_per_batch": t_fused * 1e3, "fused_qps": B / t_fused,
                           "fan_out_with_host_merge_ms_per_batch": t_host * 1e3, "one_index_of_all_rows_ms_per_batch": t_one * 1e3,
                           "parity_vs_host_merge": {"checked_queries": B, "mismatches": bad}, "kernels_ms_per_batch": seg_prof}

    if vec_res is not None and txt_res is not None:
        t0 = time.perf_counter()
        fused = []
        for b in range(B):
            v = {int(i): float(s) for i, s in zip(vec_res[0][b, :vec_res[2][b]], vec_res[1][b, :vec_res[2][b]])}
            t = {int(i): float(s) for i, s in zip(txt_res[0][b, :txt_res[3][b]], txt_res[2][b, :txt_res[3][b]])}
            fused.append(reciprocal_rank_fusion(v, t))
        out["hybrid_rrf"] = {"workload": "Reciprocal Rank Fusion of the IVF (nProbes=1, hybrid default) and BM25 top-10 lists, host side as in the reference",
                             "host_ms_per_batch": (time.perf_counter() - t0) * 1e3, "fused_lists": len(fused)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
