#!/usr/bin/env python3
"""bench_ivfpq.py — second headline measurement: IVFPQ search (BASELINE configs[3] shape: nlist=4096, nprobe=32,
M=96, nbits=8, d=768) on one MI355X. Not the driver's bench (that is bench.py / Flat); run by hand, result committed
under profiles/.

Everything is built on the GPU (k-means training, coarse assignment, residual PQ encoding — exact arithmetic, so the
index is the one the reference would build from the same vectors, bit for bit). Reports queries/s, the ADC kernel's
roofline fraction (algorithmic bytes = sum over probed lists of len * M code bytes; the SURVEY.md §8d figure with +4
id bytes per code is printed beside it), recall@10 against the exact Flat search on the same data, and a CPU oracle
baseline that searches THE SAME index (quantizers + codes exported from the GPU) with a bit-exact parity check.

Data: mixture of `--centers` Gaussian-ish blobs drawn from the SplitMix64 stream (uniform-random 768-d vectors have no
cluster structure), sigma 0.15; queries are perturbed corpus rows.
"""
from __future__ import annotations

import argparse
import ctypes as C
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--M", type=int, default=96)
    ap.add_argument("--nbits", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--train", type=int, default=0, help="training vectors (default nlist*100, capped at rows)")
    ap.add_argument("--centers", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-queries", type=int, default=0, help="0 = one per host core (max batch)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("COMET_BENCH_FORCE_DIST") == "1"
    # SURVEY §8(e) for IVFPQ: every rank trains on the same vectors (deterministic GPU k-means -> replicated centroids and
    # codebooks) and owns the lists l % world == rank (comet_index_set_shard: members of other lists are dropped on add); the
    # per-shard top-K are all-gathered and merged inside the library (RCCL, comet_amd.dist.Comm) — no torch, no host sync
    import comet_amd as ca
    from comet_amd._lib import check
    import oracle_lib as orc
    ctx = ca.Context(local_rank)
    comm = None
    if use_dist:
        from comet_amd.dist import Comm
        comm = Comm.from_env(ctx)
    d, n = args.dim, args.rows
    ntrain = min(n, args.train or args.nlist * 100)

    # ---- synthetic clustered corpus, generated chunk-wise on the host from the deterministic stream ----
    centers = orc.synth(0x5EED, 0, max(1, args.centers) * d).reshape(max(1, args.centers), d)

    def rows(lo, hi):
        noise = orc.synth(0xC0FFEE + 4, lo * d, (hi - lo) * d).reshape(hi - lo, d)
        if args.centers <= 0:          # uniform data: balanced lists (throughput shape), no cluster structure
            return noise
        # blob index from a hash of the row number so that consecutive rows belong to different blobs
        blob = ((np.arange(lo, hi, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(args.centers)
        return (centers[blob.astype(np.int64)] + noise * np.float32(0.15)).astype(np.float32)

    idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, args.nlist, args.M, args.nbits)
    t0 = time.time()
    idx.train(rows(0, ntrain))
    if world > 1:
        idx.set_shard(rank, world)
    train_s = time.time() - t0
    t0 = time.time()
    chunk = 131072
    flat = None if args.no_recall else ca.FlatIndex(ctx, d, ca.L2_SQUARED)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        X = rows(lo, hi)
        ids = np.arange(lo + 1, hi + 1, dtype=np.uint32)
        idx.add_batch(ids, X)                          # list-sharded: the index keeps only the members of this rank's lists
        if flat is not None and world == 1:
            flat.add_batch(ids, X)
    add_s = time.time() - t0
    B, K = args.batch, args.k
    qrows = (np.arange(B) * 7919) % n
    Q = np.vstack([rows(int(r), int(r) + 1) for r in qrows]) + orc.synth(0xBEEF + 4, 0, B * d).reshape(B, d) * np.float32(0.05)

    q_dev = ctx.alloc(B * d * 4); ctx.upload(q_dev, Q)
    bufs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(2)]
    oi, os_, oc = bufs[0]

    def run(nsteps):            # batch i+1 is enqueued before batch i is waited for
        prev = None
        for i in range(nsteps):
            w = i & 1
            if comm is not None:
                t = comm.search_async(idx, q_dev, B, K, bufs[w][0], bufs[w][1], bufs[w][2], K, nprobes=args.nprobe)
            else:
                t = idx.search_batch_dev_async(q_dev, B, K, bufs[w][0], bufs[w][1], bufs[w][2], K, nprobes=args.nprobe)
            if prev is not None:
                comm.search_wait(idx, prev, block=False) if comm is not None else idx.search_wait(prev)
            prev = t
        if prev is not None:
            comm.search_wait(idx, prev, block=True) if comm is not None else idx.search_wait(prev)

    def barrier():
        ctx.sync()
        if comm is not None:
            comm.sync(); comm.barrier()
    run(max(1, args.warmup))
    barrier()
    ctx.profile_only("adc_scan"); ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    el = time.perf_counter() - t0
    prof = ctx.profile_dump()
    ctx.profile_only(None); ctx.profile_reset()
    run(args.steps); barrier()
    allk = ctx.profile_dump(); ctx.profile(False)
    prof_all = {k: round(v[0] / args.steps, 4) for k, v in sorted(allk.items())}
    if comm is not None:
        el = comm.allreduce_max(el)
    # one more search into buffer set 0 for the parity / recall sections below
    if comm is not None:
        comm.search_wait(idx, comm.search_async(idx, q_dev, B, K, oi, os_, oc, K, nprobes=args.nprobe), block=True); comm.sync()
    else:
        idx.search_batch_dev(q_dev, B, K, oi, os_, oc, K, nprobes=args.nprobe)
    ctx.sync()
    g_ids = ctx.download(oi, (B, K), np.uint32); g_sc = ctx.download(os_, (B, K), np.float32); g_cn = ctx.download(oc, (B,), np.int32)
    if world > 1:
        # multi-GPU line: throughput only (the exported single-rank index needed for the roofline / oracle sections lives on one rank)
        if rank == 0:
            print(json.dumps({"metric": "queries/sec, IVFPQ search, inverted lists sharded over the ranks (in-library RCCL all-gather of per-shard top-K)",
                              "value": B * args.steps / el, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
                              "config": {"workload": f"IVFPQ L2^2 {n}x{d}, nlist={args.nlist} nprobe={args.nprobe} M={args.M} nbits={args.nbits}, batch={B}, K={K}"},
                              "kernels_ms_per_step": prof_all}), flush=True)
        comm.close()
        return

    # ---- algorithmic bytes of the ADC scan: recompute the probed lists on the host from the exported state ----
    e_ids, e_lists, e_codes = idx.export(codes_width=args.M)
    list_len = np.bincount(e_lists, minlength=args.nlist)
    cent = idx.centroids(args.nlist)
    # exact coarse ranking of the reference (L2^2, sequential fp32) is what the GPU used; for the byte count a float64
    # ranking is sufficient (ties / last-ulp swaps change the probed set by at most one list of similar length)
    q64, c64 = Q.astype(np.float64), cent.astype(np.float64)
    d2 = (q64 ** 2).sum(1)[:, None] + (c64 ** 2).sum(1)[None, :] - 2.0 * (q64 @ c64.T)
    probed = np.argsort(d2, axis=1, kind="stable")[:, :args.nprobe]
    cand = int(list_len[probed].sum())
    code_bytes = cand * args.M
    adc_ms, adc_n = prof.get("adc_scan", (0.0, 1))
    adc_avg_ms = adc_ms / max(1, args.steps)        # per step: one step may split the batch over several launches
    achieved = code_bytes / (adc_avg_ms * 1e-3) / 1e9 if adc_avg_ms > 0 else 0.0

    line = {
        "metric": "queries/sec, IVFPQ search (BASELINE configs[3] shape) on 1 MI355X",
        "value": B * args.steps / el, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "dtype": "f32 tables / u8 codes", "data": "synthetic (clustered)",
        "config": {"workload": f"IVFPQ L2^2 {n}x{d}, nlist={args.nlist} nprobe={args.nprobe} M={args.M} nbits={args.nbits}, batch={B}, K={K}",
                   "train_vectors": ntrain, "train_s": round(train_s, 2), "add_s": round(add_s, 2), "max_list_len": int(list_len.max()),
                   "mean_list_len": float(list_len.mean())},
        "roofline": {"bound": "hbm", "kernel": "adc_scan", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": None, "avg_kernel_ms": adc_avg_ms, "algorithmic_bytes_per_launch": code_bytes,
                     "algorithmic_bytes_per_launch_with_ids": cand * (args.M + 4), "candidates_per_query": cand / B,
                     "lds_lookups_per_s": cand * args.M / (adc_avg_ms * 1e-3) if adc_avg_ms > 0 else 0.0},
        "kernels_ms_per_step": prof_all,
    }
    if flat is not None:
        f_ids = flat.search_batch(Q, K, mode=1)[0]
        line["recall_at_k_vs_exact_flat"] = float(np.mean([len(set(f_ids[b]) & set(g_ids[b, :g_cn[b]])) / K for b in range(B)]))
    if not args.no_cpu:
        cores = os.cpu_count() or 1
        o = orc.IVFPQ(d, "l2_squared", args.nlist, args.M, args.nbits)
        o.set_quantizers(cent, idx.codebooks(args.M, 1 << args.nbits, d // args.M))
        order = np.argsort(e_lists, kind="stable")
        bounds = np.concatenate([[0], np.cumsum(list_len)])
        for l in range(args.nlist):
            sel = order[bounds[l]:bounds[l + 1]]
            if sel.size:
                o.append_encoded(l, e_ids[sel], e_codes[sel])
        nq = min(B, args.cpu_queries or cores)
        bad, lock = [], threading.Lock()

        def work(lo, hi):
            for b in range(lo, hi):
                cnt, ci, cs = o.search(Q[b], K, args.nprobe, cap=K)
                ok = g_cn[b] == cnt and np.array_equal(g_ids[b, :cnt], ci) and np.array_equal(g_sc[b, :cnt].view(np.uint32), cs.view(np.uint32))
                if not ok:
                    with lock:
                        bad.append(b)
        T = min(cores, nq)
        t0 = time.time()
        th = [threading.Thread(target=work, args=(nq * t // T, nq * (t + 1) // T)) for t in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        cel = time.time() - t0
        line["cpu_baseline"] = {"value": nq / cel, "unit": "queries/s", "cores": T, "kind": "port",
                                "sample": f"{nq} of the batch's queries on the GPU-built index (quantizers + codes exported), {T} threads, {cel:.2f}s",
                                "parity_checked_queries": nq, "parity_mismatches": len(bad)}
    if comm is not None:
        comm.close()
        try:                                   # RCCL's version banner goes through C stdio: keep the JSON line last
            C.CDLL(None).fflush(None)
        except Exception:
            pass
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
