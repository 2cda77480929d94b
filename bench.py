#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: queries/sec + recall@10, 1M x 768 Flat & IVFPQ, on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU. Rank 0 prints ONE JSON line.

Legs (all inside the one line):
  * headline = BASELINE configs[1]: Flat cosine 1M x 768, batch = 256 queries, K = 100 (uniform SplitMix64 data,
    SURVEY.md §8d). A "step" = one batch through the whole search path (preprocess -> scan -> top-K -> ids) with
    queries and results resident in HBM. `value` = queries/s of this leg. N > 1: the SAME 1M rows sharded by
    contiguous row blocks ("scaling": "strong"), per-shard top-K exchanged with one RCCL all-gather and merged;
  * "flat_l2" (N = 1): Flat L2^2 over a clustered 1M x 768 corpus at batches where HBM binds (B = 1 and B = 64,
    the narrow scan tile) and at B = 256, each with its own roofline — the north star's ">= 70 % of HBM roofline on
    the Flat L2 scan" is read off the B <= 64 lines;
  * "ivfpq" (N = 1): IVFPQ over the same clustered corpus, nlist 1024, nprobe 32, M 96, nbits 8, K 10, B 256:
    queries/s, roofline of adc_scan (algorithmic bytes = sum over probed lists of len * M),
    recall@10 against the exact Flat L2^2 search on the same corpus and against the CPU oracle's IVFPQ search of the
    same index (handed over in the reference's IVPQ on-disk format), its own cpu_baseline.
Timing: W warm-up steps, then R (= --regions, default 5) timed regions of EXACTLY K steps each, every region
bracketed by barrier + sync on both sides and max-reduced over ranks; the MEDIAN region is reported (all R are listed).
`roofline`: algorithmic bytes of the dominant kernel / its mean launch duration, measured with HIP events recorded on
the library's own stream around every launch of the timed regions (comet_profile_*), against the 8 TB/s HBM peak;
`traffic` comes from a separate `rocprofv3 --pmc` pass committed under profiles/ (its file is named in the line).
`cpu_baseline`: the CPU oracle (C++ restatement of the reference's Go loops — no Go toolchain in this image) timed on
this box's host cores on a bounded sample of the same workload, rank 0, N = 1, with a bit-exact parity check.
The product path never touches the oracle: it is imported only inside cpu_baseline_*().
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_ROWS, DIM, BATCH, TOPK = 1_000_000, 768, 256, 100
CORPUS_SEED, QUERY_SEED = 0xC0FFEE + 2, 0xBEEF + 2
# clustered corpus of the L2 / IVFPQ legs: 2048 centres, 65536 sub-centres at 0.15 around them (~15 rows each at 1M rows), 0.02 noise:
# a query's true neighbours are the rows of its own sub-centre
MIX_SEED, MIX_CENTERS, MIX_SIGMA, MIX_SUB, MIX_NOISE = 0xC0FFEE + 4, 2048, 0.15, 65536, 0.02
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--k", type=int, default=TOPK)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--mode", type=int, default=0, help="0 auto, 1 strict exact kernels, 2 fast path")
    ap.add_argument("--legs", default="flat,flat_l2,ivfpq", help="comma list; flat is always run (it is the headline)")
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--M", type=int, default=96)
    ap.add_argument("--nbits", type=int, default=8)
    ap.add_argument("--ivfpq-k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ helpers
def add_rows(ctx, idx, row_lo, row_hi, dim, fill):
    """Generate rows [row_lo, row_hi) on the device with fill(buf, lo, m) and add them (ids = row + 1)."""
    from comet_amd._lib import check
    chunk = 65536
    buf = ctx.alloc(chunk * dim * 4)
    idbuf = ctx.alloc(chunk * 4)
    for lo in range(row_lo, row_hi, chunk):
        m = min(chunk, row_hi - lo)
        fill(buf, lo, m)
        ctx.upload(idbuf, np.arange(lo + 1, lo + m + 1, dtype=np.uint32))
        added = C.c_int64()
        check(ctx.lib.comet_index_add_dev(idx.h, C.c_void_p(idbuf), C.c_void_p(buf), m, C.byref(added)))
        assert added.value == m
    ctx.free(buf)
    ctx.free(idbuf)


class Timer:
    """R regions of K steps; each region bracketed by barrier(); max over ranks; median reported."""

    def __init__(self, barrier, reduce_max):
        self.barrier, self.reduce_max = barrier, reduce_max

    def run(self, step_fn, steps, warmup, regions):
        step_fn(max(1, warmup))
        times = []
        for _ in range(max(1, regions)):
            self.barrier()
            t0 = time.perf_counter()
            step_fn(steps)
            self.barrier()
            times.append(self.reduce_max(time.perf_counter() - t0))
        return statistics.median(times), times


def timed(ctx, timer, step, args, dominant):
    """Timed regions with HIP events around the DOMINANT kernel only (every timed scope is two event records — barrier packets —
    and ~10 us of dispatch latency in a chain of short kernels; with all scopes timed a Flat step is 6 % slower), then one
    untimed region with every scope timed for the per-kernel breakdown. Returns (median s, region times, dominant-kernel
    profile of the timed regions, per-kernel ms per step of the breakdown region)."""
    ctx.profile_only(dominant); ctx.profile(True); ctx.profile_reset()
    med, times = timer.run(step, args.steps, args.warmup, args.regions)
    prof = ctx.profile_dump()
    ctx.profile_only(None); ctx.profile_reset()
    step(args.steps)
    ctx.sync()
    allk = ctx.profile_dump(); ctx.profile(False)
    # (the chip is power-limited in the Flat scan: with idle gaps between the kernels — which the extra event records add — the scan
    # itself runs a few percent faster, so the same kernel reads shorter in this region than in the timed ones)
    return med, times, prof, {k: round(v[0] / args.steps, 4) for k, v in sorted(allk.items())}


def kernel_stats(prof, name, launches_hint):
    ms, n = prof.get(name, (0.0, 0))
    return (ms / n if n else 0.0), n


def pmc_traffic(kernel, rows_local):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass (newest profiles/r*_pmc*.json naming the kernel)."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*pmc*.json")):
        try:
            pm = json.loads(f.read_text())
            for kname, kv in pm.get("kernels", {}).items():
                # profile scope -> kernel symbols: the scope "flat_scan_f16" is the query-stationary tile flat_scan_q8_kernel since round 2
                # (the cosine instantiation <0, ...> is the headline's; the 2 x 4 tile flat_scan_f16_kernel serves odd K-step counts)
                names = {"flat_scan_f16": ("flat_scan_q8_kernelILi0E", "flat_scan_f16_kernel")}.get(kernel, (kernel + "_kernel",))
                if kname.split("<")[0].strip() == kernel or any(nm in kname for nm in names):
                    per = kv.get("hbm_read_bytes_per_launch_corrected", 0) + kv.get("hbm_write_bytes_per_launch_uncalibrated", 0)
                    rows_ref = pm.get("rows", 1_000_000)
                    best = (per * rows_local / rows_ref, f"profiles/{f.name} (separate rocprofv3 --pmc pass: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, "
                                                         f"measured at {rows_ref} rows, scaled to {rows_local})")
        except Exception:
            continue
    return best if best else (None, "not measured in this run (PMC needs its own rocprofv3 pass)")


def flat_roofline(prof, steps_total, rows_local, dim, nq):
    ldh = (dim + 63) // 64 * 64
    name = "flat_scan_f16_n64" if (nq <= 64 and "flat_scan_f16_n64" in prof) else ("flat_scan_f16" if "flat_scan_f16" in prof else "dist_exact")
    avg_ms, n = kernel_stats(prof, name, steps_total)
    if name.startswith("flat_scan_f16"):
        qtile = 64 if name.endswith("n64") else 256
        alg = rows_local * ldh * 2 + qtile * ldh * 2          # the fp16 shadow once + the staged query tile
        flops = 2.0 * qtile * rows_local * ldh
    else:
        alg = rows_local * dim * 4                            # exact-arithmetic scan: every fp32 row once (SURVEY §8d N*d*4)
        flops = 0.0
    ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, src = pmc_traffic(name, rows_local)
    tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "avg_kernel_ms": avg_ms, "launches": n,
            "algorithmic_bytes_per_launch": alg, "fp32_rows_bytes_per_pass": rows_local * dim * 4,
            "mfma_tflops": tf, "mfma_frac_of_2500": tf / 2500.0}


def threads_map(fn, n_items, T):
    th = [threading.Thread(target=fn, args=(n_items * t // T, n_items * (t + 1) // T)) for t in range(T)]
    t0 = time.time()
    [t.start() for t in th]
    [t.join() for t in th]
    return time.time() - t0


# ------------------------------------------------------------------------------------------------ CPU baselines (the only users of the oracle)
def cpu_baseline_flat(args, ids_gpu, scores_gpu, counts_gpu):
    """Oracle Flat search on a bounded sample of the batch vs the full corpus: single-thread latency of one query, then as many
    queries as fit in ~cpu_seconds with one query per host thread; every sampled query is compared bit for bit with the GPU's."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    cores = os.cpu_count() or 1
    X = orc.synth(CORPUS_SEED, 0, args.rows * args.dim).reshape(args.rows, args.dim)
    Q = orc.synth(QUERY_SEED, 0, args.batch * args.dim).reshape(args.batch, args.dim)
    o = orc.Flat(args.dim, args.metric)
    t0 = time.time()
    o.add_batch(np.arange(1, args.rows + 1, dtype=np.uint32), X)
    build_s = time.time() - t0
    del X
    t0 = time.time()
    o.search(Q[args.batch - 1], args.k)
    single_s = time.time() - t0
    done, lock, mismatches = [], threading.Lock(), []
    deadline = time.time() + args.cpu_seconds
    nxt = [0]

    def worker(_lo, _hi):
        while True:
            with lock:
                qi = nxt[0]
                if qi >= args.batch or (time.time() > deadline and len(done) >= cores):
                    return
                nxt[0] += 1
            n, oi, os_ = o.search(Q[qi], args.k)
            with lock:
                done.append(qi)
                ok = counts_gpu[qi] == n and np.array_equal(ids_gpu[qi, :n], oi) and np.array_equal(scores_gpu[qi, :n].view(np.uint32), os_.view(np.uint32))
                if not ok:
                    mismatches.append(qi)
    el = threads_map(worker, cores, cores)
    return {"value": len(done) / el, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{len(done)} of the batch's {args.batch} queries vs the full {args.rows}x{args.dim} corpus, {cores} threads (one query each), "
                      f"{el:.1f}s; oracle index build {build_s:.1f}s not timed",
            "single_thread_latency_s": single_s, "single_thread_qps": 1.0 / single_s if single_s > 0 else None,
            "parity_checked_queries": len(done), "parity_mismatches": len(mismatches)}


def cpu_baseline_ivfpq(args, blob, Q, K, g_ids, g_sc, g_cn, flat_ids):
    """The oracle loads the GPU-built index from the reference's IVPQ on-disk bytes and searches a bounded query sample
    (one query per thread); parity + the oracle's own recall vs the exact Flat results."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    cores = os.cpu_count() or 1
    o = orc.IVFPQ(args.dim, "l2_squared", args.nlist, args.M, args.nbits)
    t0 = time.time()
    assert o.from_bytes(blob) == len(blob)
    load_s = time.time() - t0
    t0 = time.time()
    o.search(Q[0], K, args.nprobe, cap=K)
    single_s = time.time() - t0
    nq = min(len(Q), max(cores, 64))
    bad, lock, orecall = [], threading.Lock(), []

    def work(lo, hi):
        for b in range(lo, hi):
            cnt, ci, cs = o.search(Q[b], K, args.nprobe, cap=K)
            ok = g_cn[b] == cnt and np.array_equal(g_ids[b, :cnt], ci) and np.array_equal(g_sc[b, :cnt].view(np.uint32), cs.view(np.uint32))
            with lock:
                if not ok:
                    bad.append(b)
                orecall.append(len(set(flat_ids[b].tolist()) & set(ci.tolist())) / K)
    T = min(cores, nq)
    el = threads_map(work, nq, T)
    return {"value": nq / el, "unit": "queries/s", "cores": T, "kind": "port",
            "sample": f"{nq} of the batch's queries on the GPU-built index (loaded from its IVPQ bytes in {load_s:.1f}s), {T} threads, {el:.2f}s",
            "single_thread_latency_s": single_s, "parity_checked_queries": nq, "parity_mismatches": len(bad),
            "oracle_recall_at_k_vs_exact_flat": float(np.mean(orecall))}


# ------------------------------------------------------------------------------------------------ legs
def leg_flat_l2(ctx, ca, args, timer, flat2, q_dev):
    """Flat L2^2 at B = 1, 64, 256 on the clustered corpus."""
    out = {"workload": f"Flat l2_squared {args.rows}x{args.dim} (clustered corpus: {MIX_CENTERS} centres, {MIX_SUB} sub-centres at {MIX_SIGMA}, noise {MIX_NOISE}), K={args.ivfpq_k}"}
    K = args.ivfpq_k
    for B in (1, 64, 256):
        bufs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(2)]

        def step(nsteps, B=B, bufs=bufs):
            prev = None
            for i in range(nsteps):
                w = i & 1
                t = flat2.search_batch_dev_async(q_dev, B, K, bufs[w][0], bufs[w][1], bufs[w][2], K)
                if prev is not None:
                    flat2.search_wait(prev)
                prev = t
            if prev is not None:
                flat2.search_wait(prev)
        step(2)
        med, times, prof, allk = timed(ctx, timer, step, args, "flat_scan_f16_n64" if B <= 64 else "flat_scan_f16")
        out[f"batch{B}"] = {"qps": B * args.steps / med, "ms_per_step": med / args.steps * 1e3, "region_ms": [round(t * 1e3, 3) for t in times],
                            "roofline": flat_roofline(prof, args.steps * len(times), args.rows, args.dim, B),
                            "kernels_ms_per_step": allk}
        for b in bufs:
            for p in b:
                ctx.free(p)
    return out


def leg_ivfpq(ctx, ca, args, timer, flat2, q_dev, Q_host, comm=None, rank=0, world=1):
    """world > 1: every rank trains on the same vectors (deterministic GPU k-means: replicated quantisers), owns the inverted lists
    l % world == rank (comet_index_set_shard) and adds every row (foreign members are dropped); searches go through the in-library
    RCCL exchange, so every rank ends up with the merged global top-K."""
    B, K, d, n = args.batch, args.ivfpq_k, args.dim, args.rows
    idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, args.nlist, args.M, args.nbits)
    ntrain = min(n, args.nlist * 100)
    tbuf = ctx.alloc(ntrain * d * 4)
    ctx.synth_mixture(tbuf, MIX_SEED, MIX_CENTERS, MIX_SIGMA, MIX_SUB, MIX_NOISE, 0, ntrain, d)
    from comet_amd._lib import check
    t0 = time.time()
    check(ctx.lib.comet_index_train_dev(idx.h, C.c_void_p(tbuf), ntrain))
    train_s = time.time() - t0
    ctx.free(tbuf)
    if world > 1:
        idx.set_shard(rank, world)
    t0 = time.time()
    add_rows(ctx, idx, 0, n, d, lambda buf, lo, m: ctx.synth_mixture(buf, MIX_SEED, MIX_CENTERS, MIX_SIGMA, MIX_SUB, MIX_NOISE, lo, m, d))
    ctx.sync()
    add_s = time.time() - t0
    bufs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(2)]
    oi, os_, oc = bufs[0]

    def step(nsteps):       # batch i+1 is enqueued before batch i is waited for, as in the Flat leg
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
    step(2)
    stat0 = (idx.stat("adc_pairs_alive"), idx.stat("adc_pairs_behind_nearest"))
    med, times, prof, allk = timed(ctx, timer, step, args, "adc_scan")
    stat1 = (idx.stat("adc_pairs_alive"), idx.stat("adc_pairs_behind_nearest"))
    # the same search with every candidate scored (mode 1: no lower-bound pruning): what the ADC kernel itself sustains
    searches_two_stage = args.steps * len(times) + max(1, args.warmup) + args.steps
    ex_prof = ex_med = None
    if comm is None:
        def step_all(nsteps):
            for i in range(nsteps):
                idx.search_batch_dev(q_dev, B, K, bufs[i & 1][0], bufs[i & 1][1], bufs[i & 1][2], K, nprobes=args.nprobe, mode=1)
        step_all(2)
        ex_steps = max(3, args.steps // 2)
        ctx.profile_only("adc_scan"); ctx.profile(True); ctx.profile_reset()
        ctx.sync(); t0 = time.perf_counter(); step_all(ex_steps); ctx.sync(); ex_med = (time.perf_counter() - t0) / ex_steps
        ex_prof = ctx.profile_dump(); ctx.profile_only(None); ctx.profile(False)
        idx.search_batch_dev(q_dev, B, K, bufs[1][0], bufs[1][1], bufs[1][2], K, nprobes=args.nprobe, mode=1); ctx.sync()
        x_ids = ctx.download(bufs[1][0], (B, K), np.uint32); x_sc = ctx.download(bufs[1][1], (B, K), np.float32); x_cn = ctx.download(bufs[1][2], (B,), np.int32)
    if comm is not None:
        comm.search_wait(idx, comm.search_async(idx, q_dev, B, K, oi, os_, oc, K, nprobes=args.nprobe), block=True); comm.sync()
    else:
        idx.search_batch_dev(q_dev, B, K, oi, os_, oc, K, nprobes=args.nprobe)
    ctx.sync()
    g_ids = ctx.download(oi, (B, K), np.uint32); g_sc = ctx.download(os_, (B, K), np.float32); g_cn = ctx.download(oc, (B,), np.int32)
    # exact Flat L2^2 top-K on the same corpus (strict kernels): the recall reference
    f_ids = flat2.search_batch(Q_host, K, mode=1)[0]
    recall = float(np.mean([len(set(f_ids[b].tolist()) & set(g_ids[b, :g_cn[b]].tolist())) / K for b in range(B)]))
    # algorithmic bytes of one adc_scan launch: sum over the batch's probed lists of len * M (+4 id bytes beside it)
    e_ids, e_lists, _ = idx.export()
    list_len = np.bincount(e_lists, minlength=args.nlist)
    cent = idx.centroids(args.nlist).astype(np.float64)
    q64 = Q_host.astype(np.float64)
    d2 = (q64 ** 2).sum(1)[:, None] + (cent ** 2).sum(1)[None, :] - 2.0 * (q64 @ cent.T)       # fp64 ranking: enough for a byte count
    probed = np.argsort(d2, axis=1, kind="stable")[:, :args.nprobe]
    cand = int(list_len[probed].sum())
    total_steps = args.steps * len(times) + max(1, args.warmup)
    # roofline of the ADC kernel: from the every-candidate search (mode 1) when it ran — in the pruned search the kernel only sees what
    # the lower bound left, and the algorithmic-bytes convention (one code byte per probed candidate and subspace) stops measuring it
    if ex_prof is not None:
        adc_ms, adc_n = ex_prof.get("adc_scan", (0.0, 0)); adc_per_step = adc_ms / ex_steps
    else:
        adc_ms, adc_n = prof.get("adc_scan", (0.0, 0)); adc_per_step = adc_ms / total_steps
    ach = cand * args.M / (adc_per_step * 1e-3) / 1e9 if adc_per_step > 0 else 0.0
    traffic, src = pmc_traffic("adc_scan", n)
    out = {"workload": f"IVFPQ l2_squared {n}x{d} (clustered corpus), nlist={args.nlist} nprobe={args.nprobe} M={args.M} nbits={args.nbits}, batch={B}, K={K}"
                       + (f"; inverted lists sharded over {world} ranks (roofline block: rank 0's lists)" if world > 1 else ""),
           "qps": B * args.steps / med, "ms_per_step": med / args.steps * 1e3, "region_ms": [round(t * 1e3, 3) for t in times],
           "recall_at_10_vs_exact_flat": recall, "train_vectors": ntrain, "train_s": round(train_s, 2), "add_s": round(add_s, 2),
           "max_list_len": int(list_len.max()), "mean_list_len": float(list_len.mean()), "candidates_per_query": cand / B,
           "roofline": {"bound": "hbm", "kernel": "adc_scan", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": src, "avg_kernel_ms": adc_per_step, "launches": adc_n,
                        "algorithmic_bytes_per_launch": cand * args.M, "algorithmic_bytes_per_launch_with_ids": cand * (args.M + 4),
                        "lds_lookups_per_s": cand * args.M / (adc_per_step * 1e-3) if adc_per_step > 0 else 0.0,
                        "note": "SURVEY 8(d) convention: one code byte per (query, candidate, subspace). The kernel is bound by LDS gathers, not HBM: "
                                "queries probing the same list are scanned two at a time (one ds_read_b64 of the interleaved table serves both), so "
                                "a list's codes are physically read once per pair and mostly from L2 — the counter traffic beside this figure is "
                                "far below the algorithmic bytes and the fraction can exceed 1. Gather ceiling (random 8-byte reads, 32 lanes on 32 "
                                "bank pairs, ~3.5 deep): 256 CUs x 2.4 GHz x 128 values / 7 cycles = 11.2e12 lookups/s. Even this every-candidate pass "
                                "skips gathers: a wave whose partial sums are all above the running bounds after a table phase stops (exact), so "
                                "the lookups/s figure counts lookups the reference would do, not lookups issued.",
                        "lds_gather_frac_of_model_ceiling": (cand * args.M / (adc_per_step * 1e-3) / 11.2e12) if adc_per_step > 0 else 0.0},
           "kernels_ms_per_step": allk}
    out["roofline"]["measured_on"] = ("the every-candidate search (mode 1), see `every_candidate_search`" if ex_prof is not None
                                      else "the pruned search (sharded run): only the candidates the lower bound left were read, the fraction is not a bandwidth")
    alive, behind = stat1[0] - stat0[0], stat1[1] - stat0[1]
    out["two_stage"] = {"what": "stage 1 scans every query's nearest list and seeds the per-query K-th-best bounds; an exact lower bound per remaining (query, list) pair — "
                                "the serial float32 sum of the pair's table row minima — removes the pairs none of whose candidates can pass; stage 2 scans the rest. "
                                "Results are bit-identical to the every-candidate search.",
                        "pairs_behind_nearest_lists_per_batch": behind / searches_two_stage,
                        "pairs_left_alive_fraction": (alive / behind) if behind else None}
    if ex_med is not None:
        same = bool(np.array_equal(x_cn, g_cn) and all(np.array_equal(x_ids[b, :g_cn[b]], g_ids[b, :g_cn[b]]) and
                                                     np.array_equal(x_sc[b, :g_cn[b]].view(np.uint32), g_sc[b, :g_cn[b]].view(np.uint32)) for b in range(B)))
        out["every_candidate_search"] = {"qps": B / ex_med, "ms_per_step": ex_med * 1e3, "steps": ex_steps, "adc_scan_ms": adc_per_step,
                                         "identical_to_pruned_search": same}
    if not args.no_cpu_baseline and world == 1:
        blob = idx.to_bytes()                                 # the reference's IVPQ on-disk layout (flushes; nothing is soft-deleted)
        cb = cpu_baseline_ivfpq(args, blob, Q_host, K, g_ids, g_sc, g_cn, f_ids)
        out["cpu_baseline"] = cb
        out["recall_at_10_vs_oracle_ivfpq"] = 1.0 - cb["parity_mismatches"] / max(1, cb["parity_checked_queries"])   # identical lists on every sampled query -> 1.0
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    legs = set(args.legs.split(","))
    use_dist = world > 1 or os.environ.get("COMET_BENCH_FORCE_DIST") == "1"   # the env knob exercises the RCCL path at world size 1
    import comet_amd as ca
    ctx = ca.Context(local_rank)
    comm = None
    if use_dist:
        # RCCL lives inside libcomet_hip.so (comet_comm_*): the host only carries the 128-byte id from rank 0 to the others
        from comet_amd.dist import Comm
        comm = Comm.from_env(ctx)

    def barrier():
        ctx.sync()
        if comm is not None:
            comm.barrier()

    def reduce_max(x):
        return comm.allreduce_max(x) if comm is not None else x
    timer = Timer(barrier, reduce_max)

    # ---------------------------------------------------------------- headline: Flat (configs[1])
    idx = ca.FlatIndex(ctx, args.dim, args.metric)
    lo = args.rows * rank // world
    hi = args.rows * (rank + 1) // world
    t0 = time.time()
    add_rows(ctx, idx, lo, hi, args.dim, lambda buf, r0, m: ctx.synth_fill(buf, CORPUS_SEED, r0 * args.dim, m * args.dim))
    ctx.sync()
    build_s = time.time() - t0
    B, K = args.batch, args.k
    q_dev = ctx.alloc(B * args.dim * 4)
    ctx.synth_fill(q_dev, QUERY_SEED, 0, B * args.dim)
    # result-buffer sets: batch i+1 is enqueued before batch i is finalised / exchanged (software pipeline)
    ptrs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(3)]

    def run(nsteps):
        prev = None
        for i in range(nsteps):
            w = i % 3
            if comm is not None:    # shard search now; all-gather + merge of the previous batch follow it on the exchange stream
                t = comm.search_async(idx, q_dev, B, K, ptrs[w][0], ptrs[w][1], ptrs[w][2], K, mode=args.mode)
            else:
                t = idx.search_batch_dev_async(q_dev, B, K, ptrs[w][0], ptrs[w][1], ptrs[w][2], K, mode=args.mode)
            if prev is not None:
                comm.search_wait(idx, prev, block=False) if comm is not None else idx.search_wait(prev)
            prev = t
        if prev is not None:
            comm.search_wait(idx, prev, block=True) if comm is not None else idx.search_wait(prev)

    run(1)
    med, times, prof, allk = timed(ctx, timer, run, args, "flat_scan_f16" if B > 64 else "flat_scan_f16_n64")

    line = None
    if rank == 0:
        total_steps = args.steps * len(times) + max(1, args.warmup)
        line = {
            "metric": "queries/sec + recall@10, 1M x 768 Flat & IVFPQ (value = the Flat leg: exact search, recall@K = 1.0 by construction, ids bit-identical "
                      "to the CPU reference path; the IVFPQ leg with its recall@10 is in `ivfpq`)",
            "value": B * args.steps / med, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": med / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "timing": {"regions": len(times), "reported": "median region", "region_ms": [round(t * 1e3, 3) for t in times]},
            "config": {"workload": f"Flat {args.metric} {args.rows}x{args.dim}, batch={B} queries, K={K} (BASELINE configs[1])",
                       "rows": args.rows, "dim": args.dim, "batch": B, "k": K, "metric": args.metric,
                       "mode": {0: "auto", 1: "strict", 2: "fast"}[args.mode], "sharding": f"rows/{world}", "build_s": round(build_s, 2)},
            "roofline": flat_roofline(prof, total_steps, hi - lo, args.dim, B),
            "kernels_ms_per_step": allk,
            "fast_path": {k: idx.stat(k) for k in ("fast_queries", "strict_queries", "fast_candidates", "fast_expansions", "fast_overflows")},
            "recall_at_10": {"flat": 1.0},
            "scaling_note": None if world == 1 else "strong scaling of the named config: the 1M-row corpus is split over the ranks and every rank searches its shard for the "
                            "same 256 queries; per batch a rank keeps ~0.15 ms that does not shrink with its shard (post stage per shard, query preparation, "
                            "exchange + merge, launch gaps), so the measured single-GPU shard timings predict 1.7x / 2.4x / 3.1x at 2 / 4 / 8 GPUs (DESIGN.md 3.9)",
        }
        if world == 1 and not args.no_cpu_baseline:
            last = ptrs[(args.steps - 1) % 3]
            ids = ctx.download(last[0], (B, K), np.uint32)
            sc = ctx.download(last[1], (B, K), np.float32)
            cn = ctx.download(last[2], (B,), np.int32)
            line["cpu_baseline"] = cpu_baseline_flat(args, ids, sc, cn)
        else:
            line["cpu_baseline"] = None

    # ---------------------------------------------------------------- Flat L2^2 at HBM-binding batches (N = 1) + IVFPQ (any N: list shards)
    if legs & ({"flat_l2", "ivfpq"} if world == 1 else {"ivfpq"}):
        flat2 = ca.FlatIndex(ctx, args.dim, ca.L2_SQUARED)
        add_rows(ctx, flat2, 0, args.rows, args.dim, lambda buf, r0, m: ctx.synth_mixture(buf, MIX_SEED, MIX_CENTERS, MIX_SIGMA, MIX_SUB, MIX_NOISE, r0, m, args.dim))
        # queries: fresh draws from the same mixture (rows past the corpus)
        q2_dev = ctx.alloc(B * args.dim * 4)
        ctx.synth_mixture(q2_dev, MIX_SEED, MIX_CENTERS, MIX_SIGMA, MIX_SUB, MIX_NOISE, args.rows + 7, B, args.dim)
        ctx.sync()
        Q2 = ctx.download(q2_dev, (B, args.dim), np.float32)
        if "flat_l2" in legs and world == 1:
            line["flat_l2"] = leg_flat_l2(ctx, ca, args, timer, flat2, q2_dev)
        if "ivfpq" in legs:
            iv = leg_ivfpq(ctx, ca, args, timer, flat2, q2_dev, Q2, comm, rank, world)
        if "ivfpq" in legs and rank == 0:
            line["ivfpq"] = iv
            line["recall_at_10"]["ivfpq_vs_exact_flat"] = line["ivfpq"]["recall_at_10_vs_exact_flat"]
            if "recall_at_10_vs_oracle_ivfpq" in line["ivfpq"]:
                line["recall_at_10"]["ivfpq_vs_oracle_ivfpq"] = line["ivfpq"]["recall_at_10_vs_oracle_ivfpq"]

    if comm is not None:
        comm.barrier()
        comm.close()
    if rank == 0:
        # RCCL writes its version banner through C stdio: flush that first so that the JSON line is the LAST line of stdout
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
