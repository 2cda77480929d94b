#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: queries/sec + recall@10, 1M x 768 Flat & IVFPQ, on N MI355X — and every other BASELINE config
inside the same driver-run line.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under torch.distributed.run, one rank
per GPU. Rank 0 prints ONE JSON line.

Legs (all inside the one line; `--legs` selects):
  * headline `flat` = BASELINE configs[1]: Flat cosine 1M x 768, batch 256, K 100 (uniform SplitMix64 data). A "step"
    = one batch through the whole search path (preprocess -> scan -> top-K -> ids) with queries and results resident in HBM.
    `value` = queries/s of this leg. N > 1: the SAME 1M rows sharded by contiguous row blocks ("scaling": "strong"), per-shard
    top-K exchanged with one RCCL all-gather inside the library and merged;
  * `flat_l2` (N = 1): Flat L2^2 over a clustered 1M x 768 corpus at B = 1, 64 (HBM binds: the north star's ">= 70 % of the HBM
    roofline on the Flat L2 scan" is read off these) and B = 256;
  * `ivfpq`: IVFPQ 1M x 768, nlist 1024, nprobe 32, M 96, nbits 8, K 10 — the other half of the metric, with recall@10 against exact
    Flat search and against the CPU oracle on the same index; N > 1: inverted lists sharded over the ranks;
  * `ivfpq10m` = configs[3]'s shape on the GPUs present: IVFPQ 10M x 768, nlist 4096, nprobe 32, M 96, nbits 8, K 10, built on the
    GPU inside the run (N > 1: list shards, the sharded form the config names);
  * `hnsw` (N = 1) = configs[2]: HNSW L2 d 384, M 16, efSearch 128, K 10. TIMED on `hnsw_navigable`: a 1M-node (`--hnsw-nav-rows`) layer-0 graph built on the
    GPU inside the run from exact Flat k-NN (16 nearest + 16 random edges per node), searched by the same kernel, 256-query parity against the CPU oracle on that
    graph, recall@10 against exact Flat search. The reference-construction graph (`--hnsw-rows`, default 100k, built by the GPU insert kernel one insertion at a
    time as the reference's semantics demand) is a PARITY line: the reference's insertNode leaves graphs whose searches stop after a few dozen expansions;
  * `hybrid` (N = 1) = configs[4]: IVF 1M x 768 (nlist 1024; nprobe 32 and the hybrid default 1) + BM25 over 100k documents +
    Reciprocal Rank Fusion.
Every leg rotates 8 distinct query batches (no step replays the previous step's queries), reports the median of R timed regions of
EXACTLY K steps (barrier + sync on both sides, max over ranks), a `sustained` figure (>= 2 s of back-to-back steps: the Flat scan is
power-governed, a 10 ms region cannot show the settled clock), a `host_buffers` figure (queries uploaded and results downloaded
inside every call — what the reference's Execute([]float32) shape costs; never `value`), `roofline` of its dominant kernel from HIP
events attached to that kernel's dispatches in the timed regions, and `cpu_baseline` + a bit-for-bit parity count against the CPU
oracle (C++ restatement of the reference's Go loops — no Go toolchain in this image) on a bounded sample of the same workload.
The product path never touches the oracle: it is imported only inside the cpu_baseline / parity helpers.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_ROWS, DIM, BATCH, TOPK = 1_000_000, 768, 256, 100
CORPUS_SEED, QUERY_SEED = 0xC0FFEE + 2, 0xBEEF + 2
# clustered corpus of the L2 / IVFPQ / IVF legs: 2048 centres, 65536 sub-centres at 0.15 around them (~15 rows each at 1M rows), 0.02 noise:
# a query's true neighbours are the rows of its own sub-centre
MIX_SEED, MIX_CENTERS, MIX_SIGMA, MIX_SUB, MIX_NOISE = 0xC0FFEE + 4, 2048, 0.15, 65536, 0.02
HBM_PEAK_GBS = 8000.0
LANES = max(1, min(8, int(os.environ.get("COMET_LANES", "8"))))    # execution lanes of the timed regions (the library's default: 8; Flat / IVF use two of them, PQ / IVFPQ four, HNSW eight)
NQB = 8               # distinct query batches a leg rotates through
DTYPE = ("f32 results: every returned score is the reference's serial float32 sum (bit-identical to the CPU path); candidates are "
         "screened on int8 MFMA (v_mfma_i32_32x32x32_i8, exact int32 accumulate; fp16 v_mfma_f32_32x32x16_f16 where int8 is too coarse for the "
         "data) with a rigorous error bound from measured quantisation residuals, PQ tables / BM25 in f32 / f64")
INT8_PEAK_TOPS = 5000.0   # dense int8 MFMA peak (2 x the 2.5 PF fp16 peak; MI355X_MICROARCH.md: >= 3944 TOPS measured)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=11, help="timed regions of --steps steps each; the median is reported (11: the step time keeps falling over the first ~100 steps "
                    "behind the warm-up — 6.45 5.82 5.46 5.30 5.26 ms per 20 steps in round 6's record, the same slope from the C harness, profiles/r06_c_bench.txt — and the "
                    "median of five regions sat in the middle of that slope; every region's time is in `timing.region_ms`)")
    ap.add_argument("--sustain-s", type=float, default=2.0, help="length of the sustained (back-to-back) measurement of every leg")
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--k", type=int, default=TOPK)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--mode", type=int, default=0, help="0 auto, 1 strict exact kernels, 2 fast path")
    ap.add_argument("--legs", default="flat,c1,flat_l2,ivfpq,ivfpq_uniform,ivfpq10m,hnsw,hybrid", help="comma list; flat is always run (it is the headline)")
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--M", type=int, default=96)
    ap.add_argument("--nbits", type=int, default=8)
    ap.add_argument("--ivfpq-k", type=int, default=10)
    ap.add_argument("--big-rows", type=int, default=10_000_000, help="rows of the ivfpq10m leg (configs[3])")
    ap.add_argument("--big-nlist", type=int, default=4096)
    ap.add_argument("--hnsw-rows", type=int, default=100_000)         # the reference-construction graph (sequential insertion: a parity line)
    ap.add_argument("--hnsw-nav-rows", type=int, default=1_000_000)   # the navigable graph the search kernel is timed on: configs[2]'s stated size
    ap.add_argument("--hnsw-dim", type=int, default=384)
    ap.add_argument("--docs", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--shard-mode", default="auto", choices=("auto", "rows", "replica", "grid"),
                    help="N > 1: rows = one index split over all ranks (strong scaling, RCCL all-gather per batch); replica = every rank holds the whole index and "
                         "serves its own query stream (throughput; no exchange); grid = --shard-group ranks share one index, world / group replicas of that; "
                         "auto = `value` from replica (the 1M index is 0.8-3 GB of a 288 GB part) AND the rows form measured beside it (`sharded_rows`)")
    ap.add_argument("--shard-group", type=int, default=2, help="ranks per index copy in --shard-mode grid (must divide the world size)")
    ap.add_argument("--full-line", action="store_true", help="print the full record (tens of KB) instead of the compact line; the full record is always written to bench_legs.json")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ helpers
def add_rows(ctx, idx, row_lo, row_hi, dim, fill, also=()):
    """Generate rows [row_lo, row_hi) on the device with fill(buf, lo, m) and add them (ids = row + 1) to idx (and to every index in `also`)."""
    from comet_amd._lib import check
    chunk = 65536
    buf = ctx.alloc(chunk * dim * 4)
    idbuf = ctx.alloc(chunk * 4)
    for lo in range(row_lo, row_hi, chunk):
        m = min(chunk, row_hi - lo)
        fill(buf, lo, m)
        ctx.upload(idbuf, np.arange(lo + 1, lo + m + 1, dtype=np.uint32))
        for ix in (idx,) + tuple(also):
            added = C.c_int64()
            check(ctx.lib.comet_index_add_dev(ix.h, C.c_void_p(idbuf), C.c_void_p(buf), m, C.byref(added)))
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

    def sustained(self, step_fn, steps, med_region_s, seconds):
        """Back-to-back steps for >= `seconds` (the step count is derived from the max-reduced region time: the same on every rank)."""
        n = max(steps, int(np.ceil(seconds / max(med_region_s / steps, 1e-7) / steps)) * steps)
        self.barrier()
        t0 = time.perf_counter()
        step_fn(n)
        self.barrier()
        el = self.reduce_max(time.perf_counter() - t0)
        return n, el


def timed(ctx, timer, step, args, dominant):
    """Timed regions with HIP events around the DOMINANT kernel only (every timed scope is two event records — barrier packets —
    and ~10 us of dispatch latency in a chain of short kernels; with all scopes timed a Flat step is 6 % slower), then one
    untimed region with every scope timed for the per-kernel breakdown. The timed regions run as the library runs by default: two
    execution lanes, the two batches in flight on two streams — a kernel's event-to-event duration there includes what it shares
    the GPU with. The breakdown region runs on ONE lane, where a kernel's duration is its own. Returns (median s, region times,
    dominant-kernel profile of the timed regions, {kernel: (total ms, launches)} of the one-lane region)."""
    ctx.profile_only(dominant); ctx.profile(True); ctx.profile_reset()
    med, times = timer.run(step, args.steps, args.warmup, args.regions)
    prof = ctx.profile_dump()
    ctx.profile_only(None); ctx.profile_reset()
    ctx.set_lanes(1)
    step(2)
    ctx.sync(); ctx.profile_reset()
    step(args.steps)
    ctx.sync()
    allk = ctx.profile_dump(); ctx.profile(False)
    ctx.set_lanes(LANES)
    return med, times, prof, allk


def measure(ctx, timer, args, step, dominant, B):
    """The common part of every leg's record: median region, regions, sustained run, dominant-kernel profile, kernel breakdown."""
    med, times, prof, allk = timed(ctx, timer, step, args, dominant)
    n_sus, el_sus = timer.sustained(step, args.steps, med, args.sustain_s)
    rec = {"qps": B * args.steps / med, "ms_per_step": med / args.steps * 1e3, "region_ms": [round(t * 1e3, 3) for t in times],
           "sustained": {"seconds": round(el_sus, 3), "steps": n_sus, "qps": B * n_sus / el_sus, "ms_per_step": el_sus / n_sus * 1e3},
           "execution_lanes": LANES, "batches_in_flight": getattr(getattr(step, "__self__", None), "depth", None),
           "batch_latency_ms_about": (getattr(getattr(step, "__self__", None), "depth", None) or 1) * med / args.steps * 1e3,   # a batch stays in the pipe for ~ depth steps
           "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(allk.items())},
           "kernels_ms_per_step_are": "per-kernel HIP-event durations of a region of their own on ONE execution lane (each kernel alone on the GPU)"}
    # single stream: ONE execution lane, ONE batch in flight (enqueue, wait, enqueue ...) — what a caller that does not pipeline its batches sees
    pipe = getattr(step, "__self__", None)
    if isinstance(pipe, Pipe):
        d0 = pipe.depth
        pipe.depth = 1; ctx.set_lanes(1)
        m1, _t1 = timer.run(step, args.steps, 2, 3)
        pipe.depth = d0; ctx.set_lanes(LANES)
        rec["single_stream"] = {"qps": B * args.steps / m1, "ms_per_step": m1 / args.steps * 1e3, "execution_lanes": 1, "batches_in_flight": 1}
    prof = dict(prof); prof["__one_lane__"] = allk
    return rec, prof, med, times


def one_lane_stats(prof, name):
    """(average ms, launches) of a kernel in the one-lane region: its duration with the GPU to itself"""
    ms, n = (prof.get("__one_lane__") or {}).get(name, (0.0, 0))
    return (ms / n if n else 0.0), n


ROOFLINE_MEASURED = ("HIP events on the kernel's own dispatch in a timed region of this run on ONE execution lane (the kernel alone on the GPU); in the regions "
                     "`value` is measured on, the two batches in flight run on two streams and a kernel's event-to-event duration includes what it shares "
                     "the GPU with: `in_value_regions`")


def in_value_regions(prof, name, alg_bytes):
    avg, n = kernel_stats(prof, name)
    ach = alg_bytes / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
    return {"avg_kernel_ms": avg, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "launches": n, "execution_lanes": LANES}


def host_buffers_qps(fn, B, reps=5):
    """queries/s with the queries uploaded and the results downloaded inside every call (blocking host API)"""
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    el = time.perf_counter() - t0
    return {"qps": B * reps / el, "ms_per_call": el / reps * 1e3, "what": "host numpy queries in, host numpy results out, one blocking call per batch (H2D + search + D2H + sync)"}


def kernel_stats(prof, name):
    ms, n = prof.get(name, (0.0, 0))
    return (ms / n if n else 0.0), n


def source_sha():
    """fingerprint of the kernel sources: a committed PMC profile is only quoted when it was collected from the same code"""
    h = hashlib.sha256()
    for f in sorted((ROOT / "comet_amd" / "csrc").glob("*.h*")):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, rows_local, leg=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass (newest profiles/r*_pmc*.json naming the kernel). The file carries
    the fingerprint of the kernel sources it was measured on (tools/pmc_collect.py); a file measured on other code is NOT quoted."""
    best, stale = None, None
    sha = source_sha()
    for f in sorted((ROOT / "profiles").glob("r*pmc*.json")):
        try:
            pm = json.loads(f.read_text())
            for kname, kv in pm.get("kernels", {}).items():
                names = {"flat_scan_f16": ("flat_scan_q8_kernelILi0ELi128ELb0E", "flat_scan_q8_kernel<0, 128, false>", "flat_scan_f16_kernel"),
                         "flat_scan_i8": ("flat_scan_qr_kernelILi0ELi128ELi6E", "flat_scan_qr_kernel<0, 128, 6", "flat_scan_q8_kernelILi0ELi128ELb1E", "flat_scan_q8_kernel<0, 128, true>"),
                         "flat_scan_f16_n64": ("flat_scan_f16_n64_kernelILi1ELb0E", "flat_scan_f16_n64_kernel<1, false>"),
                         "flat_scan_i8_n64": ("flat_scan_qn_kernelILi1ELi6E", "flat_scan_qn_kernel<1, 6>", "flat_scan_f16_n64_kernelILi1ELb1E", "flat_scan_f16_n64_kernel<1, true>"),
                         "ivf_scan_f16": ("ivf_scan_f16_kernelILi0ELb0E", "ivf_scan_f16_kernel<0, false>"),
                         "ivf_scan_i8": ("ivf_scan_f16_kernelILi0ELb1E", "ivf_scan_f16_kernel<0, true>"),
                         "adc_scan": ("adc_scan_kernel", "adc_scan2_kernel")}.get(kernel, (kernel + "_kernel",))
                if kname.split("<")[0].strip() == kernel or any(nm in kname for nm in names):
                    # a counter value is the measurement of ONE leg: the adc_scan launches of a PMC file belong to the leg the pass ran (tools/pmc_bench.sh: PMC_ADC_LEG,
                    # default the clustered 1M leg) — the uniform and 10M legs quote their own passes or nothing
                    if leg is not None and pm.get("adc_leg", "ivfpq") != leg:
                        stale = f"{f.name} (its adc_scan launches are the {pm.get('adc_leg', 'ivfpq')} leg's, not {leg}'s)"
                        continue
                    if pm.get("source_sha") != sha:
                        stale = f.name
                        continue
                    per = kv.get("hbm_read_bytes_per_launch_corrected", 0) + kv.get("hbm_write_bytes_per_launch_uncalibrated", 0)
                    rows_ref = pm.get("rows", 1_000_000)
                    if rows_ref != rows_local:       # a counter value is a measurement of ONE size: never scaled to another
                        stale = f"{f.name} (measured at {rows_ref} rows, this leg has {rows_local})"
                        continue
                    best = (per, f"profiles/{f.name} (separate rocprofv3 --pmc pass on these kernel sources [{sha}]: FETCH_SIZE x2 gfx950 "
                                                         f"correction + WRITE_SIZE, measured at {rows_ref} rows)")
        except Exception:
            continue
    if best:
        return best
    return (None, f"not quoted: the newest committed PMC pass naming this kernel ({stale}) was measured on other kernel sources or at another size" if stale
            else "not measured in this run (PMC needs its own rocprofv3 pass: tools/pmc_bench.sh)")


FLAT_SCAN_SCOPES = "flat_scan_i8|flat_scan_f16|flat_scan_i8_n64|flat_scan_f16_n64"
IVF_SCAN_SCOPES = "ivf_scan_i8|ivf_scan_f16"


def flat_roofline(prof, rows_local, dim, nq):
    """roofline of the scan kernel the timed regions actually ran: the int8 shadow (1 byte per dimension, rows padded to 256) or the fp16 one"""
    cands = ("flat_scan_i8_n64", "flat_scan_f16_n64") if nq <= 64 else ("flat_scan_i8", "flat_scan_f16")
    name = next((c for c in sorted(cands, key=lambda c: -prof.get(c, (0.0, 0))[1]) if c in prof), "dist_exact")
    avg_ms, n = one_lane_stats(prof, name)
    extra = {}
    if name.startswith("flat_scan"):
        qtile = 64 if name.endswith("n64") else 256
        i8 = "_i8" in name
        ldr, eb = ((dim + 255) // 256 * 256, 1) if i8 else ((dim + 63) // 64 * 64, 2)
        alg = rows_local * ldr * eb + qtile * ldr * eb        # the shadow once + the staged query tile
        flops = 2.0 * qtile * rows_local * ldr
        tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        extra = ({"mfma_tops_int8": tf, "mfma_frac_of_5000": tf / INT8_PEAK_TOPS, "shadow": "int8, one scale per 256-row tile"} if i8
                 else {"mfma_tflops": tf, "mfma_frac_of_2500": tf / 2500.0, "shadow": "fp16"})
    else:
        alg = rows_local * dim * 4                            # exact-arithmetic scan: every fp32 row once (SURVEY 8d N*d*4)
    ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, src = pmc_traffic(name, rows_local)
    return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "avg_kernel_ms": avg_ms, "launches": n,
            "algorithmic_bytes_per_launch": alg, "fp32_rows_bytes_per_pass": rows_local * dim * 4, **extra,
            "measured": ROOFLINE_MEASURED, "in_value_regions": in_value_regions(prof, name, alg)}


def threads_map(fn, n_items, T):
    th = [threading.Thread(target=fn, args=(n_items * t // T, n_items * (t + 1) // T)) for t in range(T)]
    t0 = time.time()
    [t.start() for t in th]
    [t.join() for t in th]
    return time.time() - t0


def same_rows(g_ids, g_sc, g_cn, b, cnt, oi, os_):
    return bool(g_cn[b] == cnt and np.array_equal(g_ids[b, :cnt], oi) and np.array_equal(g_sc[b, :cnt].view(np.uint32), np.asarray(os_, np.float32).view(np.uint32)))


class Pipe:
    """`depth` batches in flight on device-resident buffers (2, or 4 for the index kinds whose searches use four execution lanes), rotating
    NQB query batches: batch i + depth - 1 is enqueued before batch i is waited for."""

    def __init__(self, ctx, idx, q_ptrs, B, K, comm=None, depth=2, **params):
        self.ctx, self.idx, self.q, self.B, self.K, self.comm, self.params = ctx, idx, q_ptrs, B, K, comm, params
        self.depth = max(1, min(int(depth), 8 if comm is None else 3))      # a communicator has four slots; an HNSW index eight pending searches
        self.bufs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(self.depth + 1)]
        self.i = 0

    def step(self, nsteps):
        flying = []
        for _ in range(nsteps):
            w = self.i % len(self.bufs); q = self.q[self.i % len(self.q)]; self.i += 1
            if self.comm is not None:
                t = self.comm.search_async(self.idx, q, self.B, self.K, *self.bufs[w], self.K, **self.params)
            else:
                t = self.idx.search_batch_dev_async(q, self.B, self.K, *self.bufs[w], self.K, **self.params)
            flying.append(t)
            if len(flying) >= self.depth:
                prev = flying.pop(0)
                self.comm.search_wait(self.idx, prev, block=False) if self.comm is not None else self.idx.search_wait(prev)
        for j, prev in enumerate(flying):
            last = j == len(flying) - 1
            self.comm.search_wait(self.idx, prev, block=last) if self.comm is not None else self.idx.search_wait(prev)

    def results_of(self, qi, **override):
        """one blocking search of query batch qi; host copies of (ids, scores, counts)"""
        p = dict(self.params); p.update(override)
        b = self.bufs[0]
        if self.comm is not None:
            self.comm.search_wait(self.idx, self.comm.search_async(self.idx, self.q[qi], self.B, self.K, *b, self.K, **p), block=True); self.comm.sync()
        else:
            self.idx.search_batch_dev(self.q[qi], self.B, self.K, *b, self.K, **p)
        self.ctx.sync()
        return self.ctx.download(b[0], (self.B, self.K), np.uint32), self.ctx.download(b[1], (self.B, self.K), np.float32), self.ctx.download(b[2], (self.B,), np.int32)

    def free(self):
        for b in self.bufs:
            for p in b:
                self.ctx.free(p)


def query_batches(ctx, B, d, fill):
    """NQB device-resident query batches: fill(ptr, batch_index)"""
    base = ctx.alloc(NQB * B * d * 4)
    ptrs = [base + i * B * d * 4 for i in range(NQB)]
    for i, p in enumerate(ptrs):
        fill(p, i)
    ctx.sync()
    return ptrs


# ------------------------------------------------------------------------------------------------ CPU baselines (the only users of the oracle)
def oracle():
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    return orc


def cpu_baseline_flat(args, ids_gpu, scores_gpu, counts_gpu):
    """Oracle Flat search on a bounded sample of query batch 0 vs the full corpus: single-thread latency of one query, then as many
    queries as fit in ~cpu_seconds with one query per host thread; every sampled query is compared bit for bit with the GPU's."""
    orc = oracle()
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
                if not same_rows(ids_gpu, scores_gpu, counts_gpu, qi, n, oi, os_):
                    mismatches.append(qi)
    el = threads_map(worker, cores, cores)
    return {"value": len(done) / el, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{len(done)} of query batch 0's {args.batch} queries vs the full {args.rows}x{args.dim} corpus, {cores} threads (one query each), "
                      f"{el:.1f}s; oracle index build {build_s:.1f}s not timed",
            "single_thread_latency_s": single_s, "single_thread_qps": 1.0 / single_s if single_s > 0 else None,
            "parity_checked_queries": len(done), "parity_mismatches": len(mismatches)}


def cpu_baseline_from_bytes(make_oracle, blob, search, Q, K, g, what, flat_ids=None, max_q=None):
    """The oracle loads the GPU-built index from the reference's on-disk bytes and searches a bounded query sample (one query per
    thread); parity + (flat_ids given) the oracle's own recall vs the exact Flat results."""
    cores = os.cpu_count() or 1
    o = make_oracle()
    t0 = time.time()
    assert o.from_bytes(blob) == len(blob)
    load_s = time.time() - t0
    t0 = time.time()
    search(o, Q[0])
    single_s = time.time() - t0
    nq = min(len(Q), max_q or max(cores, 64))
    bad, lock, orecall = [], threading.Lock(), []

    def work(lo, hi):
        for b in range(lo, hi):
            cnt, ci, cs = search(o, Q[b])
            ok = same_rows(g[0], g[1], g[2], b, cnt, ci, cs)
            with lock:
                if not ok:
                    bad.append(b)
                if flat_ids is not None:
                    orecall.append(len(set(flat_ids[b].tolist()) & set(ci.tolist())) / K)
    T = min(cores, nq)
    el = threads_map(work, nq, T)
    out = {"value": nq / el, "unit": "queries/s", "cores": T, "kind": "port",
           "sample": f"{nq} of query batch 0's queries on the GPU-built index ({what}, loaded from its on-disk bytes in {load_s:.1f}s), {T} threads, {el:.2f}s",
           "single_thread_latency_s": single_s, "parity_checked_queries": nq, "parity_mismatches": len(bad)}
    if flat_ids is not None:
        out["oracle_recall_at_k_vs_exact_flat"] = float(np.mean(orecall))
    return out


def recall_of(f_ids, g_ids, g_cn, K):
    return float(np.mean([len(set(f_ids[b, :K].tolist()) & set(g_ids[b, :g_cn[b]].tolist())) / K for b in range(len(g_cn))]))


def mix_fill(ctx, d, nsub=MIX_SUB):
    return lambda buf, lo, m: ctx.synth_mixture(buf, MIX_SEED, MIX_CENTERS, MIX_SIGMA, nsub, MIX_NOISE, lo, m, d)


# ------------------------------------------------------------------------------------------------ legs
def leg_flat_l2(ctx, ca, args, timer, flat2, q_ptrs):
    """Flat L2^2 at B = 1, 64, 256 on the clustered corpus."""
    out = {"workload": f"Flat l2_squared {args.rows}x{args.dim} (clustered corpus: {MIX_CENTERS} centres, {MIX_SUB} sub-centres at {MIX_SIGMA}, noise {MIX_NOISE}), K={args.ivfpq_k}"}
    K = args.ivfpq_k
    for B in (1, 64, 256):
        pipe = Pipe(ctx, flat2, q_ptrs, B, K)
        pipe.step(2)
        rec, prof, _med, _t = measure(ctx, timer, args, pipe.step, FLAT_SCAN_SCOPES, B)
        rec["roofline"] = flat_roofline(prof, args.rows, args.dim, B)
        out[f"batch{B}"] = rec
        pipe.free()
    return out


def leg_c1(ctx, ca, args):
    """configs[0] as a caller of the reference sees it: Flat L2^2 10 000 x 128, K = 10, ONE query per Execute() — host buffers in,
    host results out (flat_index_search.go:144-150 runs one query per call). Oracle single-thread latency beside the GPU's blocking single-query call."""
    n, d, K, reps = 10_000, 128, 10, 1000
    orc = oracle() if not args.no_cpu_baseline else None
    seedx, seedq = 0xC0FFEE + 1, 0xBEEF + 1
    idx = ca.FlatIndex(ctx, d, ca.L2_SQUARED)
    buf = ctx.alloc(n * d * 4)
    ctx.synth_fill(buf, seedx, 0, n * d)
    X = ctx.download(buf, (n, d), np.float32)
    ctx.free(buf)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    idx.add_batch(ids, X)
    qb = ctx.alloc(reps * d * 4)
    ctx.synth_fill(qb, seedq, 0, reps * d)
    Q = ctx.download(qb, (reps, d), np.float32)
    ctx.free(qb)
    idx.search_batch(Q[:1], K)
    t0 = time.perf_counter()
    g = [idx.search_batch(Q[i:i + 1], K) for i in range(reps)]
    gpu_s = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    gb = idx.search_batch(Q[:256], K)
    gpu_b256_s = time.perf_counter() - t0
    out = {"workload": f"Flat l2_squared {n}x{d}, K={K}, ONE query per call, host buffers (BASELINE configs[0]; {reps} calls)",
           "gpu_single_query_latency_ms": gpu_s * 1e3, "gpu_single_query_qps": 1.0 / gpu_s,
           "gpu_batch256_host_buffers_qps": 256 / gpu_b256_s,
           "what": "comet_index_search with B = 1: upload of the query, the whole search path, download of K results, one host sync — what a Go caller's Execute() costs"}
    if orc:
        o = orc.Flat(d, "l2_squared"); o.add_batch(ids, X)
        o.search(Q[0], K)
        t0 = time.perf_counter()
        res = [o.search(Q[i], K) for i in range(reps)]
        cpu_s = (time.perf_counter() - t0) / reps
        bad = sum(0 if (g[i][2][0] == res[i][0] and np.array_equal(g[i][0][0, :res[i][0]], res[i][1]) and
                        np.array_equal(g[i][1][0, :res[i][0]].view(np.uint32), np.asarray(res[i][2], np.float32).view(np.uint32))) else 1 for i in range(reps))
        bad += sum(0 if (gb[2][i] == res[i][0] and np.array_equal(gb[0][i, :res[i][0]], res[i][1])) else 1 for i in range(256))
        out.update({"cpu_single_thread_latency_ms": cpu_s * 1e3, "cpu_single_thread_qps": 1.0 / cpu_s, "parity_checked_queries": reps + 256, "parity_mismatches": bad,
                    "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port", "sample": f"{reps} single-query searches, one thread",
                                     "parity_checked_queries": reps + 256, "parity_mismatches": bad}})
    idx.close()
    return out


def adc_lookups_ceiling():
    """random 8-byte LDS gather rate of this GPU, measured by tools/lds_gather_probe (built by __graft_entry__.build())"""
    exe = ROOT / "tools" / "lds_gather_probe"
    try:
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:          # the ceiling is context, not a result: the line says why it is missing
        return {"error": f"tools/lds_gather_probe did not run: {e}"}


def leg_ivfpq(ctx, ca, args, timer, flat_exact, q_ptrs, Q0, rows, nlist, tag, comm=None, rank=0, world=1, every_candidate=True, nsub=MIX_SUB, fill=None, corpus="clustered corpus",
              replicas=1, solo=True):
    """world > 1: every rank trains on the same vectors (deterministic GPU k-means: replicated quantisers), owns the inverted lists
    dealt to it (comet_index_set_shard: by list length, identically on every rank) and adds every row (foreign members are dropped); searches go through the in-library
    RCCL exchange, so every rank ends up with the merged global top-K. `replicas`: groups of `world` ranks, each with its own copy of the index and its
    own query stream (qps counts all of them); `solo`: a one-GPU job (the every-candidate pass and the CPU baseline run only there)."""
    B, K, d, n = args.batch, args.ivfpq_k, args.dim, rows
    idx = ca.IVFPQIndex(ctx, d, ca.L2_SQUARED, nlist, args.M, args.nbits)
    ntrain = min(n, nlist * 100)
    fill = fill or mix_fill(ctx, d, nsub)
    tbuf = ctx.alloc(ntrain * d * 4)
    fill(tbuf, 0, ntrain)
    t0 = time.time()
    idx.train_dev(tbuf, ntrain)
    train_s = time.time() - t0
    ctx.free(tbuf)
    if world > 1:
        idx.set_shard(rank, world)
    t0 = time.time()
    add_rows(ctx, idx, 0, n, d, fill)
    ctx.sync()
    add_s = time.time() - t0
    pipe = Pipe(ctx, idx, q_ptrs, B, K, comm, depth=4, nprobes=args.nprobe)     # no kernel of an IVFPQ step fills the GPU: four searches in flight on four lanes
    pipe.step(2)
    # the library's search counters are device atomics (0.1 ms per batch when on): counted in a pass of their own, off in the timed regions
    idx.stat("adc_stats_on"); idx.stat("adc_stats_reset")
    pipe.step(NQB)
    ctx.sync()
    searches = NQB
    stat0, stat1 = (0.0, 0.0), (idx.stat("adc_pairs_alive"), idx.stat("adc_pairs_behind_nearest"))
    norm_kills = idx.stat("adc_norm_bound_kills")
    idx.stat("adc_stats_off")
    rec, prof, med, times = measure(ctx, timer, args, pipe.step, "adc_scan", B * replicas)
    g = pipe.results_of(0)
    out = {"workload": f"IVFPQ l2_squared {n}x{d} ({corpus}), nlist={nlist} nprobe={args.nprobe} M={args.M} nbits={args.nbits}, batch={B}, K={K}"
                       + (f"; inverted lists sharded over {world} ranks" if world > 1 else "") + (f"; {replicas} index copies, one query stream each" if replicas > 1 else ""), **rec,
           "layout": {"ranks_per_index_copy": world, "index_copies": replicas, "queries_per_step_all_ranks": B * replicas},
           "train_vectors": ntrain, "train_s": round(train_s, 2), "add_s": round(add_s, 2)}
    alive, behind = stat1[0] - stat0[0], stat1[1] - stat0[1]
    out["two_stage"] = {"what": "stage 1 scans every query's nearest list and seeds the per-query K-th-best bounds; an exact lower bound per remaining (query, list) pair — "
                                "the serial float32 sum of the pair's table row minima — removes the pairs none of whose candidates can pass; stage 2 scans the rest. "
                                "Results are bit-identical to the every-candidate search.",
                        "pairs_behind_nearest_lists_per_batch": behind / max(1, searches), "pairs_left_alive_fraction": (alive / behind) if behind else None,
                        "pairs_removed_by_the_table_free_bound_fraction": (norm_kills / behind) if behind else None,
                        "table_free_bound": "before any table entry is formed: a candidate's sum is |r - x|^2 >= (|r| - R(list))^2 for the query's residual r and R(list) = the largest "
                                            "decoded-residual norm among the list's members (kept per list); margins cover the float32 rounding of both sides (DESIGN.md 3.5)"}
    if rank == 0 and flat_exact is not None:
        f_ids = flat_exact.search_batch(Q0, K)[0]
        out["recall_at_10_vs_exact_flat"] = recall_of(f_ids, g[0], g[2], K)
    else:
        f_ids = None
    if solo:
        # what the ADC kernel itself sustains: the same search with every candidate scored (mode 1: no lower-bound pruning)
        e_ids, e_lists, _ = idx.export()
        list_len = np.bincount(e_lists, minlength=nlist)
        out.update({"max_list_len": int(list_len.max()), "mean_list_len": float(list_len.mean())})
        if every_candidate:
            ctx.set_lanes(1)                                      # the kernel's own rate: one scan at a time on the GPU
            p1 = Pipe(ctx, idx, q_ptrs, B, K, None, nprobes=args.nprobe, mode=1)
            p1.step(2)
            ex_steps = max(3, args.steps // 2)
            idx.stat("adc_stats_on"); idx.stat("adc_stats_reset")
            p1.step(NQB); ctx.sync()
            ns = max(1.0, idx.stat("adc_searches"))              # per search (= per adc_scan launch here: one stage, one sub-batch)
            cand = idx.stat("adc_candidates") / ns / B
            code_bytes, table_bytes = idx.stat("adc_code_bytes") / ns, idx.stat("adc_table_bytes") / ns
            idx.stat("adc_stats_off")
            ctx.profile_only("adc_scan"); ctx.profile(True); ctx.profile_reset()
            ctx.sync(); t0 = time.perf_counter(); p1.step(ex_steps); ctx.sync(); ex_el = (time.perf_counter() - t0) / ex_steps
            ex_prof = ctx.profile_dump(); ctx.profile_only(None); ctx.profile(False)
            x = p1.results_of(0)
            p1.free()
            ctx.set_lanes(LANES)
            adc_ms = ex_prof.get("adc_scan", (0.0, 0))[0] / ex_steps
            phys = code_bytes + (0.0 if args.nbits == 8 and (d // args.M) in (4, 8) and not os.environ.get("COMET_ADC_STREAM_TABLES") else table_bytes)
            same = bool(np.array_equal(x[2], g[2]) and all(np.array_equal(x[0][b, :g[2][b]], g[0][b, :g[2][b]]) and
                                                         np.array_equal(x[1][b, :g[2][b]].view(np.uint32), g[1][b, :g[2][b]].view(np.uint32)) for b in range(B)))
            ceil_ = adc_lookups_ceiling()
            lookups = cand * B * args.M
            # what has to come from HBM at least: the code words of every DISTINCT probed list once (the duos of a list share them through L2 / the
            # Infinity Cache) + the lookup tables of the scanned pairs once (built by pq_lut into HBM, streamed back by the scan)
            cent = idx.centroids(nlist).astype(np.float64); q64 = Q0.astype(np.float64)
            d2 = (q64 ** 2).sum(1)[:, None] + (cent ** 2).sum(1)[None, :] - 2.0 * (q64 @ cent.T)       # fp64 ranking: enough for a byte count
            probed = np.unique(np.argsort(d2, axis=1, kind="stable")[:, :args.nprobe])
            uniq_code_bytes = float(list_len[probed].sum()) * ((args.M + 3) // 4 * 4)
            pair_tables = float((list_len[np.argsort(d2, axis=1, kind="stable")[:, :args.nprobe]] > 0).sum()) * args.M * min(1 << args.nbits, 256) * 4
            # round 5: 8-bit codebooks with 4 / 8 dimensions per subspace build their tables in LDS (adc_scan_kernel<DSUB>): no table byte touches HBM
            tables_in_lds = args.nbits == 8 and (d // args.M) in (4, 8) and not os.environ.get("COMET_ADC_STREAM_TABLES")
            if tables_in_lds:
                pair_tables = 0.0
            must = uniq_code_bytes + pair_tables
            ach = must / (adc_ms * 1e-3) / 1e9 if adc_ms > 0 else 0.0
            traffic, src = pmc_traffic("adc_scan", n, leg=tag)
            out["every_candidate_search"] = {"qps": B / ex_el, "ms_per_step": ex_el * 1e3, "steps": ex_steps, "adc_scan_ms": adc_ms, "identical_to_pruned_search": same,
                                             "candidates_per_query": cand}
            out["roofline"] = {"bound": "hbm", "kernel": "adc_scan", "measured_on": "the every-candidate search (mode 1) on one execution lane: the pruned search's launches see only what the lower bound left, and two lanes would overlap two scans",
                               "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                               "avg_kernel_ms": adc_ms,
                               "algorithmic_bytes_per_launch": must,
                               "algorithmic_bytes_are": "what must come from HBM: the code words of every distinct probed list once" + (" — the lookup tables are built in LDS by the scanning "
                                                        "workgroups and never touch HBM (what `traffic` holds beyond the code words: duos of one list on different XCDs re-reading its codes, and the part of "
                                                        "the codebook slices, re-read per item, that misses L2)" if tables_in_lds else " + one lookup table per scanned (query, list) pair (built into HBM by pq_lut, streamed back once)"),
                               "tables": "built in LDS (adc_scan_kernel<DSUB>)" if tables_in_lds else "streamed through HBM (pq_lut_kernel)",
                               "unique_code_bytes": uniq_code_bytes, "pair_table_bytes": pair_tables,
                               "requested_bytes_per_launch": phys,
                               "requested_bytes_are": "what the launch's work items ask the memory system for: every item's code words (a list's codes once per PAIR of queries sharing it) "
                                                      "+ (streamed tables only) its duo's table per item — counted by the kernel that lays out the work (get_stat adc_code_bytes / adc_table_bytes); "
                                                      "the difference to the HBM bytes is served by L2 and the 256 MB Infinity Cache (the 1M index's codes are 96 MB)",
                               "requested_code_bytes": code_bytes, "requested_table_bytes": table_bytes,
                               "requested_GBps": phys / (adc_ms * 1e-3) / 1e9 if adc_ms > 0 else 0.0,
                               "binding_resource": "LDS gathers: see lds_frac_of_measured_ceiling",
                               "lds_lookups_per_s": lookups / (adc_ms * 1e-3) if adc_ms > 0 else 0.0,
                               "lds_gather_ceiling": ceil_,
                               "lds_frac_of_measured_ceiling": (lookups / (adc_ms * 1e-3) / ceil_["lookups_per_s_two_queries_per_gather"])
                               if adc_ms > 0 and isinstance(ceil_, dict) and ceil_.get("lookups_per_s_two_queries_per_gather") else None,
                               "note": "the scan is bound by LDS gathers (one ds_read_b64 of the interleaved table serves the two queries of a duo), not by HBM: the HBM fraction says how "
                                       "little of the memory system the gathers leave used; the LDS fraction is lookups the reference would do per second over the measured random-gather "
                                       "rate of this chip (lookups of candidates pruned inside the scan are counted although never issued, so it can exceed the issue-rate ceiling)"}
        if not args.no_cpu_baseline:
            orc = oracle()
            blob = idx.to_bytes()                                 # the reference's IVPQ on-disk layout (flushes; nothing is soft-deleted)
            cb = cpu_baseline_from_bytes(lambda: orc.IVFPQ(d, "l2_squared", nlist, args.M, args.nbits), blob,
                                         lambda o, q: o.search(q, K, args.nprobe, cap=K), Q0, K, g, "IVPQ", f_ids)
            out["cpu_baseline"] = cb
            out["recall_at_10_vs_oracle_ivfpq"] = 1.0 - cb["parity_mismatches"] / max(1, cb["parity_checked_queries"])   # identical lists on every sampled query -> 1.0
    pipe.free()
    out["_g0"] = g                 # results of query batch 0 (numpy; main() compares layouts with it and drops it before the record is written)
    return out


def leg_hnsw(ctx, ca, args, timer):
    """configs[2]: HNSW L2, M 16, efConstruction 200, efSearch 128, K 10; graph built by the GPU insert kernel (the reference's sequential
    semantics: one insertion at a time), searched at B = 256 and in a batch sweep; recall@10 vs exact Flat; oracle on the same graph."""
    n, d, K, B = args.hnsw_rows, args.hnsw_dim, 10, args.batch
    M, efc, efs = 16, 200, 128
    g = ca.HNSWIndex(ctx, d, ca.EUCLIDEAN, M, efc, efs)
    flat = ca.FlatIndex(ctx, d, ca.EUCLIDEAN)
    g.set_level_seed(7)
    # clustered rows (the mixture of the other legs at this dimension, ~15 rows per sub-centre): on i.i.d. uniform 384-d data every row is nearly
    # equidistant from every query and recall measures nothing
    nsub = max(64, n // 15)
    fill = lambda buf, lo, m: ctx.synth_mixture(buf, MIX_SEED + 1, MIX_CENTERS, MIX_SIGMA, nsub, MIX_NOISE, lo, m, d)
    t0 = time.time()
    add_rows(ctx, g, 0, n, d, fill, also=(flat,))
    ctx.sync()
    build_s = time.time() - t0
    q_ptrs = query_batches(ctx, 8192, d, lambda p, i: ctx.synth_mixture(p, MIX_SEED + 1, MIX_CENTERS, MIX_SIGMA, nsub, MIX_NOISE, n + 7 + i * 8192, 8192, d))   # 8 x 8192 fresh draws (the sweep's largest batch)
    Q0 = ctx.download(q_ptrs[0], (B, d), np.float32)
    params = dict(ef_search=efs)
    pipe = Pipe(ctx, g, q_ptrs, B, K, None, depth=min(8, LANES), **params)     # a 256-query search is 256 waves: eight of them side by side
    pipe.step(2)
    rec, prof, med, times = measure(ctx, timer, args, pipe.step, "hnsw_search", B)
    gr = pipe.results_of(0)
    evals, exps = g.stat("hnsw_distance_evals"), g.stat("hnsw_expansions")          # counted by the kernel, per search (here: query batch 0)
    pipe.free()
    avg_ms, nl = kernel_stats(prof, "hnsw_search")
    alg = evals * d * 4 + exps * 2 * M * 4            # distance evaluations read a row each; every expansion reads one (layer-0: 2M slots) edge list
    ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    f_ids = flat.search_batch(Q0, K)[0]
    out = {"workload": f"HNSW l2 {n}x{d}, M={M} efConstruction={efc} efSearch={efs}, batch={B}, K={K} (BASELINE configs[2] at {n} rows: the graph is built inside the run by "
                       f"the GPU insert kernel, one insertion at a time as the reference's semantics demand — {build_s:.0f}s; a 1M-node build is minutes, its line is in profiles/)",
           **rec, "rows": n, "build_s": round(build_s, 1), "inserts_per_s": n / build_s,
           "distance_evals_per_query": evals / B, "expansions_per_query": exps / B,
           "recall_at_10_vs_exact_flat": recall_of(f_ids, gr[0], gr[2], K),
           "recall_note": "the reference's insertNode never promotes the entry point and prunes before the new node is linked (DESIGN.md 1), so its graphs recall poorly by "
                          "construction; parity means identical results on the identical graph",
           "roofline": {"bound": "hbm", "kernel": "hnsw_search", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "avg_kernel_ms": avg_ms, "launches": nl, "algorithmic_bytes_per_launch": alg,
                        "algorithmic_bytes_are": "distance evaluations x d x 4 + expansions x 2M x 4, both counted by the kernel (random 1.5 KB row reads, one wave per query)",
                        "note": "the kernel is instruction-issue-bound, not memory-bound (one wave per SIMD issues an instruction per ~8 clocks; DESIGN.md 3.8, profiles/r05_hnsw_trace.txt): the HBM fraction is informational"}}
    sweep = {}
    for Bs in (1024, 4096, 8192):
        p2 = Pipe(ctx, g, q_ptrs, Bs, K, None, depth=4, **params)
        p2.step(2)
        ctx.sync(); t0 = time.perf_counter(); p2.step(6); ctx.sync(); el = (time.perf_counter() - t0) / 6
        sweep[f"batch{Bs}"] = {"qps": Bs / el, "ms_per_step": el * 1e3}
        p2.free()
    out["batch_sweep"] = sweep
    out["host_buffers"] = host_buffers_qps(lambda: g.search_batch(Q0, K, ef_search=efs), B)
    if not args.no_cpu_baseline:
        orc = oracle()
        blob = g.to_bytes()
        out["cpu_baseline"] = cpu_baseline_from_bytes(lambda: orc.HNSW(d, "l2", M, efc, efs), blob, lambda o, q: o.search(q, K, efs), Q0, K, gr, "HNSW")
    # ---- the same kernel on a NAVIGABLE graph (clearly not the reference's build): layer 0 only, every node linked to its 16 exact nearest
    # neighbours (Flat search on the GPU) + 16 random nodes, loaded through comet_hnsw_load_graph — the reference's own insertNode leaves a graph
    # whose searches stop after ~33 expansions; here a search runs its ~efSearch expansions, which is what the kernel is for ----
    def navigable():
        nn = args.hnsw_nav_rows
        nsub_n = max(64, nn // 15)
        fill_n = lambda buf, lo, m: ctx.synth_mixture(buf, MIX_SEED + 1, MIX_CENTERS, MIX_SIGMA, nsub_n, MIX_NOISE, lo, m, d)
        buf = ctx.alloc(nn * d * 4); fill_n(buf, 0, nn); Xh = ctx.download(buf, (nn, d), np.float32); ctx.free(buf)
        fl = ca.FlatIndex(ctx, d, ca.EUCLIDEAN)
        ids_n = np.arange(1, nn + 1, dtype=np.uint32)
        fl.add_batch(ids_n, Xh)
        t0 = time.time()
        knn = np.zeros((nn, 17), np.uint32)
        for lo in range(0, nn, 4096):
            knn[lo:lo + 4096] = fl.search_batch(Xh[lo:lo + 4096], 17)[0]
        fl.close()
        rng = np.random.default_rng(5)
        edges = np.concatenate([knn[:, 1:17], rng.integers(1, nn + 1, (nn, 16), dtype=np.uint32)], axis=1)      # node IDS; a self / duplicate edge is harmless (visited set)
        gn = ca.HNSWIndex(ctx, d, ca.EUCLIDEAN, M, efc, efs)
        gn.load_graph(ids_n, np.zeros(nn, np.int32), Xh, np.arange(0, (nn + 1) * 32, 32, dtype=np.int64), edges.reshape(-1), 1, 0)
        build_nav_s = time.time() - t0
        qn_ = query_batches(ctx, B, d, lambda p, i: ctx.synth_mixture(p, MIX_SEED + 1, MIX_CENTERS, MIX_SIGMA, nsub_n, MIX_NOISE, nn + 7 + i * B, B, d))
        Qn = ctx.download(qn_[0], (B, d), np.float32)
        pn = Pipe(ctx, gn, qn_, B, K, None, depth=min(8, LANES), **params)      # one wave per query: 8 x 256 queries in flight = two waves per SIMD
        pn.step(2)
        recn, profn, _m, _t = measure(ctx, timer, args, pn.step, "hnsw_search", B)
        grn = pn.results_of(0)
        ev, ex = gn.stat("hnsw_distance_evals"), gn.stat("hnsw_expansions")
        pn.free()
        a_ms, a_n = one_lane_stats(profn, "hnsw_search")
        algn = ev * d * 4 + ex * 2 * M * 4
        achn = algn / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
        fl2 = ca.FlatIndex(ctx, d, ca.EUCLIDEAN); fl2.add_batch(ids_n, Xh)
        f2 = fl2.search_batch(Qn, K)[0]; fl2.close()
        r = {"workload": f"HNSW l2 {nn}x{d} on a NAVIGABLE layer-0 graph (16 exact nearest neighbours + 16 random edges per node, built in {build_nav_s:.1f}s on the GPU, loaded through "
                         f"comet_hnsw_load_graph): NOT the reference's construction — shown to time the search kernel at ~efSearch expansions", **recn,
             "rows": nn, "graph_build_s": round(build_nav_s, 1),
             "distance_evals_per_query": ev / B, "expansions_per_query": ex / B, "recall_at_10_vs_exact_flat": recall_of(f2, grn[0], grn[2], K),
             "roofline": {"bound": "hbm", "kernel": "hnsw_search", "achieved": achn, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achn / HBM_PEAK_GBS, "traffic": None,
                          "avg_kernel_ms": a_ms, "launches": a_n, "algorithmic_bytes_per_launch": algn}}
        if not args.no_cpu_baseline:
            orc = oracle()
            r["cpu_baseline"] = cpu_baseline_from_bytes(lambda: orc.HNSW(d, "l2", M, efc, efs), gn.to_bytes(), lambda o, q: o.search(q, K, efs), Qn, K, grn, "HNSW")
        gn.close(); ctx.free(qn_[0])
        return r
    try:
        out["navigable"] = navigable()
    except Exception as e:      # noqa: BLE001
        import traceback
        out["navigable"] = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc(limit=3)}
    ctx.free(q_ptrs[0])
    return out


def leg_hybrid(ctx, ca, args, timer, q_ptrs, Q0):
    """configs[4]: IVF 1M x 768 + BM25 over 100k documents + Reciprocal Rank Fusion (K = 60), k = 10."""
    from comet_amd.hybrid import reciprocal_rank_fusion
    B, K, d, n, nlist = args.batch, 10, args.dim, args.rows, args.nlist
    ivf = ca.IVFIndex(ctx, d, nlist, ca.COSINE)
    ntrain = min(n, nlist * 100)
    tbuf = ctx.alloc(ntrain * d * 4)
    mix_fill(ctx, d)(tbuf, 0, ntrain)
    t0 = time.time(); ivf.train_dev(tbuf, ntrain); train_s = time.time() - t0
    ctx.free(tbuf)
    t0 = time.time(); add_rows(ctx, ivf, 0, n, d, mix_fill(ctx, d)); ctx.sync(); add_s = time.time() - t0
    out = {"workload": f"Hybrid (BASELINE configs[4]): IVF cosine {n}x{d} nlist={nlist} (clustered corpus; GPU train {train_s:.1f}s, add {add_s:.1f}s) + BM25 over {args.docs} documents "
                       f"+ Reciprocal Rank Fusion (K=60), batch={B}, k={K}"}
    ldh = (d + 63) // 64 * 64
    vec_last = None
    for npb in (args.nprobe, 1):
        pipe = Pipe(ctx, ivf, q_ptrs, B, K, None, nprobes=npb)
        pipe.step(2)
        rec, prof, med, times = measure(ctx, timer, args, pipe.step, IVF_SCAN_SCOPES, B)
        g = pipe.results_of(0)
        scan_rows = ivf.stat("ivf_scan_rows")
        kname = "ivf_scan_i8" if prof.get("ivf_scan_i8", (0.0, 0))[1] >= prof.get("ivf_scan_f16", (0.0, 0))[1] and "ivf_scan_i8" in prof else "ivf_scan_f16"
        avg_ms, nl = one_lane_stats(prof, kname)
        row_bytes = (d + 127) // 128 * 128 if kname == "ivf_scan_i8" else ldh * 2
        alg = scan_rows * row_bytes
        ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, src = pmc_traffic(kname, n)
        rec["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": src, "avg_kernel_ms": avg_ms, "launches": nl, "algorithmic_bytes_per_launch": alg,
                           "algorithmic_bytes_are": ("int8" if kname == "ivf_scan_i8" else "fp16") + " shadow rows of every probed list, once per group of <= 64 of its "
                                                    f"queries (counted by the kernel that lays out the work: {int(scan_rows)} rows of query batch 0) x {row_bytes} bytes",
                           "measured": ROOFLINE_MEASURED, "in_value_regions": in_value_regions(prof, kname, alg)}
        rec["fast_path"] = {k: ivf.stat(k) for k in ("fast_queries", "strict_queries", "fast_candidates", "fast_overflows", "i8_slices", "i8_backoffs")}
        x = pipe.results_of(0, mode=1)                                                     # the exact kernels (search mode 1) on the same batch
        rec["identical_to_exact_kernels"] = bool(np.array_equal(x[2], g[2]) and np.array_equal(x[0], g[0]) and np.array_equal(x[1].view(np.uint32), g[1].view(np.uint32)))
        p1 = Pipe(ctx, ivf, q_ptrs, B, K, None, nprobes=npb, mode=1)
        p1.step(1); ctx.sync(); t0 = time.perf_counter(); p1.step(4); ctx.sync()
        rec["exact_kernels_ms_per_step"] = (time.perf_counter() - t0) / 4 * 1e3
        p1.free()
        rec["host_buffers"] = host_buffers_qps(lambda npb=npb: ivf.search_batch(Q0, K, nprobes=npb), B)
        out[f"ivf_nprobe{npb}"] = rec
        vec_last = g if npb == 1 else vec_last
        if npb == args.nprobe:
            vec32 = g
        pipe.free()
    # ---- BM25: 100k documents of Zipf(1.1) token ids over a 50k vocabulary, 64-256 tokens, 3-term queries ----
    orc = oracle() if not args.no_cpu_baseline else None
    rng = np.random.default_rng(3)
    vocab, nd = 50_000, args.docs
    zipf = lambda size: np.minimum(vocab - 1, (rng.pareto(1.1, size) * 20).astype(np.int64)).astype(np.uint32)
    lens = rng.integers(64, 257, nd)
    gt = ca.BM25SearchIndex(ctx)
    ot = orc.BM25() if orc else None
    t0 = time.time()
    for i, l in enumerate(lens):
        t = zipf(int(l))
        gt.add(i + 1, t)
        if ot:
            ot.add(i + 1, t)
    tbuild = time.time() - t0
    qsets = [[zipf(3).tolist() for _ in range(B)] for _ in range(NQB)]
    gt.search_batch(qsets[0], K)
    it = [0]

    def tstep(nsteps):
        for _ in range(nsteps):
            gt.search_batch(qsets[it[0] % NQB], K); it[0] += 1
    trec, tprof, tmed, _t = measure(ctx, timer, args, tstep, "bm25_score", B)
    tr = gt.search_batch(qsets[0], K)
    trec["roofline"] = {"bound": "hbm", "kernel": "bm25 (per-token scoring launches + top-K)", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "note": "SURVEY 8d: sum over terms of df x 12 bytes + n_docs x 8 is microseconds of traffic at 100k documents — the leg is launch- and latency-bound, a "
                                "roofline fraction is not meaningful and is not reported"}
    if ot:
        nq = min(B, os.cpu_count() or 1)
        bad = []

        def work(lo, hi):
            for b in range(lo, hi):
                c, oi, _, os64 = ot.search(qsets[0][b], K)
                if not (tr[3][b] == c and np.array_equal(tr[0][b, :c], oi) and np.array_equal(tr[2][b, :c].view(np.uint64), np.asarray(os64, np.float64).view(np.uint64))):
                    bad.append(b)
        cel = threads_map(work, nq, nq)
        trec["cpu_baseline"] = {"value": nq / cel, "unit": "queries/s", "cores": nq, "kind": "port", "sample": f"{nq} of query set 0's 3-term queries over the {nd} documents, {nq} threads, {cel:.2f}s",
                                "parity_checked_queries": nq, "parity_mismatches": len(bad), "parity_is": "ids and float64 score bit patterns"}
    trec["workload"] = f"BM25 over {nd} documents (Zipf(1.1) token ids, vocabulary {vocab}, 64-256 tokens), batch={B} 3-term queries, K={K} (host-side token lists in, host results out; both indexes built in {tbuild:.0f}s)"
    out["bm25"] = trec
    # ---- fusion: the reference cuts both legs to k before fusing (hybrid_search_index.go:518,555) and ranks on the host: one vectorised call for
    # the batch (comet_amd.hybrid.reciprocal_rank_fusion_batch; checked against the per-query restatement of fusion.go:174-243) ----
    from comet_amd.hybrid import reciprocal_rank_fusion_batch
    e2e = {}
    for tag, vec in (("nprobe1", vec_last), (f"nprobe{args.nprobe}", vec32)):
        reciprocal_rank_fusion_batch(vec[0], vec[2], tr[0], tr[3], K)            # (the first call pays numpy's one-time set-up: 0.5 ms against 0.18)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            f_ids, f_sc, f_cn = reciprocal_rank_fusion_batch(vec[0], vec[2], tr[0], tr[3], K)
            ts.append((time.perf_counter() - t0) * 1e3)
        fuse_ms = sorted(ts)[2]                                                    # median of five warm calls: what a serving loop pays per batch
        vleg = out["ivf_nprobe1" if tag == "nprobe1" else f"ivf_nprobe{args.nprobe}"]
        e2e_ms = vleg["ms_per_step"] + trec["ms_per_step"] + fuse_ms
        e2e[tag] = {"qps": B / (e2e_ms * 1e-3), "ms_per_batch": e2e_ms, "fusion_host_ms_per_batch": fuse_ms}
    # parity of the fused lists on a sample: the per-query dict form
    bad_f = 0
    f_ids, f_sc, f_cn = reciprocal_rank_fusion_batch(vec_last[0], vec_last[2], tr[0], tr[3], K)
    for b in range(min(B, 64)):
        v = {int(i): float(s) for i, s in zip(vec_last[0][b, :vec_last[2][b]], vec_last[1][b, :vec_last[2][b]])}
        t = {int(i): float(s) for i, s in zip(tr[0][b, :tr[3][b]], tr[2][b, :tr[3][b]])}
        want = sorted(reciprocal_rank_fusion(v, t).items(), key=lambda kv: (-kv[1]))[:K]
        if [int(x) for x in f_ids[b, :f_cn[b]]] != [d for d, _ in want]:
            bad_f += 1
    out["rrf"] = {"host_ms_per_batch": e2e["nprobe1"]["fusion_host_ms_per_batch"], "batch_form_mismatches_vs_per_query_form": bad_f,
                  "what": "reciprocalRankFusion.Combine + sort + cut (fusion.go:174-243) on the host for the whole batch in one vectorised call (O(k^2) per query)"}
    out["legs_added"] = dict(e2e["nprobe1"], what="round 4's figure: vector leg (nprobe 1) + text leg + host fusion, their times ADDED (run one after the other)")
    out["legs_added_nprobe32"] = dict(e2e[f"nprobe{args.nprobe}"], what=f"the same with the vector leg at nprobe {args.nprobe}")
    # ---- the whole hybrid search as ONE call (comet_hybrid_rrf_search): host queries + token lists in, fused lists out; the vector leg on a second lane beside
    # the text leg, the fusion on the device (rrf_fuse_kernel). Timed as a caller sees it: blocking calls, rotating query batches. ----
    from comet_amd.hybrid import hybrid_rrf_search_batch
    Qh = [ctx.download(q_ptrs[i], (B, d), np.float32) for i in range(NQB)]
    # token ids as the C ABI takes them (flat uint32 + offsets): what the Go shim hands over after tokenising; the Python list form costs 0.1 ms of flattening per batch
    flat = [(np.ascontiguousarray([t for qt in qs for t in qt], np.uint32), np.concatenate([[0], np.cumsum([len(qt) for qt in qs])]).astype(np.int32)) for qs in qsets]
    for tag, npb in (("end_to_end", 1), ("end_to_end_nprobe32", args.nprobe)):
        hybrid_rrf_search_batch(ivf, gt, Qh[0], flat[0], k=K, n_probes=npb)
        reps = max(10, args.steps)
        ctx.sync(); t0 = time.perf_counter()
        for i in range(reps):
            hybrid_rrf_search_batch(ivf, gt, Qh[i % NQB], flat[i % NQB], k=K, n_probes=npb)
        el = (time.perf_counter() - t0) / reps
        di, ds, dc = hybrid_rrf_search_batch(ivf, gt, Qh[0], flat[0], k=K, n_probes=npb)
        vec = vec_last if npb == 1 else vec32
        hi_, hs_, hc_ = reciprocal_rank_fusion_batch(vec[0], vec[2], tr[0], tr[3], K)
        both = (np.asarray(vec[2]) > 0) & (np.asarray(tr[3]) > 0)         # (a query with an empty leg keeps the other leg's own scores in the reference: compared in tests/test_hybrid.py)
        bad = int(sum(1 for b in range(B) if both[b] and not (dc[b] == hc_[b] and np.array_equal(ds[b, :dc[b]], hs_[b, :dc[b]]) and
                                                             set(di[b, :dc[b]][ds[b, :dc[b]] > ds[b, dc[b] - 1]].tolist()) == set(hi_[b, :dc[b]][hs_[b, :dc[b]] > hs_[b, dc[b] - 1]].tolist()))))
        out[tag] = {"qps": B / el, "ms_per_batch": el * 1e3, "fusion": "on the device (rrf_fuse_kernel, one wave per query)", "mismatches_vs_host_fusion_of_the_two_legs": bad,
                    "queries_compared": int(both.sum()),
                    "what": f"ONE blocking call per batch: {B} host queries + token lists in, fused top-{K} out; vector leg (nprobe {npb}) on a second execution lane beside the text leg"}
    if orc:
        blob = ivf.to_bytes()
        out["cpu_baseline"] = cpu_baseline_from_bytes(lambda: orc.IVF(d, "cosine", nlist), blob, lambda o, q: o.search(q, K, args.nprobe, cap=K), Q0, K, vec32, "IVFX", None)
        out["cpu_baseline"]["covers"] = f"the IVF leg at nprobe {args.nprobe}; the BM25 leg's baseline is in `bm25`"
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here, one process per GPU, exactly as
    `torch.distributed.run --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); rank 0 inherits stdout and
    prints the one line. Returns the exit status (the first failing rank's; the others are stopped by PID)."""
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), COMET_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p_ in list(live):
            c = p_.poll()
            if c is None:
                continue
            live.remove(p_)
            if c != 0 and rc == 0:
                rc = c
                for q_ in live:          # a rank died: its peers would wait in a collective for ever
                    q_.terminate()
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    legs = set(args.legs.split(","))
    use_dist = world > 1 or os.environ.get("COMET_BENCH_FORCE_DIST") == "1"   # the env knob exercises the RCCL path at world size 1
    # the layout of the job: `R` ranks share one copy of an index (row / list shards + the in-library exchange), `world / R` such groups serve
    # their own query streams. rows: R = world; replica: R = 1; grid: R = --shard-group. auto: replica for `value`, rows measured beside it.
    mode = args.shard_mode if world > 1 else "replica"
    R_head = {"rows": world, "replica": 1, "grid": args.shard_group, "auto": 1}[mode]
    if world % R_head:
        raise SystemExit(f"--shard-group {R_head} does not divide the world size {world}")
    also_rows = world > 1 and mode == "auto"
    import comet_amd as ca
    # COMET_BENCH_DEVICE: every rank on one device — only for exercising the multi-rank code path on a one-GPU box (with COMET_RCCL_LIB)
    ctx = ca.Context(int(os.environ.get("COMET_BENCH_DEVICE", local_rank)))
    wcomm = None
    comms = {}
    if use_dist:
        # RCCL lives inside libcomet_hip.so (comet_comm_*): the host only carries the 128-byte id from rank 0 to the others
        from comet_amd.dist import Comm
        wcomm = Comm.from_env(ctx)

    def group_comm(R):
        """the communicator of this rank's group of R ranks (None: no exchange; the world communicator when the group is the world)"""
        if R <= 1 or wcomm is None:
            return wcomm if (R == 1 and world == 1 and wcomm is not None) else None      # COMET_BENCH_FORCE_DIST: a world-1 communicator
        if R == world:
            return wcomm
        if R not in comms:
            from comet_amd.dist import Comm
            base = int(os.environ.get("COMET_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29540")) + 101))
            comms[R] = Comm(ctx, rank % R, R, os.environ.get("MASTER_ADDR", "127.0.0.1"), base + 1 + rank // R)
        return comms[R]

    def barrier():
        ctx.sync()
        if wcomm is not None:
            wcomm.barrier()

    def reduce_max(x):
        return wcomm.allreduce_max(x) if wcomm is not None else x
    timer = Timer(barrier, reduce_max)
    t_start = time.time()
    B, K = args.batch, args.k

    # ---------------------------------------------------------------- headline: Flat (configs[1])
    def flat_pass(R):
        """one Flat measurement with R ranks per index copy: (record, profile, times, results of query batch 0, fast-path counters, build s, local rows)"""
        Qn, g, r = world // R, rank // R, rank % R
        gc = group_comm(R)
        idx = ca.FlatIndex(ctx, args.dim, args.metric)
        lo, hi = args.rows * r // R, args.rows * (r + 1) // R
        t0 = time.time()
        add_rows(ctx, idx, lo, hi, args.dim, lambda buf, r0, m: ctx.synth_fill(buf, CORPUS_SEED, r0 * args.dim, m * args.dim))
        ctx.sync()
        build_s = time.time() - t0
        # group 0's query batch 0 is the stream the CPU baseline regenerates (QUERY_SEED from offset 0); the other batches and groups continue it
        q_ptrs = query_batches(ctx, B, args.dim, lambda p, i: ctx.synth_fill(p, QUERY_SEED, (g * NQB + i) * B * args.dim, B * args.dim))
        pipe = Pipe(ctx, idx, q_ptrs, B, K, gc, mode=args.mode)
        pipe.step(1)
        rec, prof, _med, times = measure(ctx, timer, args, pipe.step, FLAT_SCAN_SCOPES, B * Qn)
        g0 = pipe.results_of(0)
        stats = {k: idx.stat(k) for k in ("fast_queries", "strict_queries", "fast_candidates", "fast_expansions", "fast_overflows", "i8_slices", "i8_backoffs", "i8_max_residual")}
        hb = None
        if world == 1:
            Qh = ctx.download(q_ptrs[0], (B, args.dim), np.float32)
            hb = host_buffers_qps(lambda: idx.search_batch(Qh, K, mode=args.mode), B)
        pipe.free(); idx.close(); ctx.free(q_ptrs[0])
        return {"rec": rec, "prof": prof, "times": times, "g0": g0, "stats": stats, "build_s": build_s, "rows_local": hi - lo, "host_buffers": hb,
                "layout": {"ranks_per_index_copy": R, "index_copies": Qn, "queries_per_step_all_ranks": B * Qn,
                           "exchange": None if R == 1 else "one RCCL all-gather of the packed per-shard top-K blocks per batch + merge_topk_kernel (comm.hip)"}}

    hp = flat_pass(R_head)
    rec, prof, times, g0 = hp["rec"], hp["prof"], hp["times"], hp["g0"]
    sharding = (f"rows/{world}" if R_head == world and world > 1 else "none (1 GPU)" if world == 1 else
                f"{world} replicas of the whole index, one query stream each (no exchange)" if R_head == 1 else f"rows/{R_head} x {world // R_head} replicas")
    line = None
    if rank == 0:
        line = {
            "metric": "queries/sec + recall@10, 1M x 768 Flat & IVFPQ (value = the Flat leg: exact search, recall@K = 1.0 by construction, ids bit-identical "
                      "to the CPU reference path; the IVFPQ leg with its recall@10 is in `ivfpq`; the other BASELINE configs are in `ivfpq10m`, `hnsw`, `hybrid`)",
            "value": rec["qps"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong" if (R_head == world and world > 1) else "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "timing": {"regions": len(times), "reported": "median region", "region_ms": rec["region_ms"], "query_batches_rotated": NQB},
            "sustained": rec["sustained"], "single_stream": rec.get("single_stream"),
            "execution_lanes": rec["execution_lanes"], "batches_in_flight": rec["batches_in_flight"], "batch_latency_ms_about": rec["batch_latency_ms_about"],
            "config": {"workload": f"Flat {args.metric} {args.rows}x{args.dim}, batch={B} queries, K={K} (BASELINE configs[1])",
                       "rows": args.rows, "dim": args.dim, "batch": B, "k": K, "metric": args.metric,
                       "mode": {0: "auto", 1: "strict", 2: "fast"}[args.mode], "sharding": sharding, "shard_mode": mode, "layout": hp["layout"], "build_s": round(hp["build_s"], 2)},
            "roofline": flat_roofline(prof, hp["rows_local"], args.dim, B),
            "kernels_ms_per_step": rec["kernels_ms_per_step"],
            "fast_path": hp["stats"],
            "recall_at_10": {"flat": 1.0},
            "scaling_note": None if world == 1 else
                            ("a step = every rank's batch of 256 queries; `value` = all ranks' queries / the slowest rank's time (barrier + max over ranks). "
                             + ("Every GPU holds the whole 1M x 768 index (int8 shadow 0.77 GB + f32 rows 3.1 GB of 288 GB) and searches its own query stream: no data-path "
                                "collective (throughput mode; DESIGN.md 3.9). The row-sharded form of the same config with the RCCL all-gather is in `sharded_rows`."
                                if R_head == 1 else
                                "The 1M-row corpus is split over the ranks of a group and every rank searches its shard for the group's 256 queries; per batch a rank keeps "
                                "a fixed cost that does not shrink with its shard (post stage per shard, query preparation, exchange + merge, launch gaps), DESIGN.md 3.9.")),
        }
        if hp["host_buffers"]:
            line["host_buffers"] = hp["host_buffers"]
        if not args.no_cpu_baseline:
            cb = cpu_baseline_flat(args, *g0)
            if world == 1:
                line["cpu_baseline"] = cb
            else:       # N > 1: the CPU leg is a parity check only (the contract times it at N = 1)
                line["cpu_baseline"] = None
                line["parity"] = {"parity_checked_queries": cb["parity_checked_queries"], "parity_mismatches": cb["parity_mismatches"], "what": "rank 0's results for query batch 0 against the CPU oracle, bit for bit"}
        else:
            line["cpu_baseline"] = None
    if also_rows:
        # the partitioned form north_star names: ONE index split over all ranks by contiguous row blocks, every rank searches its shard for the same
        # 256 queries, per-shard top-K all-gathered inside the library and merged — strong scaling of the 1M config
        sp = flat_pass(world)
        if rank == 0:
            same = bool(np.array_equal(sp["g0"][2], g0[2]) and np.array_equal(sp["g0"][0], g0[0]) and np.array_equal(sp["g0"][1].view(np.uint32), g0[1].view(np.uint32)))
            r2 = sp["rec"]
            line["sharded_rows"] = {"qps": r2["qps"], "ms_per_step": r2["ms_per_step"], "scaling": "strong", "sharding": f"rows/{world}", "layout": sp["layout"],
                                    "sustained": r2["sustained"], "single_stream": r2.get("single_stream"), "region_ms": r2["region_ms"],
                                    "kernels_ms_per_step": r2["kernels_ms_per_step"], "roofline": flat_roofline(sp["prof"], sp["rows_local"], args.dim, B),
                                    "identical_to_the_replica_results_for_batch_0": same}

    def guarded(name, fn):
        """a leg that fails leaves its error in the line instead of taking the headline with it"""
        try:
            return fn()
        except Exception as e:        # noqa: BLE001
            import traceback
            return {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc(limit=4)}

    if "c1" in legs and world == 1:
        line["c1"] = guarded("c1", lambda: leg_c1(ctx, ca, args))

    def mix_queries(g, rows, nsub=MIX_SUB):
        """fresh draws from the corpus's mixture (rows past the corpus); replica group g continues group 0's stream"""
        return query_batches(ctx, B, args.dim, lambda p, i: ctx.synth_mixture(p, MIX_SEED, MIX_CENTERS, MIX_SIGMA, nsub, MIX_NOISE, rows + 7 + (g * NQB + i) * B, B, args.dim))

    # ---------------------------------------------------------------- clustered 1M corpus: Flat L2^2 (N = 1), IVFPQ (any N), hybrid (N = 1)
    need_mix = legs & ({"flat_l2", "ivfpq", "ivfpq_uniform", "hybrid"} if world == 1 else {"ivfpq"})
    if need_mix:
        flat2 = None
        if rank == 0 or world == 1:
            flat2 = ca.FlatIndex(ctx, args.dim, ca.L2_SQUARED)
            add_rows(ctx, flat2, 0, args.rows, args.dim, mix_fill(ctx, args.dim))
        q2 = mix_queries(rank // R_head, args.rows)
        Q2 = ctx.download(q2[0], (B, args.dim), np.float32)
        if "flat_l2" in legs and world == 1:
            r = guarded("flat_l2", lambda: leg_flat_l2(ctx, ca, args, timer, flat2, q2))
            line["flat_l2"] = r
        if "ivfpq" in legs:
            iv = guarded("ivfpq", lambda: leg_ivfpq(ctx, ca, args, timer, flat2, q2, Q2, args.rows, args.nlist, "ivfpq", group_comm(R_head), rank % R_head, R_head,
                                                    replicas=world // R_head, solo=world == 1))
            g_rep = iv.pop("_g0", None)
            if rank == 0:
                line["ivfpq"] = iv
                if "recall_at_10_vs_exact_flat" in iv:
                    line["recall_at_10"]["ivfpq_vs_exact_flat"] = iv["recall_at_10_vs_exact_flat"]
                if "recall_at_10_vs_oracle_ivfpq" in iv:
                    line["recall_at_10"]["ivfpq_vs_oracle_ivfpq"] = iv["recall_at_10_vs_oracle_ivfpq"]
            if also_rows:
                # the same index with its inverted lists dealt over ALL ranks, every rank searching for the same queries (group 0's stream)
                q2s = q2 if rank // R_head == 0 else mix_queries(0, args.rows)
                Q2s = ctx.download(q2s[0], (B, args.dim), np.float32)
                iv = guarded("ivfpq_sharded", lambda: leg_ivfpq(ctx, ca, args, timer, flat2, q2s, Q2s, args.rows, args.nlist, "ivfpq_sharded", wcomm, rank, world, replicas=1, solo=False))
                if q2s is not q2:
                    ctx.free(q2s[0])
                g_sh = iv.pop("_g0", None)
                if rank == 0:
                    if g_rep is not None and g_sh is not None:
                        iv["identical_to_the_replica_results_for_batch_0"] = bool(np.array_equal(g_sh[2], g_rep[2]) and all(
                            np.array_equal(g_sh[0][b, :g_rep[2][b]], g_rep[0][b, :g_rep[2][b]]) and np.array_equal(g_sh[1][b, :g_rep[2][b]].view(np.uint32), g_rep[1][b, :g_rep[2][b]].view(np.uint32)) for b in range(B)))
                    line["ivfpq_sharded"] = iv
        if "ivfpq_uniform" in legs and world == 1:
            # SURVEY 8(d)'s primary input: i.i.d. SplitMix64 rows — no cluster structure, the two-stage lower bound has nothing to remove: the honest worst case
            def uni():
                useed, qseed = 0xC0FFEE + 3, 0xBEEF + 3
                ufill = lambda buf, lo, m: ctx.synth_fill(buf, useed, lo * args.dim, m * args.dim)
                fu = ca.FlatIndex(ctx, args.dim, ca.L2_SQUARED)
                add_rows(ctx, fu, 0, args.rows, args.dim, ufill)
                qu = query_batches(ctx, B, args.dim, lambda p, i: ctx.synth_fill(p, qseed, i * B * args.dim, B * args.dim))
                Qu = ctx.download(qu[0], (B, args.dim), np.float32)
                r = leg_ivfpq(ctx, ca, args, timer, fu, qu, Qu, args.rows, args.nlist, "ivfpq_uniform", None, 0, 1, fill=ufill, corpus="UNIFORM SplitMix64 rows, SURVEY 8d")
                r.pop("_g0", None)
                fu.close(); ctx.free(qu[0])
                return r
            line["ivfpq_uniform"] = guarded("ivfpq_uniform", uni)
            if "recall_at_10_vs_exact_flat" in line["ivfpq_uniform"]:
                line["recall_at_10"]["ivfpq_uniform_vs_exact_flat"] = line["ivfpq_uniform"]["recall_at_10_vs_exact_flat"]
        if "hybrid" in legs and world == 1:
            line["hybrid"] = guarded("hybrid", lambda: leg_hybrid(ctx, ca, args, timer, q2, Q2))
        if flat2 is not None:
            flat2.close()
        ctx.free(q2[0])
    # ---------------------------------------------------------------- configs[3]: IVFPQ 10M (single GPU: the whole index; N > 1: list shards over ALL ranks,
    # the sharded form the config names — whatever --shard-mode says; `auto` / `replica` / `grid` add the throughput layout beside it)
    if "ivfpq10m" in legs:
        nb = args.big_rows
        nsub = max(MIX_SUB, nb * MIX_SUB // N_ROWS)         # the same ~15 rows per sub-centre as the 1M corpus (with 65536 sub-centres a 10M corpus has 150 near-duplicates per query)

        def big(R, tag):
            qb = mix_queries(rank // R, nb, nsub)
            Qb = ctx.download(qb[0], (B, args.dim), np.float32)
            fx = None
            if rank == 0:         # exact ground truth at 10M rows: a Flat index of the same rows (30 GB + 15 GB of fp16 shadow: nothing on a 288 GB part)
                fx = ca.FlatIndex(ctx, args.dim, ca.L2_SQUARED)
                add_rows(ctx, fx, 0, nb, args.dim, mix_fill(ctx, args.dim, nsub))
            r = leg_ivfpq(ctx, ca, args, timer, fx, qb, Qb, nb, args.big_nlist, tag, group_comm(R), rank % R, R, every_candidate=True, nsub=nsub,
                          replicas=world // R, solo=world == 1)
            r.pop("_g0", None)
            r["corpus"] = f"{MIX_CENTERS} centres, {nsub} sub-centres at {MIX_SIGMA}, noise {MIX_NOISE}"
            if fx is not None:
                fx.close()
            ctx.free(qb[0])
            return r
        r = guarded("ivfpq10m", lambda: big(world, "ivfpq10m"))
        if rank == 0:
            line["ivfpq10m"] = r
            if "recall_at_10_vs_exact_flat" in r:
                line["recall_at_10"]["ivfpq10m_vs_exact_flat"] = r["recall_at_10_vs_exact_flat"]
        if world > 1 and R_head != world:
            r = guarded("ivfpq10m_throughput", lambda: big(R_head, "ivfpq10m_throughput"))
            if rank == 0:
                line["ivfpq10m_throughput"] = r
    if "hnsw" in legs and world == 1:
        line["hnsw"] = guarded("hnsw", lambda: leg_hnsw(ctx, ca, args, timer))
        if "recall_at_10_vs_exact_flat" in line["hnsw"]:
            nav = line["hnsw"].get("navigable") if isinstance(line["hnsw"].get("navigable"), dict) else {}
            if "recall_at_10_vs_exact_flat" in nav:              # configs[2] is timed (and its recall quoted) on the navigable 1M graph; the reference-construction graph is a parity line
                line["recall_at_10"]["hnsw_navigable_1m_vs_exact_flat"] = nav["recall_at_10_vs_exact_flat"]
            line["recall_at_10"]["hnsw_reference_graph_vs_exact_flat"] = line["hnsw"]["recall_at_10_vs_exact_flat"]

    if wcomm is not None:
        wcomm.barrier()
        for cm in comms.values():
            cm.close()
        wcomm.close()
    if rank == 0:
        line["wall_s"] = round(time.time() - t_start, 1)
        # the full record (every leg's regions, rooflines, notes: tens of KB) goes to a file; stdout's LAST line is a compact record the
        # driver can parse from the tail it keeps
        full_json = json.dumps(line)
        for dst in (ROOT / "bench_legs.json", ROOT / "gpurun_out" / "bench_legs.json"):
            try:
                if dst.parent.is_dir():
                    dst.write_text(full_json + "\n")
            except OSError:
                pass
        sys.stdout.flush()
        print(full_json if args.full_line else json.dumps(compact_line(line)), flush=True)


def _rnd(x, nd=4):
    if isinstance(x, float):
        if abs(x) >= 1000:
            return int(round(x))                 # rates and byte counts: whole numbers (the compact line is read from a bounded tail)
        return float(f"{x:.{nd + 2}g}") if abs(x) < 1 else round(x, nd)
    return x


def _roof(r):
    if not isinstance(r, dict):
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "launches", "algorithmic_bytes_per_launch", "mfma_tops_int8",
            "mfma_frac_of_5000", "mfma_tflops", "mfma_frac_of_2500", "lds_frac_of_measured_ceiling")
    return {k: _rnd(r[k]) for k in keep if k in r}


def _cpu(c):
    if not isinstance(c, dict):
        return None
    keep = ("value", "unit", "cores", "kind", "single_thread_qps", "single_thread_latency_s", "parity_checked_queries", "parity_mismatches")
    out = {k: _rnd(c[k]) for k in keep if k in c}
    if "sample" in c:
        out["sample"] = str(c["sample"])[:150]
    return out


def _leg(rec):
    """one small summary per leg: qps, ms_per_step, single-stream qps, roofline fraction + kernel, parity mismatches"""
    if not isinstance(rec, dict):
        return None
    if "error" in rec:
        return {"error": str(rec["error"])[:200]}
    out = {}
    for k in ("qps", "ms_per_step"):
        if k in rec:
            out[k] = _rnd(rec[k])
    if isinstance(rec.get("sustained"), dict):
        out["sustained_qps"] = _rnd(rec["sustained"].get("qps"))
    if isinstance(rec.get("single_stream"), dict):
        out["single_stream_qps"] = _rnd(rec["single_stream"].get("qps"))
    if isinstance(rec.get("host_buffers"), dict):
        out["host_buffers_qps"] = _rnd(rec["host_buffers"].get("qps"))
    r = rec.get("roofline")
    if isinstance(r, dict):
        out["roofline"] = {k: _rnd(r.get(k)) for k in ("kernel", "frac", "avg_kernel_ms", "traffic") if r.get(k) is not None}
        if "lds_frac_of_measured_ceiling" in r:
            out["roofline"]["lds_frac"] = _rnd(r["lds_frac_of_measured_ceiling"])
    c = rec.get("cpu_baseline")
    if isinstance(c, dict):
        out["cpu_qps"] = _rnd(c.get("value")); out["parity_mismatches"] = c.get("parity_mismatches")
        out["parity_checked"] = c.get("parity_checked_queries")
    for k in ("recall_at_10_vs_exact_flat", "recall_at_10_vs_oracle_ivfpq", "rows", "identical_to_exact_kernels", "expansions_per_query", "identical_to_the_replica_results_for_batch_0"):
        if k in rec:
            out[k] = _rnd(rec[k])
    return out


def compact_line(full):
    """the driver-facing record: the contract's keys + roofline + cpu_baseline of the headline + one small summary per leg (< 6 KB)"""
    out = {"metric": "queries/sec + recall@10, 1M x 768 Flat & IVFPQ (value = Flat leg, exact search; IVFPQ + other configs under `legs`)"}
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"):
        out[k] = full.get(k)
    out["dtype"] = "int8 MFMA screening (i32 accumulate) + exact f32 rescoring: returned scores are f32, bit-identical to the CPU path"
    out["data"] = full.get("data")
    cfg = full.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "rows", "dim", "batch", "k", "metric", "sharding", "shard_mode", "layout") if k in cfg}
    out["timing"] = {"regions": (full.get("timing") or {}).get("regions"), "reported": "median region", "region_ms": (full.get("timing") or {}).get("region_ms")}
    out["roofline"] = _roof(full.get("roofline"))
    if isinstance(out["roofline"], dict):
        # the other legs' dominant-kernel fractions ride inside `roofline` (the part of the line every consumer keeps): HBM fraction unless named otherwise
        lf = {}
        for b in ("batch1", "batch64", "batch256"):
            r_ = ((full.get("flat_l2") or {}).get(b) or {}).get("roofline") if isinstance(full.get("flat_l2"), dict) else None
            if isinstance(r_, dict):
                lf[f"flat_l2_{b}"] = _rnd(r_.get("frac"))
        for nm in ("ivfpq", "ivfpq_uniform", "ivfpq10m"):
            r_ = (full.get(nm) or {}).get("roofline") if isinstance(full.get(nm), dict) else None
            if isinstance(r_, dict):
                lf[f"{nm}_adc"] = _rnd(r_.get("frac")); lf[f"{nm}_adc_lds"] = _rnd(r_.get("lds_frac_of_measured_ceiling"))
        if lf:
            out["roofline"]["other_legs_frac"] = lf
    out["cpu_baseline"] = _cpu(full.get("cpu_baseline"))
    if isinstance(full.get("parity"), dict):
        out["parity"] = {k: full["parity"].get(k) for k in ("parity_checked_queries", "parity_mismatches")}
    if isinstance(full.get("sharded_rows"), dict):
        sr = full["sharded_rows"]
        out["sharded_rows"] = {"qps": _rnd(sr.get("qps")), "ms_per_step": _rnd(sr.get("ms_per_step")), "scaling": "strong", "sharding": sr.get("sharding"),
                               "sustained_qps": _rnd((sr.get("sustained") or {}).get("qps")), "single_stream_qps": _rnd((sr.get("single_stream") or {}).get("qps")),
                               "identical_to_the_replica_results_for_batch_0": sr.get("identical_to_the_replica_results_for_batch_0"),
                               "scan_frac": _rnd((sr.get("roofline") or {}).get("frac"))}
    out["lanes"] = full.get("execution_lanes"); out["batches_in_flight"] = full.get("batches_in_flight")
    out["note_overlap"] = "value: `batches_in_flight` batches on `lanes` streams; single_stream_qps: 1 lane, 1 batch in flight"
    if isinstance(full.get("single_stream"), dict):
        out["single_stream_qps"] = _rnd(full["single_stream"].get("qps"))
    if isinstance(full.get("sustained"), dict):
        out["sustained_qps"] = _rnd(full["sustained"].get("qps"))
    if isinstance(full.get("host_buffers"), dict):
        out["host_buffers_qps"] = _rnd(full["host_buffers"].get("qps"))
    out["recall_at_10"] = {k: _rnd(v) for k, v in (full.get("recall_at_10") or {}).items()}
    legs = {}
    full = dict(full)
    if isinstance(full.get("hnsw"), dict) and isinstance(full["hnsw"].get("navigable"), dict):
        full["hnsw_navigable"] = full["hnsw"]["navigable"]
    for name in ("c1", "flat_l2", "ivfpq", "ivfpq_sharded", "ivfpq_uniform", "ivfpq10m", "ivfpq10m_throughput", "hnsw", "hnsw_navigable", "hybrid"):
        rec = full.get(name)
        if not isinstance(rec, dict):
            continue
        if "error" in rec:
            legs[name] = {"error": str(rec["error"])[:200]}
            continue
        if name == "flat_l2":
            legs[name] = {b: _leg(rec[b]) for b in ("batch1", "batch64", "batch256") if b in rec}
        elif name == "hybrid":
            legs[name] = {b: {k: v for k, v in _leg(rec[b]).items() if k in ("qps", "single_stream_qps", "cpu_qps", "parity_mismatches", "parity_checked")}
                          for b in rec if b.startswith("ivf_nprobe") or b == "bm25"}
            for b in ("rrf", "end_to_end", "end_to_end_nprobe32"):
                if isinstance(rec.get(b), dict):
                    legs[name][b] = {k: _rnd(v) for k, v in rec[b].items() if isinstance(v, (int, float)) and k != "queries_compared"}
            if isinstance(rec.get("legs_added"), dict):
                legs[name]["legs_added_qps"] = _rnd(rec["legs_added"].get("qps"))
            if isinstance(rec.get("cpu_baseline"), dict):
                legs[name]["cpu_qps"] = _rnd(rec["cpu_baseline"].get("value")); legs[name]["parity_mismatches"] = rec["cpu_baseline"].get("parity_mismatches")
        elif name == "c1":
            legs[name] = {k: _rnd(v) for k, v in rec.items() if isinstance(v, (int, float, bool))}
        elif name == "hnsw":
            # the reference-construction graph is a PARITY line: insertNode never promotes the entry point and prunes before the new node is linked, a search on it
            # stops after a few dozen expansions (recall ~0) — its q/s says nothing about the kernel and is not quoted; configs[2] is timed on `hnsw_navigable`
            lg = _leg(rec)
            legs["hnsw_reference_graph_parity"] = {k: lg[k] for k in ("parity_mismatches", "parity_checked", "recall_at_10_vs_exact_flat", "expansions_per_query") if k in lg}
            legs["hnsw_reference_graph_parity"]["rows"] = rec.get("rows"); legs["hnsw_reference_graph_parity"]["gpu_build_s"] = rec.get("build_s")
        else:
            legs[name] = _leg(rec)
            ec = rec.get("every_candidate_search")
            if isinstance(ec, dict):
                legs[name]["every_candidate_qps"] = _rnd(ec.get("qps")); legs[name]["adc_scan_ms"] = _rnd(ec.get("adc_scan_ms"))
            ts = rec.get("two_stage")
            if isinstance(ts, dict):
                legs[name]["pairs_left_alive_fraction"] = _rnd(ts.get("pairs_left_alive_fraction"))
    out["legs"] = legs
    out["full_record"] = "bench_legs.json (every leg's regions, kernels, rooflines with their definitions, CPU baselines)"
    out["wall_s"] = full.get("wall_s")
    return out


if __name__ == "__main__":
    main()
