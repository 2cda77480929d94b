#!/usr/bin/env python3
"""shard_probe.py — ONE-GPU MODEL of what a rank of an N-GPU job does per batch. No multi-GPU box was available to the
builder: this is a model, not a scaling measurement.

For N in 1, 2, 4, 8 the one GPU plays EVERY rank of the job, one after the other, through the library's real sharded search path
(comet_index_search_sharded_async / _wait, merge_topk_kernel) with tests/libseq_rccl.so standing in for RCCL (COMET_RCCL_LIB):
  pass 1 (record): every virtual rank runs the timing loop; its collective contributions (stage-1 bounds of the two-stage IVFPQ search, its
                   top-K block) are logged;
  pass 2 (replay): the ranks run the same loop again, each collective answered from ALL ranks' logs — so a rank prunes with the GLOBAL bound
                   and merges the real blocks — and this pass is timed. Reported per N: the slowest rank's ms per step (what the job's step costs,
                   apart from the collectives' own transfer time, which the model prices from the block size) and rank 0's kernel breakdown.
(The round-3 probe ran rank 0 alone on a world-1 communicator: a list shard then pruned with ITS OWN stage-1 bound only, which is +inf-loose for the
queries whose nearest list lives on another rank — the "negative scaling" of the IVFPQ list shards at 2 / 4 ranks was that artefact.)

usage: shard_probe.py [flat] [ivfpq1m] [ivfpq10m] [--batch B]   (default: all three at B = 256; --batch 1024 for the large batch)"""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SEQ = ROOT / "tests" / "libseq_rccl.so"
os.environ["COMET_RCCL_LIB"] = str(SEQ)
import bench  # noqa: E402  (generators / constants only)
import comet_amd as ca  # noqa: E402
from comet_amd._lib import check  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 256
what = set(args) or {"flat", "ivfpq1m", "ivfpq10m"}
ctx = ca.Context(0)
lib = ctx.lib
dim, steps, regions = 768, 20, 3
seq = None


class VComm:
    """a communicator of virtual rank `rank` (no rendezvous: every rank lives in this process and takes the same id)"""

    def __init__(self, ident, rank, world):
        self.h = C.c_void_p()
        check(lib.comet_comm_create(ctx.h, ident, rank, world, C.byref(self.h)))

    def search_async(self, index, q, B_, k, oi, os_, oc, **kw):
        from comet_amd._lib import SearchParams
        p = SearchParams(k=int(k), threshold=0.0, nprobes=int(kw.get("nprobes", 0)), ef_search=0, filter_ids=None, n_filter=0, mode=int(kw.get("mode", 0)))
        t = C.c_uint64()
        check(lib.comet_index_search_sharded_async(index.h, self.h, C.c_void_p(q), int(B_), C.byref(p), C.c_void_p(oi), C.c_void_p(os_), C.c_void_p(oc), int(k), C.byref(t)))
        return t.value

    def wait(self, index, t, block):
        check(lib.comet_index_search_sharded_wait(index.h, self.h, C.c_uint64(t), 1 if block else 0))

    def sync(self):
        check(lib.comet_comm_sync(self.h))

    def close(self):
        lib.comet_comm_destroy(self.h)


def loop(comm, idx, q_ptrs, K, timed, **kw):
    ptrs = [(ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4)) for _ in range(3)]
    it = [0]

    def run(n):
        prev = None
        for _ in range(n):
            i = it[0]; it[0] += 1
            t = comm.search_async(idx, q_ptrs[i % len(q_ptrs)], B, K, *ptrs[i % 3], **kw)
            if prev is not None:
                comm.wait(idx, prev, False)
            prev = t
        comm.wait(idx, prev, True)
    run(3); comm.sync()
    if timed:
        ctx.profile(True); ctx.profile_reset()
    times = []
    for _ in range(regions):
        comm.sync(); t0 = time.perf_counter(); run(steps); comm.sync(); times.append(time.perf_counter() - t0)
    prof = ctx.profile_dump() if timed else {}
    ctx.profile(False)
    for b in ptrs:
        for p in b:
            ctx.free(p)
    tot = steps * regions
    return {"ms_per_step": sorted(times)[len(times) // 2] / steps * 1e3, "kernels_ms_per_step": {k: round(v[0] / tot, 4) for k, v in sorted(prof.items())}}


def model(name, N, make_rank, q_ptrs, K, **kw):
    """build the N shards, record, replay (timed); returns the slowest rank's step and rank 0's breakdown"""
    global seq
    ident = (C.c_uint8 * 128)()
    check(lib.comet_comm_unique_id(ident))
    if seq is None:
        seq = C.CDLL(str(SEQ))            # the handle comm.hip's dlopen made: same library, same logs
    seq.seq_rccl_reset()
    ranks = [(make_rank(r, N), VComm(ident, r, N)) for r in range(N)]
    # priming pass: the first sharded search of an index on a communicator compares the list placement across the ranks (two host all-reduces, once) — it must
    # not be part of the recorded sequence, which the replay pass repeats WITHOUT it
    seq.seq_rccl_set_mode(0)
    for idx, cm in ranks:
        o = (ctx.alloc(B * K * 4), ctx.alloc(B * K * 4), ctx.alloc(B * 4))
        cm.wait(idx, cm.search_async(idx, q_ptrs[0], B, K, *o, **kw), True); cm.sync()
        for p_ in o:
            ctx.free(p_)
    seq.seq_rccl_reset()
    seq.seq_rccl_set_mode(0)
    for idx, cm in ranks:
        loop(cm, idx, q_ptrs, K, False, **kw)
    ctx.sync()
    seq.seq_rccl_set_mode(1)
    res = [loop(cm, idx, q_ptrs, K, True, **kw) for idx, cm in ranks]
    rows = [len(idx) for idx, _ in ranks]
    for idx, cm in ranks:
        cm.close(); idx.close()
    slow = max(res, key=lambda r: r["ms_per_step"])
    out = {"world": N, "batch": B, "slowest_rank_ms_per_step": slow["ms_per_step"], "fastest_rank_ms_per_step": min(r["ms_per_step"] for r in res),
           "rank0_kernels_ms_per_step": res[0]["kernels_ms_per_step"], "rows_per_rank": rows}
    print(name, json.dumps(out), flush=True)
    return out


results = {}
if "flat" in what:
    rows = 1_000_000
    q_ptrs = bench.query_batches(ctx, B, dim, lambda p, i: ctx.synth_fill(p, bench.QUERY_SEED, i * B * dim, B * dim))

    def flat_rank(r, N):
        idx = ca.FlatIndex(ctx, dim, ca.COSINE)
        lo, hi = rows * r // N, rows * (r + 1) // N
        bench.add_rows(ctx, idx, lo, hi, dim, lambda buf, r0, m: ctx.synth_fill(buf, bench.CORPUS_SEED, r0 * dim, m * dim))
        return idx
    results["flat"] = [model("flat", N, flat_rank, q_ptrs, 100) for N in (1, 2, 4, 8)]
    ctx.free(q_ptrs[0])

for tag, rows, nlist in (("ivfpq1m", 1_000_000, 1024), ("ivfpq10m", 10_000_000, 4096)):
    if tag not in what:
        continue
    nsub = max(bench.MIX_SUB, rows * bench.MIX_SUB // bench.N_ROWS)
    mix = lambda buf, r0, m, nsub=nsub: ctx.synth_mixture(buf, bench.MIX_SEED, bench.MIX_CENTERS, bench.MIX_SIGMA, nsub, bench.MIX_NOISE, r0, m, dim)
    q_ptrs = bench.query_batches(ctx, B, dim, lambda p, i, nsub=nsub, rows=rows: ctx.synth_mixture(p, bench.MIX_SEED, bench.MIX_CENTERS, bench.MIX_SIGMA, nsub, bench.MIX_NOISE, rows + 7 + i * B, B, dim))
    ntrain = nlist * 100
    tbuf = ctx.alloc(ntrain * dim * 4)
    mix(tbuf, 0, ntrain)

    def pq_rank(r, N, rows=rows, nlist=nlist, tbuf=tbuf, ntrain=ntrain, mix=mix):
        idx = ca.IVFPQIndex(ctx, dim, ca.L2_SQUARED, nlist, 96, 8)
        check(lib.comet_index_train_dev(idx.h, C.c_void_p(tbuf), ntrain))
        if N > 1:
            idx.set_shard(r, N)
        bench.add_rows(ctx, idx, 0, rows, dim, mix)
        return idx
    results[tag] = [model(tag, N, pq_rank, q_ptrs, 10, nprobes=32) for N in (1, 2, 4, 8)]
    ctx.free(tbuf); ctx.free(q_ptrs[0])
print(json.dumps(results))
# What a node of W GPUs delivers under the layouts bench.py --shard-mode offers (rows = R ranks share one index copy and exchange; replica = R 1; grid = R x W / R):
# W / R groups serve their own query streams, a group's step is the modelled step of an R-rank job. Throughput relative to one GPU = (W / R) * t(1) / t(R).
# (The exchange's own transfer time over xGMI is not in t(R): one all-gather of R x B x K x 8 bytes per batch — 0.2 MB per rank at B 256, K 100 — a few microseconds.)
print("\n# layouts of a W-GPU node from the per-rank steps above: queries/s relative to ONE GPU (slowest rank's step; B = %d per group)" % B)
for tag, res in results.items():
    t = {r["world"]: r["slowest_rank_ms_per_step"] for r in res}
    print("# %s: t(R) ms = %s" % (tag, {k: round(v, 4) for k, v in t.items()}))
    for W in (2, 4, 8):
        cells = []
        for R in (1, 2, 4, 8):
            if R <= W and R in t:
                cells.append("R=%d x %d copies: %.2fx (%.0f q/s)" % (R, W // R, (W / R) * t[1] / t[R], (W / R) * B / (t[R] * 1e-3)))
        print("#   W=%d  " % W + "   ".join(cells))
