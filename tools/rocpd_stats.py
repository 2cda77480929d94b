#!/usr/bin/env python3
"""Kernel statistics from a rocprofv3 results database (rocprofv3 --kernel-trace ... writes <name>_results.db, SQLite): per kernel calls / total / average / percentiles,
the launch resources the runtime recorded (dynamic + static LDS, VGPRs, AGPRs, scratch, grid, workgroup) and the durations of the dispatches that ran ALONE on the GPU
(no other dispatch of the listed kernels overlaps them) — with batches in flight on several streams a kernel's begin-to-end time includes what it shares the GPU with.

    python tools/rocpd_stats.py gpurun_out/cprof/cb_results.db [--alone 'flat_scan_qr|fast_post|prep_queries']
"""
import argparse
import re
import sqlite3
import statistics
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--alone", default="flat_scan|fast_post|prep_queries|adc_scan|hnsw_search")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = list(db.execute("select name, start, end, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, grid_x, workgroup_x from kernels order by start"))
    by = {}
    for r in rows:
        by.setdefault(r[0], []).append(r)
    tot = sum(r[2] - r[1] for r in rows) or 1
    short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "").replace("comet::", "")[:56]
    print(f"# {len(rows)} dispatches, {tot / 1e6:.1f} ms of kernel time")
    print(f"# {'kernel':<56} {'calls':>6} {'total ms':>9} {'%':>6} {'avg us':>8} {'min':>8} {'p25':>8} {'median':>8} {'p75':>8} {'max':>8}   lds B  vgpr agpr scratch   grid    wg")
    for n, rs in sorted(by.items(), key=lambda t: -sum(r[2] - r[1] for r in t[1])):
        d = sorted((r[2] - r[1]) / 1e3 for r in rs)
        q = lambda p: d[int(p * (len(d) - 1))]
        r0 = rs[0]
        print(f"  {short(n):<56} {len(d):>6} {sum(d) / 1e3:>9.2f} {100 * sum(d) * 1e3 / tot:>6.2f} {sum(d) / len(d):>8.2f} {d[0]:>8.2f} {q(.25):>8.2f} {q(.5):>8.2f} {q(.75):>8.2f} {d[-1]:>8.2f} "
              f"{r0[3]:>7} {r0[4]:>5} {r0[5]:>4} {r0[7]:>7} {r0[8]:>6} {r0[9]:>5}")
    pat = re.compile(a.alone)
    sel = [r for r in rows if pat.search(r[0])]
    alone = {}
    for i, r in enumerate(sel):
        ov = (i > 0 and sel[i - 1][2] > r[1]) or (i + 1 < len(sel) and sel[i + 1][1] < r[2])
        if not ov:
            alone.setdefault(r[0], []).append((r[2] - r[1]) / 1e3)
    print("# dispatches that overlapped no other dispatch of /" + a.alone + "/ (the kernel alone on the GPU):")
    for n, v in sorted(alone.items(), key=lambda t: -sum(t[1])):
        print(f"  {short(n):<56} {len(v):>6} avg {sum(v) / len(v):>8.2f} us, median {statistics.median(v):>8.2f}, min {min(v):>8.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
