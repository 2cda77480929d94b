#!/usr/bin/env python3
"""Turn rocprofv3 `--kernel-trace --stats` results (rocpd sqlite .db, or one or more *_kernel_stats.csv — a profiled command that starts
child processes leaves one per process; they are merged) into a small text summary suitable for committing under profiles/.
usage: rocprof_summary.py [--require name1,name2,...] [--out out.txt] <db-or-csv> [<csv> ...]
--require: exit 1 unless every listed substring names at least one kernel of the summary (a summary of the wrong process is not a summary)."""
import argparse
import csv
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [(n, int(c), float(t)) for n, c, t, _a, _p in rows]   # the rocpd views are already in microseconds


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--require", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("src", nargs="+")
    a = ap.parse_args()
    merged = {}
    for src in a.src:
        for n, c, t in (from_db(src) if src.endswith(".db") else from_csv(src)):
            c0, t0 = merged.get(n, (0, 0.0))
            merged[n] = (c0 + c, t0 + t)
    rows = sorted(((n, c, t) for n, (c, t) in merged.items()), key=lambda r: -r[2])
    total = sum(r[2] for r in rows) or 1.0
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {' '.join(a.src)}", f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel"]
    for n, c, t in rows:
        lines.append(f"{c:7d} {t:14.1f} {t / max(c, 1):12.2f} {100.0 * t / total:7.2f}  {n}")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    else:
        sys.stdout.write(text)
    missing = [r for r in a.require.split(",") if r and not any(r in n for n, _c, _t in rows)]
    if missing:
        print(f"rocprof_summary: no kernel named like {missing} in {a.src}", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
