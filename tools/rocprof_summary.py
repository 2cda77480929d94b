#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db, or *_kernel_stats.csv) into a
small text summary suitable for committing under profiles/.  usage: rocprof_summary.py <db-or-csv> [out.txt]"""
import csv
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in rows]   # the rocpd views are already in microseconds


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    src = sys.argv[1]
    rows = from_db(src) if src.endswith(".db") else from_csv(src)
    rows.sort(key=lambda r: -r[2])
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {src}", f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel"]
    for n, c, t, a, p in rows:
        lines.append(f"{c:7d} {t:14.1f} {a:12.2f} {p:7.2f}  {n}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
