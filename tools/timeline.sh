#!/bin/bash
# usage (on the GPU box): tools/timeline.sh [bench args]  — kernel timeline of the last dispatches of a short bench.py run with the idle gaps between them
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --regions 2 --steps 10 "$@" > /tmp/kt.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$F" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-${TL_N:-24}:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1000 if prev_end else 0
    print("%8.1f us gap  %8.1f us  %s" % (gap, (e - s) / 1000, r["Kernel_Name"][:80]))
    prev_end = e
PY
