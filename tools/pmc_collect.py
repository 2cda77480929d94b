#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one directory per pass) into per-kernel, per-launch averages.
usage: pmc_collect.py <dir-with-p*/...csv> <out.json>
HBM read bytes use the gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) reports half the bytes of a wide
coalesced streaming read -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is uncalibrated (reported as is)."""
import collections
import csv
import glob
import json
import re
import sys

src, out = sys.argv[1], sys.argv[2]
adc_leg = sys.argv[3] if len(sys.argv) > 3 else "ivfpq"                  # which bench leg the pass's adc_scan launches belong to
rows_of_pass = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-dispatch values
for f in sorted(glob.glob(f"{src}/p*/**/*counter_collection.csv", recursive=True)):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r.get("Dispatch_Id") or r.get("Correlation_Id"), r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[key[0]] = r["Kernel_Name"]
    for (did, cname), v in per_dispatch.items():
        acc[names[did]][cname].append(v)


# effective shader clock per kernel: the pass that carries --kernel-trace next to GRBM_GUI_ACTIVE gives every dispatch's begin / end time; GRBM_GUI_ACTIVE is
# summed over the chip's 8 XCDs, so clock = (GRBM_GUI_ACTIVE / 8) / (end - begin)
clock = collections.defaultdict(list)
for d in sorted(glob.glob(f"{src}/p*")):
    ktr = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not ktr or not cc:
        continue
    dur = {}
    for f in ktr:
        for r in csv.DictReader(open(f)):
            k = r.get("Dispatch_Id") or r.get("Correlation_Id")
            dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    for f in cc:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            k = r.get("Dispatch_Id") or r.get("Correlation_Id")
            if k in dur and dur[k][0] > 0:
                clock[dur[k][1]].append((float(r["Counter_Value"]) / 8.0 / dur[k][0] * 1e3, dur[k][0]))     # MHz, ns


def short(name):
    m = re.search(r"comet::(\w+)", name) or re.search(r"_ZN5comet\d+(\w+?_kernel)", name)
    return m.group(1) if m else name.split("(")[0]


kern = {}
for name, ctrs in acc.items():
    if "comet" not in name:
        continue
    e = {"launches": max(len(v) for v in ctrs.values())}
    for cname, vals in ctrs.items():
        e[f"{cname}_per_launch"] = sum(vals) / len(vals)
    if "FETCH_SIZE" in ctrs:
        e["hbm_read_bytes_per_launch_corrected"] = 2.0 * e["FETCH_SIZE_per_launch"] * 1024.0
    if "WRITE_SIZE" in ctrs:
        e["hbm_write_bytes_per_launch_uncalibrated"] = e["WRITE_SIZE_per_launch"] * 1024.0
    if "SQ_LDS_BANK_CONFLICT" in ctrs and e.get("SQ_LDS_IDX_ACTIVE_per_launch"):
        e["lds_conflict_fraction"] = e["SQ_LDS_BANK_CONFLICT_per_launch"] / e["SQ_LDS_IDX_ACTIVE_per_launch"]
    if name in clock and len(clock[name]) >= 3:
        c = sorted(x[0] for x in clock[name][len(clock[name]) // 4:])        # (the first quarter: warm-up launches)
        e["effective_clock_mhz"] = {"median": round(c[len(c) // 2]), "p10": round(c[len(c) // 10]), "p90": round(c[(9 * len(c)) // 10]), "dispatches": len(c),
                                    "kernel_us_in_this_pass": round(sorted(x[1] for x in clock[name])[len(clock[name]) // 2] / 1e3, 1),
                                    "how": "GRBM_GUI_ACTIVE / 8 XCDs / (End - Start) per dispatch, rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE"}
    key = name if name not in kern else name + "#2"
    kern[key] = e
    e["short"] = short(name)
import hashlib
import pathlib
_h = hashlib.sha256()
for _f in sorted((pathlib.Path(__file__).resolve().parent.parent / "comet_amd" / "csrc").glob("*.h*")):
    _h.update(_f.name.encode()); _h.update(_f.read_bytes())
json.dump({"source_sha": _h.hexdigest()[:16],       # fingerprint of the kernel sources (bench.py quotes this file only while it matches)
           "what": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | LDS | TCC, one pass each) over `bench.py --no-cpu-baseline --regions 1 --steps 3 --warmup 1`",
           "correction": "gfx950: read bytes = 2 * FETCH_SIZE(KB) * 1024 (MI355X_MICROARCH.md §HBM); WRITE_SIZE uncalibrated",
           "rows": rows_of_pass, "adc_leg": adc_leg, "kernels": kern}, open(out, "w"), indent=1)
print(f"wrote {out}: {len(kern)} kernels")
