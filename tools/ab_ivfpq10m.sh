#!/bin/bash
# tools/ab_ivfpq10m.sh libA.so libB.so ... : the IVFPQ 1M + 10M legs (pruned and every-candidate) with each of several builds of the library, on ONE box
show() { python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
for leg in ('ivfpq','ivfpq_uniform','ivfpq10m'):
    iv=d.get(leg) or {}
    if 'qps' not in iv: print('$1', leg, iv.get('error')); continue
    ec=iv.get('every_candidate_search',{})
    print('$1', leg, round(iv['qps']), 'q/s', 'single', round(iv.get('single_stream',{}).get('qps',0)), 'every-candidate', round(ec.get('qps',0)), 'adc_scan_ms', round(ec.get('adc_scan_ms',0),4), {k: round(v, 4) for k, v in iv.get('kernels_ms_per_step', {}).items()})
"; }
cp comet_amd/libcomet_hip.so /tmp/orig.so
for L in "$@"; do cp $L comet_amd/libcomet_hip.so; python bench.py --legs ivfpq,ivfpq_uniform,ivfpq10m --no-cpu-baseline --full-line 2>/dev/null | show $(basename $L); done
cp /tmp/orig.so comet_amd/libcomet_hip.so
