#!/bin/bash
# tools/ab_adc.sh libA.so libB.so ... : same-box A/B of the IVFPQ legs (every-candidate scan + pruned search) for several builds of the library
cp comet_amd/libcomet_hip.so /tmp/orig.so
for L in "$@"; do
  cp $L comet_amd/libcomet_hip.so
  python bench.py --legs ${AB_LEGS:-ivfpq,ivfpq_uniform,ivfpq10m} --no-cpu-baseline --regions 3 --sustain-s 0.5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
for k, v in d['legs'].items():
    print('$(basename $L)', k, 'pruned q/s', round(v.get('qps', 0)), 'single', round(v.get('single_stream_qps', 0)), 'every-cand q/s', round(v.get('every_candidate_qps', 0)), 'adc_scan ms', v.get('adc_scan_ms'), 'lds', v['roofline'].get('lds_frac'))
"
done
cp /tmp/orig.so comet_amd/libcomet_hip.so
