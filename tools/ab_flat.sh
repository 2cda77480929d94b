#!/bin/bash
# tools/ab_flat.sh libA.so libB.so ... : the Flat headline leg with each of several builds of the library, on ONE box (kernel times per step + q/s)
cp comet_amd/libcomet_hip.so /tmp/orig.so
for L in "$@"; do cp $L comet_amd/libcomet_hip.so; python bench.py --legs flat --no-cpu-baseline --full-line 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('$(basename $L)', round(d['value']), 'q/s', 'single', round(d['single_stream']['qps']), d['kernels_ms_per_step'])
"; done
cp /tmp/orig.so comet_amd/libcomet_hip.so
