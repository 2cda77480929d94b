#!/bin/bash
# tools/ab_libs_env.sh "ENV=1 ..." libA.so libB.so ... : like ab_libs.sh with extra environment for every run
E="$1"; shift
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); iv=d['ivfpq']; print('$1', round(iv['qps']), 'q/s', round(iv['ms_per_step'], 4), 'ms', {k: round(v, 4) for k, v in iv.get('kernels_ms_per_step', {}).items() if k in ('adc_scan', 'pq_lut')})"; }
cp comet_amd/libcomet_hip.so /tmp/orig.so
for i in 1 2; do for L in "$@"; do cp $L comet_amd/libcomet_hip.so; env $E python bench.py --legs ivfpq --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | show $(basename $L); done; done
cp /tmp/orig.so comet_amd/libcomet_hip.so
