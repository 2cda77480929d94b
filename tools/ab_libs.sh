#!/bin/bash
# tools/ab_libs.sh libA.so libB.so ... : the IVFPQ leg with each of several builds of the library, on ONE box
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); iv=d['ivfpq']; print('$1', round(iv['qps']), 'q/s', round(iv['ms_per_step'], 4), 'ms', {k: round(v, 4) for k, v in iv.get('kernels_ms_per_step', {}).items()})"; }
cp comet_amd/libcomet_hip.so /tmp/orig.so
for i in 1 2; do for L in "$@"; do cp $L comet_amd/libcomet_hip.so; python bench.py --legs ivfpq --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | show $(basename $L); done; done
cp /tmp/orig.so comet_amd/libcomet_hip.so
