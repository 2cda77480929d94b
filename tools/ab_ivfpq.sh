#!/bin/bash
# A/B of two builds of libcomet_hip.so on ONE box (box-to-box spread is larger than most kernel changes):
#   comet_amd/libcomet_hip.so (new) against comet_amd/libcomet_hip_old.so; prints the IVFPQ leg's queries/s and scan time.
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); iv=d['ivfpq']; print('$1', round(iv['qps']), 'q/s', iv['ms_per_step'], 'ms', {k: round(v, 4) for k, v in iv.get('kernels_ms_per_step', {}).items() if 'adc' in k or 'select' in k})"; }
cp comet_amd/libcomet_hip.so /tmp/new.so
for i in 1 2 3; do
  cp /tmp/new.so comet_amd/libcomet_hip.so
  python bench.py --legs ivfpq --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | show NEW
  cp comet_amd/libcomet_hip_old.so comet_amd/libcomet_hip.so
  python bench.py --legs ivfpq --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | show OLD
done
cp /tmp/new.so comet_amd/libcomet_hip.so
