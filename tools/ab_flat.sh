#!/bin/bash
# tools/ab_flat.sh libA.so libB.so ... : the Flat legs of bench.py with each of several builds of the library, on ONE box
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['flat_l2']; print('$1', round(d['value']), 'q/s', round(d['ms_per_step'], 4), 'ms frac', round(d['roofline']['frac'], 3), '| L2 legs', {k: (round(v['ms_per_step'], 4), round(v['roofline']['frac'], 3)) for k, v in f.items() if isinstance(v, dict) and 'qps' in v})"; }
cp comet_amd/libcomet_hip.so /tmp/orig.so
for i in 1 2 3; do for L in "$@"; do cp $L comet_amd/libcomet_hip.so; python bench.py --legs flat,flat_l2 --no-cpu-baseline --steps 40 2>/dev/null | show $(basename $L); done; done
cp /tmp/orig.so comet_amd/libcomet_hip.so
