#!/bin/bash
# A/B of an environment switch on ONE box: tools/ab_env.sh VAR   (IVFPQ leg of bench.py with and without VAR=1, three rounds)
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); iv=d['ivfpq']; print('$1', round(iv['qps']), 'q/s', round(iv['ms_per_step'], 4), 'ms recall', iv['recall_at_10_vs_exact_flat'], {k: round(v, 4) for k, v in iv.get('kernels_ms_per_step', {}).items()})"; }
for i in 1 2 3; do
  python bench.py --legs ivfpq --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | show "default "
  env $1=1 python bench.py --legs ivfpq --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | show "$1=1"
done
