set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bm25_gpu.py tests/test_hybrid.py tests/test_paper_kats.py tests/test_configs_gpu.py -m gpu -x -q -k "bm25 or hybrid or paper or config4" > gpurun_out/r6_t1.log 2>&1; tail -4 gpurun_out/r6_t1.log
timeout 900 python bench.py --legs hybrid --no-cpu-baseline --regions 3 --sustain-s 0.5 > gpurun_out/r6_hy.log 2>gpurun_out/r6_hy.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_hy.log').read().strip().splitlines()[-1])
print(json.dumps(d['legs']['hybrid']))
f=json.load(open('bench_legs.json')) if __import__('os').path.exists('bench_legs.json') else None
PY
python - <<'PY'
import json,glob
for f in glob.glob('bench_legs.json')+glob.glob('gpurun_out/bench_legs.json'):
    d=json.load(open(f)); h=d.get('hybrid',{}); b=h.get('bm25',{})
    print(f, {k:v for k,v in b.items() if 'kernel' in k or k in ('qps',)})
PY
