set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lanes_gpu.py tests/test_hnsw_gpu.py tests/test_concurrency_gpu.py tests/test_hybrid.py -m gpu -x -q > gpurun_out/r6_t1.log 2>&1; tail -3 gpurun_out/r6_t1.log
( time timeout 1200 python bench.py --legs hnsw --regions 3 --sustain-s 0.5 --no-cpu-baseline > gpurun_out/r6_hnsw.log 2> gpurun_out/r6_hnsw.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_hnsw.log').read().strip().splitlines()[-1])
print(json.dumps(d['legs'], indent=0))
PY
