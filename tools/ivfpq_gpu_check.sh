mkdir -p gpurun_out/r1g
timeout 900 python -m pytest tests/test_quant_gpu.py -x -q 2>&1 | tail -3
timeout 600 python bench_ivfpq.py --rows 10000000 --nlist 4096 --nprobe 32 --centers 0 --steps 10 --cpu-queries 64 --no-recall > gpurun_out/r1g/ivfpq_10m_uniform.json 2> gpurun_out/r1g/err1
timeout 400 python bench_ivfpq.py --rows 1000000 --nlist 1024 --nprobe 32 --steps 10 --cpu-queries 64 --no-recall > gpurun_out/r1g/ivfpq_1m.json 2> gpurun_out/r1g/err2
python - <<PY
import json
for f in ["ivfpq_10m_uniform","ivfpq_1m"]:
    j=json.load(open("gpurun_out/r1g/%s.json"%f)); print(f, round(j["value"]), round(j["ms_per_step"],3), round(j["roofline"]["frac"],3), j["kernels_ms_per_step"], j.get("cpu_baseline",{}).get("parity_mismatches"))
PY
tail -n 3 gpurun_out/r1g/err1
