set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_adc2_gpu.py tests/test_quant_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r6_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t1.log
tail -15 gpurun_out/r6_t1.log
