set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
COMET_ADC_KERNEL=1 timeout 900 python -m pytest tests/test_quant_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "pq or adc or config3" > gpurun_out/r6_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t1.log
tail -3 gpurun_out/r6_t1.log
AB_LEGS=ivfpq,ivfpq_uniform,ivfpq10m timeout 1500 bash tools/ab_adc.sh comet_amd/libcomet_hip.so > gpurun_out/r6_ab_auto.log 2>&1; cat gpurun_out/r6_ab_auto.log
COMET_ADC_KERNEL=1 AB_LEGS=ivfpq10m timeout 1500 bash tools/ab_adc.sh comet_amd/libcomet_hip.so > gpurun_out/r6_ab_v1.log 2>&1; cat gpurun_out/r6_ab_v1.log
