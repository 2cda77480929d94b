set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_quant_gpu.py -m gpu -x -q -k "ivfpq or pq or adc" > gpurun_out/r6_t1.log 2>&1; tail -2 gpurun_out/r6_t1.log
timeout 600 python tools/a2_trace.py > gpurun_out/r6_a2_trace.log 2>&1; grep -A6 "adc_scan of\|wave 4" gpurun_out/r6_a2_trace.log
AB_LEGS=${AB_LEGS:-ivfpq,ivfpq_uniform} timeout 1500 bash tools/ab_adc.sh comet_amd/libcomet_hip.so > gpurun_out/r6_ab_v2.log 2>&1; cat gpurun_out/r6_ab_v2.log
