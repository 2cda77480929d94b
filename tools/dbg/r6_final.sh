set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh r06 > gpurun_out/r06_profile.log 2>&1
tail -c 1500 gpurun_out/r06_bench_line.json
ls -la gpurun_out | grep r06
# the step trace of adc_scan2_kernel (a -DA2_TRACE build of the library beside the product one)
cp comet_amd/libcomet_hip.so /tmp/product.so; cp tools/dbg/libcomet_hip_trace.so comet_amd/libcomet_hip.so
( echo "# tools/a2_trace.py 10000000 4096 (IVFPQ 10M x 768 UNIFORM rows, nlist 4096, nprobe 32, M 96, B 256, every-candidate search; COMET_ADC_KERNEL=2)"; COMET_ADC_KERNEL=2 timeout 600 python tools/a2_trace.py 10000000 4096 2>&1 | grep -v amdgpu.ids
  echo; echo "# tools/a2_trace.py (IVFPQ 1M x 768 UNIFORM rows, nlist 1024; COMET_ADC_KERNEL=2)"; COMET_ADC_KERNEL=2 timeout 600 python tools/a2_trace.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_adc2_trace.txt
cp /tmp/product.so comet_amd/libcomet_hip.so
head -12 gpurun_out/r06_adc2_trace.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_full_gpu_tests.log 2>&1 ) 2>&1 | tail -3; tail -3 gpurun_out/r06_full_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
AB_LEGS=ivfpq10m timeout 900 bash tools/ab_adc.sh comet_amd/libcomet_hip.so 2>&1 | tail -1
