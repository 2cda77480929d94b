set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_comm_gpu.py tests/test_comm_multirank_gpu.py tests/test_hybrid.py tests/test_nonfinite_gpu.py -m gpu -x -q > gpurun_out/r6_t1.log 2>&1 ) 2>&1 | tail -3; tail -30 gpurun_out/r6_t1.log
