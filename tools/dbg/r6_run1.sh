set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_full_gpu_tests.log 2>&1 ) 2>&1 | tail -3; tail -5 gpurun_out/r6_full_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
