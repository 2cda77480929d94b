cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export COMET_GUARDS=1
for i in 1 2 3 4 5 6 7 8; do ( HSA_ENABLE_SDMA=0 timeout 300 python tools/soak.py 110 60016001 flat,ivf,pq,ivfpq 2>&1 | tail -1 ) > gpurun_out/r6_sdma_hunt_$i.log 2>&1; done
tail -n 1 gpurun_out/r6_sdma_hunt_*.log | cut -c1-700
