# usage: r6_soak.sh <seed>: three 900 s soaks of one seed on the round's code (guards on: they are what would notice a late write), all five kinds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export COMET_GUARDS=1
S=$1
( timeout 1000 python tools/soak.py 900 $S flat,ivf,pq,ivfpq,hnsw 2>&1 | tail -2 ) > gpurun_out/r6_soak_${S}_a.log 2>&1
( HSA_ENABLE_SDMA=0 timeout 1000 python tools/soak.py 900 $S flat,ivf,pq,ivfpq 2>&1 | tail -2 ) > gpurun_out/r6_soak_${S}_b.log 2>&1
( COMET_ADC_KERNEL=2 timeout 1000 python tools/soak.py 900 $S flat,ivf,pq,ivfpq 2>&1 | tail -2 ) > gpurun_out/r6_soak_${S}_c.log 2>&1
tail -n 3 gpurun_out/r6_soak_${S}_*.log
