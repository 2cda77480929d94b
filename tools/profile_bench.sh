#!/bin/bash
# usage (on the GPU box): tools/profile_bench.sh <tag>
#   1. the driver's bench line (with the CPU baselines)            -> gpurun_out/<tag>_bench_line.json
#   2. rocprofv3 --kernel-trace --stats of a short run             -> gpurun_out/<tag>_bench_rocprofv3_kernel_stats.txt
#   3. separate rocprofv3 --pmc passes (tools/pmc_bench.sh)        -> gpurun_out/<tag>_bench_pmc.json
# Copy what should be judged into profiles/ afterwards. Every profiler run sits under its own timeout.
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && timeout 1500 python bench.py 2> $O/${T}_bench.err | tail -1 > $O/${T}_bench_line.json; cp -f $R/bench_legs.json $O/${T}_bench_legs.json 2>/dev/null
# one execution lane (COMET_LANES=1): a kernel's duration under the profiler is its own, as in the line's `roofline` (measured in a one-lane region)
CMD="env COMET_LANES=1 python $R/bench.py --legs flat,flat_l2,ivfpq,ivfpq_uniform,ivfpq10m,hybrid,hnsw --hnsw-rows 20000 --hnsw-nav-rows 200000 --docs 20000 --no-cpu-baseline --regions 2 --steps 10 --sustain-s 0.2"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -- $CMD > /tmp/rp.log 2>&1
# bench.py starts child processes (tools/lds_gather_probe), each with a result directory of its own: every *kernel_stats.csv is merged, and
# the summary FAILS unless the product kernels are in it
CSVS=$(find /tmp/rp -name "*kernel_stats.csv")
python $R/tools/rocprof_summary.py --require flat_scan,adc_scan,hnsw_search --out $O/${T}_bench_rocprofv3_kernel_stats.txt $CSVS || echo "WARNING: rocprof summary lacks the product kernels" | tee -a $O/${T}_bench.err >&2
sed -i "1s|.*|# rocprofv3 --kernel-trace --stats of \`$CMD\` (Flat cosine, Flat L2^2 B=1/64/256, IVFPQ 1M clustered / uniform and 10M x nlist 4096 incl. GPU train + add, IVF + BM25, HNSW 20k incl. GPU build + 200k navigable), MI355X, $T|" $O/${T}_bench_rocprofv3_kernel_stats.txt
cd $R && tools/pmc_bench.sh ${T}_bench_pmc
# the other IVFPQ legs quote PMC passes of their own (a counter value is ONE leg's)
cd $R && PMC_LEGS=ivfpq_uniform PMC_ADC_LEG=ivfpq_uniform tools/pmc_bench.sh ${T}_bench_pmc_uniform
cd $R && PMC_LEGS=ivfpq10m PMC_ADC_LEG=ivfpq10m PMC_ROWS=10000000 tools/pmc_bench.sh ${T}_bench_pmc_10m
