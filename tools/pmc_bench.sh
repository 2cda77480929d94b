#!/bin/bash
# usage (on the GPU box): tools/pmc_bench.sh <outname>          (PMC_LEGS=ivfpq_uniform PMC_ADC_LEG=ivfpq_uniform tools/pmc_bench.sh <outname>: a pass of its own for another
#                                                                 IVFPQ leg — a file's adc_scan figures are ONE leg's; PMC_ROWS: that leg's row count, 1000000)
# Separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, LDS conflict counters, TCC) over a short bench.py run; the per-kernel
# averages land in gpurun_out/<outname>.json in the format bench.py's roofline.traffic reads (profiles/r*_pmc*.json; the file carries the
# fingerprint of the kernel sources it was measured on). PMC passes carry no trace options besides the implicit kernel dispatch
# records (gpurun refuses --pmc with sys/hip traces); every pass runs under its own timeout.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# COMET_ADC_ONE_STAGE=1: every adc_scan launch of the pass is the every-candidate scan the bench line's IVFPQ roofline is measured on
# (the pruned search's launches read a few percent of that and would only dilute the per-launch average)
RUN="env COMET_ADC_ONE_STAGE=1 COMET_LANES=1 python $GRAFT_REPO_ROOT/bench.py --legs ${PMC_LEGS:-flat,flat_l2,ivfpq,hybrid} --docs 2000 --no-cpu-baseline --regions 1 --steps 3 --warmup 1 --sustain-s 0.05"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 420 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- $RUN > $OUT/p$i.log 2>&1
done
# effective shader clock per kernel: GRBM_GUI_ACTIVE with the dispatches' begin / end times (--kernel-trace is the one trace option gpurun takes next to --pmc)
timeout 420 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p9 -- $RUN > $OUT/p9.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_collect.py $OUT gpurun_out/$1.json ${PMC_ADC_LEG:-ivfpq} ${PMC_ROWS:-1000000}
# the raw counter CSVs are tens of MB per pass: gpurun copies back at most 64 MiB of gpurun_out/
mkdir -p /tmp/pmc_raw && mv $OUT /tmp/pmc_raw/ 2>/dev/null || rm -rf $OUT
