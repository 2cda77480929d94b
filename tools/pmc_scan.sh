#!/bin/bash
# usage: pmc_scan.sh <variant> <outdir> ; PMC passes over the Flat fast-path scan kernel (separate passes, kernel filter)
V=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" \
         "TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_ANY"; do
  i=$((i+1))
  COMET_SCAN_VARIANT=$V timeout 200 rocprofv3 --pmc $P --kernel-include-regex "flat_scan_f16" --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/tools/scan_microbench.py 1000000 256 3 2 > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
tot = {}
for f in sorted(glob.glob("gpurun_out/$2/p*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): tot[k] = sum(v) / len(v)
for k in sorted(tot): print("%-40s %.4g" % (k, tot[k]))
PY
