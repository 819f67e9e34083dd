#!/bin/bash
# Collects PMC counters for one stage in separate rocprofv3 passes (never combined with trace domains).
# usage: tools/pmc.sh <stage> <outdir-under-gpurun_out> ; prints one line per (kernel, counter) aggregated.
# Every pass stays inside one block's counter budget (round 1 asked for six TA/TD counters in one pass and rocprofv3
# aborted with "Request exceeds the capabilities of the hardware to collect": the TA and the TD counters now have
# a pass each).  A failed pass FAILS the script (exit 1) after the remaining passes and the summary have run.
STAGE=${1:-field}; OUT=${2:-pmc_$STAGE}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
FAILED=""
for PASS in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES" ; do
  i=$((i+1))
  if ! timeout -k 10 240 rocprofv3 --pmc $PASS --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/tools/field_bench.py --stage $STAGE --iters 3 --physical > $R/gpurun_out/$OUT.p$i.log 2>&1 ; then
    echo "pass $i FAILED ($PASS): see gpurun_out/$OUT.p$i.log"
    FAILED="$FAILED $i"
  fi
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/$OUT > $R/gpurun_out/$OUT.summary.txt 2>&1
cat $R/gpurun_out/$OUT.summary.txt
if [ -n "$FAILED" ]; then
  echo "FAILED passes:$FAILED" | tee -a $R/gpurun_out/$OUT.summary.txt
  exit 1
fi
