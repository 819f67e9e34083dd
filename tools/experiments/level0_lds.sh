#!/bin/bash
# VERDICT r2 item 7, a bounded experiment kept OUT of the product kernel: level 0 of the hash table (16^3 rows, 32 KB)
# served from LDS inside field_kernel (tools/experiments/level0_lds.patch on csrc/field_eval.hip, -DQF_L0_LDS).
# Build (in the build container): apply the patch to a copy of field_eval.hip, compile with -DQF_L0_LDS, compile misc.cpp
# with -DQF_ABI_VERSION_OFFSET=1000, link with the other objects of csrc/_obj into tools/experiments/_build/libqf_l0lds.so.
# Run (GPU box): bash tools/experiments/level0_lds.sh > gpurun_out/level0_lds.log
# Times bench.py's field kernel with both libraries and collects the L2 / fabric / TA counters of both.
R=$GRAFT_REPO_ROOT
L=$R/tools/experiments/_build/libqf_l0lds.so
run() { timeout -k 10 200 python $R/bench.py --no-configs --scenes 0 --no-cpu-baseline --no-reference-route 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1: ms_per_frame', round(r['ms_per_step'],4), 'field_ms', round(r['stage_ms']['field'],4), 'points', r['quadrature_points_per_frame'])"; }
pmc() { cd /tmp && export TMPDIR=/tmp; for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do n=$(echo $P | cut -c1-6 | tr -d ' '); rm -rf /tmp/pmc_$1_$n; timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_$1_$n -- python3 $R/tools/field_bench.py --stage field --iters 3 --physical > /tmp/pmc_$1_$n.log 2>&1; python3 $R/tools/pmc_summary.py /tmp/pmc_$1_$n | grep -A4 "field_kernel" | sed "s/^/$1: /"; done; cd $R; }
for rep in 1 2; do run "product (rep $rep)"; done
pmc product
export QF_HIP_LIBRARY=$L QF_HIP_LIBRARY_EXPERIMENT=1
for rep in 1 2; do run "level 0 in LDS (rep $rep)"; done
pmc l0lds
