#!/bin/bash
# The cell-record fetch-pattern experiment (field_eval.hip, QF_REC_SIM): times bench.py's field kernel with the clean
# build and with libraries built with -DQF_REC_SIM=1|2 (tools/_libqf_sim{1,2}.so, built by
#   QF_EXTRA_HIPCC_FLAGS=-DQF_REC_SIM=1 python -m quadraturefields_amd.build --force && cp quadraturefields_amd/libqf_hip.so tools/_libqf_sim1.so)
# for record tables of 64 GB / 2 GB / 128 MB, then one rocprofv3 --pmc pass (L2 / fabric request counters) per variant.
# usage (GPU box): bash tools/cell_record_experiment.sh > gpurun_out/cell_record_experiment.log
R=$GRAFT_REPO_ROOT
run() { timeout -k 10 200 python $R/bench.py --no-configs --scenes 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1: ms_per_frame', round(r['ms_per_step'],4), 'field_ms', round(r['stage_ms']['field'],4), 'points', r['quadrature_points_per_frame'])"; }
pmc() { cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$1; timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_$1 -- python3 $R/tools/field_bench.py --stage field --iters 3 --physical > /tmp/pmc_$1.log 2>&1; python3 $R/tools/pmc_summary.py /tmp/pmc_$1 | grep -A5 field_kernel | sed "s/^/$1: /"; cd $R; }
cp $R/quadraturefields_amd/libqf_hip.so /tmp/orig.so
run "hashed gather (product build)"
pmc product
for v in 1 2; do
  cp $R/tools/_libqf_sim$v.so $R/quadraturefields_amd/libqf_hip.so
  for lg in 29 24 20; do QF_REC_SIM_LOG2=$lg run "record pattern, $v line(s), table 2^$lg lines"; done
  QF_REC_SIM_LOG2=29 pmc sim${v}_64GB
done
cp /tmp/orig.so $R/quadraturefields_amd/libqf_hip.so
