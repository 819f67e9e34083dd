#!/bin/bash
# The cell-record fetch-pattern experiment of round 2 (profiles/r2/cell_record_experiment.md), kept OUT of the product
# kernel: tools/experiments/cell_record_sim.patch adds the -DQF_REC_SIM=1|2 switch to a COPY of csrc/field_eval.hip; the
# patched sources are built into gpurun_out/_recsim/libqf_sim{1,2}.so, which report qf_abi_version() + 1000, and are
# loaded through QF_HIP_LIBRARY + QF_HIP_LIBRARY_EXPERIMENT=1 -- quadraturefields_amd/libqf_hip.so is never touched, so
# an interrupted run cannot leave an experiment build behind as the product library.
# The field kernel of a sim build returns garbage VALUES; only its timing and counters mean anything.
# usage (GPU box): bash tools/cell_record_experiment.sh > gpurun_out/cell_record_experiment.log
set -u
R=$GRAFT_REPO_ROOT
W=$R/gpurun_out/_recsim
rm -rf $W && mkdir -p $W/src
trap 'rm -rf $W/src $W/obj*' EXIT
cp -r $R/quadraturefields_amd/csrc/*.hip $R/quadraturefields_amd/csrc/*.cpp $R/quadraturefields_amd/csrc/*.h $W/src/
(cd $W/src && patch -p3 < $R/tools/experiments/cell_record_sim.patch) || { echo "patch failed"; exit 1; }
build() {   # $1 = 1|2
  mkdir -p $W/obj$1
  for f in field_eval field_eval_bf16 grid_backward mlp_train scan composite optim; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$W/src -DQF_REC_SIM=$1 -c $W/src/$f.hip -o $W/obj$1/$f.o || return 1
  done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$W/src -ffp-contract=off -c $W/src/exact.hip -o $W/obj$1/exact.o || return 1
  for f in bvh_build misc frame; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$W/src -DQF_ABI_VERSION_OFFSET=1000 -x hip -c $W/src/$f.cpp -o $W/obj$1/$f.o || return 1
  done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $W/libqf_sim$1.so $W/obj$1/*.o
}
run() { timeout -k 10 200 python $R/bench.py --no-configs --scenes 0 --no-cpu-baseline --no-reference-route 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1: ms_per_frame', round(r['ms_per_step'],4), 'field_ms', round(r['stage_ms']['field'],4), 'points', r['quadrature_points_per_frame'])"; }
pmc() { cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$1; timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_$1 -- python3 $R/tools/field_bench.py --stage field --iters 3 --physical > /tmp/pmc_$1.log 2>&1; python3 $R/tools/pmc_summary.py /tmp/pmc_$1 | grep -A5 field_kernel | sed "s/^/$1: /"; cd $R; }
run "hashed gather (product build)"
pmc product
for v in 1 2; do
  build $v || { echo "sim build $v failed"; exit 1; }
  export QF_HIP_LIBRARY=$W/libqf_sim$v.so QF_HIP_LIBRARY_EXPERIMENT=1
  for lg in 29 24 20; do QF_REC_SIM_LOG2=$lg run "record pattern, $v line(s), table 2^$lg lines"; done
  QF_REC_SIM_LOG2=29 pmc sim${v}_64GB
  unset QF_HIP_LIBRARY QF_HIP_LIBRARY_EXPERIMENT
done
