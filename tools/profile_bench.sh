#!/bin/bash
# The profile set behind bench.py's numbers, run on the GPU box:
#   1. rocprofv3 --kernel-trace --stats over the bench command  -> <out>/kernel_stats.csv, <out>/bench.json
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (never combined with trace domains)
#      -> <out>/field_traffic.json (bytes per launch of the dominant kernel)
# usage: tools/profile_bench.sh <outdir-under-gpurun_out>
OUT=${1:-prof_bench}
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-reference-route --scenes 0 > $D/bench.json 2> $D/bench.err || echo "trace pass failed"
cp $(find $D/trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $D/pmc_$C -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-reference-route --scenes 0 > $D/pmc_$C.json 2> $D/pmc_$C.err || echo "pmc pass $C failed"
done
python3 $R/tools/traffic_json.py $D > $D/field_traffic.json
cat $D/field_traffic.json
rm -rf $D/trace/*/*kernel_trace.csv
# config 3 (dense shells, wide candidate lists): kernel stats only
mkdir -p $D/config3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/config3/trace -- python3 $R/tools/config3_bench.py --steps 10 --warmup 4 > $D/config3/bench.json 2> $D/config3/bench.err || echo "config3 pass failed"
cp $(find $D/config3/trace -name "*kernel_stats.csv" | head -1) $D/config3/kernel_stats.csv
rm -rf $D/config3/trace
