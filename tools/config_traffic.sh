#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, never combined with trace domains) for the dominant kernels
# of the two other bench lines: field_kernel_bf16 (configs[2], tools/config3_bench.py) and texture_shade_packed_kernel
# (configs[4], tools/config5_bench.py --frame-path).  Writes <out>/config3_traffic.json and <out>/config5_traffic.json.
# usage: tools/config_traffic.sh <outdir-under-gpurun_out>
OUT=${1:-cfg_traffic}
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D/c3_$C -- python3 $R/tools/config3_bench.py --steps 3 --warmup 4 > $D/c3_$C.json 2> $D/c3_$C.err || echo "config3 pass $C failed"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D/c5_$C -- python3 $R/tools/config5_bench.py --steps 3 --warmup 3 --frame-path > $D/c5_$C.json 2> $D/c5_$C.err || echo "config5 pass $C failed"
done
python3 $R/tools/config_traffic.py $D c3 field_kernel_bf16 points_per_frame > $D/config3_traffic.json
python3 $R/tools/config_traffic.py $D c5 texture_shade_packed_kernel points_per_frame > $D/config5_traffic.json
cat $D/config3_traffic.json $D/config5_traffic.json
rm -rf $D/c3_FETCH_SIZE $D/c3_WRITE_SIZE $D/c5_FETCH_SIZE $D/c5_WRITE_SIZE
