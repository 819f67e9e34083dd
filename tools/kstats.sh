#!/bin/bash
# rocprofv3 --kernel-trace --stats over one command, on the GPU box; keeps only the per-kernel summary:
#   tools/kstats.sh <name> <program> [args...]      ->  gpurun_out/<name>_kernel_stats.csv (+ <name>.out / .err)
# The program itself goes after `--` (never env / bash -c: the profiler's preload has already initialised the GPU).
NAME=$1; shift
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/_kstats_$NAME
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- "$@" > $R/gpurun_out/$NAME.out 2> $R/gpurun_out/$NAME.err || echo "kstats: profiled command failed"
cp $(find $D -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${NAME}_kernel_stats.csv
rm -rf $D
cd $R
