set -u
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/t_all2.log 2>&1; tail -4 gpurun_out/t_all2.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; tail -2 gpurun_out/bench_d.err
python tools/bvh_bench.py > gpurun_out/bvh_bench3.json 2>/dev/null; cat gpurun_out/bvh_bench3.json
QF_HIP_LIBRARY=$R/tools/experiments/_build/libqf_trav_stats.so QF_HIP_LIBRARY_EXPERIMENT=1 python tools/trav_stats.py > gpurun_out/trav_stats2.json 2> /dev/null
