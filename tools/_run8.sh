set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_final.log 2>&1; tail -3 gpurun_out/t_final.log
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.err
timeout -k 10 300 python tools/finetune_step_bench.py > gpurun_out/finetune_step.json 2>/dev/null; tail -c 600 gpurun_out/finetune_step.json; echo
timeout -k 10 300 python tools/band_bench.py > gpurun_out/band_bench.json 2>/dev/null; tail -c 700 gpurun_out/band_bench.json; echo
timeout -k 10 300 python examples/finetune_synthetic.py > gpurun_out/example.log 2>&1; tail -3 gpurun_out/example.log
timeout -k 10 300 python tools/bvh_bench.py > gpurun_out/bvh_bench_final.json 2>/dev/null; cat gpurun_out/bvh_bench_final.json
