set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_mesh.py -x -q > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
timeout -k 10 300 python tools/bvh_bench.py > gpurun_out/bvh_bench4.json 2>/dev/null; cat gpurun_out/bvh_bench4.json
