set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_refroute.py tests/test_gpu_composite.py tests/test_gpu_render.py tests/test_gpu_next.py tests/test_gpu_train.py -x -q > gpurun_out/t6.log 2>&1; tail -4 gpurun_out/t6.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --scenes 0 --no-configs > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; tail -2 gpurun_out/bench_e.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_e.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:round(v['ms_per_frame'],3) for k,v in d['reference_route'].items() if isinstance(v,dict)})"
bash tools/kstats.sh refroute3 python3 $R/tools/reference_route_profile.py --frames 16 --out gpurun_out/reference_route_profile.json
