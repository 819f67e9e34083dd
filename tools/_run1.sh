set -u
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_render.py -x -q -k "config3_bf16_dense" > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log
B="python bench.py --no-configs --scenes 0 --no-cpu-baseline --no-reference-route --repeats 3"
ext() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_frame', round(r['ms_per_step'],4), 'field_ms', round(r['stage_ms']['field'],4), 'points', r['quadrature_points_per_frame'])"; }
$B 2>/dev/null | ext product > gpurun_out/field_floor.txt
for v in no_mlp no_gather; do
  QF_HIP_LIBRARY=$R/tools/experiments/_build/libqf_$v.so QF_HIP_LIBRARY_EXPERIMENT=1 $B 2>/dev/null | ext $v >> gpurun_out/field_floor.txt
done
cat gpurun_out/field_floor.txt
QF_HIP_LIBRARY=$R/tools/experiments/_build/libqf_trav_stats.so QF_HIP_LIBRARY_EXPERIMENT=1 python tools/trav_stats.py > gpurun_out/trav_stats.json 2> gpurun_out/trav_stats.err; tail -5 gpurun_out/trav_stats.err
bash tools/kstats.sh refroute python3 tools/reference_route_profile.py --frames 16
python tools/bvh_bench.py > gpurun_out/bvh_bench.json 2>/dev/null; cat gpurun_out/bvh_bench.json
