set -u
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_mesh.py tests/test_gpu_refroute.py -x -q > gpurun_out/t4.log 2>&1; tail -3 gpurun_out/t4.log
python tools/bvh_bench.py > gpurun_out/bvh_bench2.json 2>/dev/null; cat gpurun_out/bvh_bench2.json
B="python bench.py --scenes 0 --no-cpu-baseline --no-reference-route --repeats 0 --steps 10"
ext() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['configs'][1]; print('$1', 'frame_ms', round(c['ms_per_frame'],4), 'shade_ms', round(c['roofline']['avg_launch_ms'],4), 'frac', round(c['roofline']['frac'],4))"; }
$B 2>/dev/null | ext product > gpurun_out/tex_variants.txt
for v in tex_stride48 tex_tiled64 tex_tiled48 tex_morton48 tex_ldspair tex_ldspair_tiled48; do
  QF_HIP_LIBRARY=$R/tools/experiments/_build/libqf_$v.so QF_HIP_LIBRARY_EXPERIMENT=1 $B 2>/dev/null | ext $v >> gpurun_out/tex_variants.txt
done
cat gpurun_out/tex_variants.txt
bash tools/kstats.sh refroute python3 $R/tools/reference_route_profile.py --frames 16
tail -12 gpurun_out/refroute.out
