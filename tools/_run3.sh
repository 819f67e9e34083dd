set -u
R=$GRAFT_REPO_ROOT
cd $R
( timeout -k 10 500 python tools/fuzz_raster.py --cases 1500 --oracle; timeout -k 10 300 python tools/fuzz_raster.py --cases 600 --dense --seed 3; timeout -k 10 300 python tools/fuzz_pack.py --cases 600 ) > gpurun_out/fuzz_r4.log 2>&1; tail -4 gpurun_out/fuzz_r4.log
for v in product trav_unordered; do
  if [ $v = product ]; then python tools/bvh_bench.py 2>/dev/null; else QF_HIP_LIBRARY=$R/tools/experiments/_build/libqf_$v.so QF_HIP_LIBRARY_EXPERIMENT=1 python tools/bvh_bench.py 2>/dev/null; fi | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', {k: round(v,4) for k,v in r.items() if k.endswith('_ms')})"
done > gpurun_out/trav_unordered.txt; cat gpurun_out/trav_unordered.txt
ext() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'frame_ms', round(r['ms_per_frame'],3), 'field_ms', round(r['stage_ms']['field'],4), 'stages', {k: round(v,3) for k,v in r['stage_ms'].items()})"; }
python tools/config3_bench.py --steps 6 --warmup 4 2>/dev/null | ext product > gpurun_out/bf16_variants.txt
for v in bf16_no_mlp bf16_no_gather bf16_xpair; do
  QF_HIP_LIBRARY=$R/tools/experiments/_build/libqf_$v.so QF_HIP_LIBRARY_EXPERIMENT=1 python tools/config3_bench.py --steps 6 --warmup 4 2>/dev/null | ext $v >> gpurun_out/bf16_variants.txt
done
cat gpurun_out/bf16_variants.txt
python bench.py --gpus 4 --backend gloo --single-device --steps 5 --warmup 2 --scenes 2 --sharded-frames 2 > gpurun_out/bench_4rank_gloo.json 2> gpurun_out/bench_4rank_gloo.err; tail -2 gpurun_out/bench_4rank_gloo.err
