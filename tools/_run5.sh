set -u
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_bench.sh r4prof > gpurun_out/r4prof.log 2>&1; tail -3 gpurun_out/r4prof.log
bash tools/config_traffic.sh r4traffic > gpurun_out/r4traffic.log 2>&1; tail -3 gpurun_out/r4traffic.log
bash tools/kstats.sh refroute2 python3 $R/tools/reference_route_profile.py --frames 16 --out gpurun_out/reference_route_profile.json
