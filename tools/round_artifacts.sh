#!/bin/bash
# Everything profiles/<tag>_* is made of, in one go on the GPU box: tools/round_artifacts.sh r03_b
# (then, here: tools/pmc_summary.py + copies, see tools/profile_run.sh)
set -o pipefail
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
# (parts: "profile" = tools/profile_run.sh, "bench" = the other pairs + shards, "probes" = leaf order, schedule, build, ring stats;
#  a gpurun call is limited to 20 minutes: PARTS="bench" tools/round_artifacts.sh r04, then PARTS="probes" ...)
PARTS=${PARTS:-"profile bench probes"}
if [[ " $PARTS " == *" profile "* ]]; then
tools/profile_run.sh $TAG > gpurun_out/${TAG}_run.log 2>&1
echo "profile_run done"
fi
if [[ " $PARTS " == *" bench "* ]]; then
for p in "USCounty Zipcode" "USCounty NestedBlockGroup" "WaterBodies BlockGroup" "LakesNA ParksNA" "Gaussian5M Gaussian1M" "WaterBodiesLike BlockGroup" "LakesLike ParksLike" "BlockGroup WaterBodiesLike"; do
  set -- $p
  timeout -k 10 300 python3 bench.py --base $1 --query $2 --check --steps 20 --warmup 5 --detail gpurun_out/${TAG}_bench_$1_$2.json 2>/dev/null | tail -c 400
  echo "bench $1 $2 done"
done
for n in 2 4 8; do
  timeout -k 10 200 python3 bench.py --emulate-shard $n --steps 30 --warmup 6 --no-cpu-baseline --detail gpurun_out/${TAG}_shard$n.json 2>/dev/null | tail -c 300
done
timeout -k 10 200 python3 bench.py --base WaterBodies --query BlockGroup --emulate-shard 8 --steps 30 --warmup 6 --no-cpu-baseline --detail gpurun_out/${TAG}_wb_shard8.json 2>/dev/null | tail -c 300
timeout -k 10 200 python3 bench.py --serial-kernels --no-secondary --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/${TAG}_bench_serial.json 2>/dev/null | tail -c 300
echo "shards done"
fi
if [[ " $PARTS " == *" probes "* ]]; then
: > gpurun_out/${TAG}_leaf_order.txt
for p in "USCounty BlockGroup" "USCounty Zipcode" "USCounty NestedBlockGroup" "WaterBodies BlockGroup" "LakesNA ParksNA" "Gaussian5M Gaussian1M" "WaterBodiesLike BlockGroup" "LakesLike ParksLike"; do
  set -- $p
  timeout -k 10 400 python3 tools/leaf_order_probe.py --base $1 --query $2 --reps 3 2>/dev/null | tail -3 >> gpurun_out/${TAG}_leaf_order.txt
done
echo "leaf order done"
timeout -k 10 200 python3 tools/schedule_probe.py --steps 8 2>/dev/null | grep "^{" > gpurun_out/${TAG}_schedule_probe.txt
timeout -k 10 400 python3 tools/build_probe.py --maps USCounty,BlockGroup,WaterBodies,LakesNA,WaterBodiesLike,LakesLike 2>/dev/null | tail -12 > gpurun_out/${TAG}_build_probe.txt
timeout -k 10 300 python3 tools/ring_stats_probe.py 2>/dev/null > gpurun_out/${TAG}_ring_stats.txt
timeout -k 10 500 python3 tools/stack_depth_probe.py 2>/dev/null > gpurun_out/${TAG}_stack_depth.txt
fi
echo "all done"
