#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python -m pytest tests/test_gpu_leaf_order.py tests/test_gpu_strip.py tests/test_gpu_realistic.py -x -q -m gpu > gpurun_out/r06j_tests.log 2>&1; tail -2 gpurun_out/r06j_tests.log
for p in "WaterBodiesLike BlockGroup" "LakesLike ParksLike" "Gaussian5M Gaussian1M"; do
  set -- $p
  timeout -k 10 300 python3 bench.py --base $1 --query $2 --check --steps 20 --warmup 5 --detail gpurun_out/r06_bench_$1_$2.json 2>/dev/null | tail -c 300
  echo
done
timeout -k 10 500 python3 bench.py --check --steps 20 --warmup 5 --detail gpurun_out/r06_bench.json 2>gpurun_out/r06_bench_stderr.txt | grep "^{" > gpurun_out/r06_bench_line.json
wc -c gpurun_out/r06_bench_line.json
