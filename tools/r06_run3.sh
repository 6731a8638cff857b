#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python -m pytest tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r06c_tests.log 2>&1; tail -2 gpurun_out/r06c_tests.log
for p in "WaterBodiesLike BlockGroup" "LakesLike ParksLike" "WaterBodies BlockGroup" "Gaussian5M Gaussian1M"; do
  set -- $p
  python3 tools/pip_alone_probe.py --base $1 --query $2 | tee -a gpurun_out/r06c_strip_order.txt
  python3 tools/pip_alone_probe.py --base $1 --query $2 --query-order 2 | tee -a gpurun_out/r06c_strip_order.txt
  python3 tools/pip_alone_probe.py --base $1 --query $2 --query-order 2 --strip-major 1 | tee -a gpurun_out/r06c_strip_order.txt
done
