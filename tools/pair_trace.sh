#!/bin/bash
# kernel timeline of one pair's settled steps (run ON the GPU box): tools/pair_trace.sh <tag> <base> <query>
set -e -o pipefail
TAG=${1:?tag}; BASE=${2:?base}; QUERY=${3:?query}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt_${BASE}_${QUERY} -o g -- python3 $R/bench.py --base $BASE --query $QUERY --no-cpu-baseline --no-secondary --steps 10 --warmup 5 > $R/gpurun_out/${TAG}_kt_${BASE}_${QUERY}.log 2>&1
cd $R
python3 tools/trace_tables.py gpurun_out/${TAG}_kt_${BASE}_${QUERY}/g_kernel_trace.csv --steps 2 --skip 6 > gpurun_out/${TAG}_step_trace_${BASE}_${QUERY}.txt
python3 tools/trace_tables.py gpurun_out/${TAG}_kt_${BASE}_${QUERY}/g_kernel_trace.csv --regimes > gpurun_out/${TAG}_regimes_${BASE}_${QUERY}.csv
