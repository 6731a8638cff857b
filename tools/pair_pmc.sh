#!/bin/bash
# SQ counters of every kernel of one pair's step, each alone on the chip (GPU box): tools/pair_pmc.sh <tag> <base> <query>
set -e -o pipefail
TAG=${1:?tag}; BASE=${2:?base}; QUERY=${3:?query}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq_${BASE}_${QUERY} -o g -- python3 $R/bench.py --base $BASE --query $QUERY --no-cpu-baseline --no-secondary --serial-kernels --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_sq_${BASE}_${QUERY}.log 2>&1
python3 $R/tools/pmc_quick.py $R/gpurun_out/${TAG}_sq_${BASE}_${QUERY} k_lsi k_pip
