#!/bin/bash
# SQ instruction counters of the records kernels (k_lsi_points, k_lsi_points_gcd), each alone on the chip (GPU box):
#   tools/points_pmc.sh <tag> [base query]
set -e -o pipefail
TAG=${1:?tag}; BASE=${2:-USCounty}; QUERY=${3:-BlockGroup}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for split in 0 1; do
  RJ_POINTS_SPLIT=$split timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pts$split -o g -- python3 $R/bench.py --base $BASE --query $QUERY --no-cpu-baseline --no-secondary --serial-kernels --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_pts$split.log 2>&1
  echo "lsi_points_split $split"
  python3 $R/tools/pmc_quick.py $R/gpurun_out/${TAG}_pts$split k_lsi_points
done
