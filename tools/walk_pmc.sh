#!/bin/bash
# SQ instruction counters of k_pip and k_pip_walk alone on the chip (GPU box): tools/walk_pmc.sh <tag>
set -e -o pipefail
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -o g -- python3 $R/tools/walk_probe.py --reps 2 --no-stats > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python3 tools/pmc_quick.py gpurun_out/${TAG}_sq k_pip
