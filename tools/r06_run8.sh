#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for p in "WaterBodiesLike BlockGroup" "LakesLike ParksLike" "WaterBodies BlockGroup"; do
  set -- $p
  python3 tools/pip_alone_probe.py --base $1 --query $2 | tee -a gpurun_out/r06i_noinfo.txt
  RAYJOIN_AMD_LIB=$R/rayjoin_amd/variants/librj_noinfo.so python3 tools/pip_alone_probe.py --base $1 --query $2 | tee -a gpurun_out/r06i_noinfo.txt
done
