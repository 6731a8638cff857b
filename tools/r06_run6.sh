#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for p in "USCounty BlockGroup" "WaterBodies BlockGroup" "LakesNA ParksNA" "WaterBodiesLike BlockGroup"; do
  set -- $p
  python3 tools/lsi_stats_probe.py --base $1 --query $2 2>/dev/null | tee -a gpurun_out/r06f_lsi_stats.txt
done
tools/ysort_ab.sh r06f "WaterBodiesLike BlockGroup" "USCounty BlockGroup" "LakesLike ParksLike"
