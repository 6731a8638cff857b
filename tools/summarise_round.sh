#!/bin/bash
# After tools/profile_run.sh <tag> ran on the GPU box and gpurun_out/ came back: everything profiles/<tag>_* that is made HERE
# (counter summaries with their sections, copies of the trace tables), in one go.  usage: tools/summarise_round.sh r05
TAG=${1:?tag}; G=gpurun_out
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
SEC=""
for s in shared WaterBodiesLike_BlockGroup LakesLike_ParksLike USCounty_NestedBlockGroup WaterBodies_BlockGroup USCounty_CrossingZipcode; do
  [ -d $G/${TAG}_${s}_fetch ] && SEC="$SEC --section $s $G/${TAG}_${s}_fetch $G/${TAG}_${s}_write $G/${TAG}_${s}_sq1 $G/${TAG}_${s}_sq2"
done
python3 tools/pmc_summary.py $TAG $G/${TAG}_fetch $G/${TAG}_write $G/${TAG}_sq1 $G/${TAG}_sq2 $SEC > /dev/null
cp $G/${TAG}_kt/g_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $G/${TAG}_kernel_regimes.csv $G/${TAG}_step_trace.txt profiles/
python3 -c "
import json,sys
sys.path.insert(0,'$R')
from rayjoin_amd._capi import kernel_source_hash
t=json.load(open('profiles/traffic.json')); print('traffic.json', t['kernel_source_hash'], 'tree', kernel_source_hash(), t['traffic'])"
