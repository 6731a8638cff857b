#!/bin/bash
# The column index forced on / off on pairs whose base map is a lattice (the topology rule builds it for maps of closed rings only):
# bench.py's step -- LSI query + PIP query, the handle's schedule settled -- with RJ_PIP_COLUMNS=0 and =1 on one box.
#   tools/columns_ab.sh <tag> "Base Query" ...      -> gpurun_out/<tag>_columns_ab.txt
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
: > gpurun_out/${TAG}_columns_ab.txt
for p in "$@"; do
  set -- $p
  for c in 0 1; do
    RJ_PIP_COLUMNS=$c timeout -k 10 300 python3 bench.py --base $1 --query $2 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/${TAG}_cab_$1_$2_$c.json 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'pair':'$1 x $2','columns':$c,'ms_per_step':d['ms_per_step'],'pipelined':d.get('ms_per_step_pipelined'),'schedule':d['config']['kernel_schedule'],
  'pip_kernel':d['roofline']['kernel'] if d['roofline']['kernel'].startswith('k_pip') else d['roofline_other']['kernel'],
  'pip_ms_in_step':(d['roofline'] if d['roofline']['kernel'].startswith('k_pip') else d['roofline_other'])['kernel_ms'],
  'pip_ms_alone':(d['roofline'] if d['roofline']['kernel'].startswith('k_pip') else d['roofline_other']).get('kernel_ms_alone'),
  'lsi_ms_in_step':(d['roofline_other'] if d['roofline']['kernel'].startswith('k_pip') else d['roofline'])['kernel_ms'],
  'build_ms':d['build_index_ms'],'rebuild_ms':d['rebuild_index_ms'],'digest':d['result_digest']['pip_eids']}))" >> gpurun_out/${TAG}_columns_ab.txt
  done
done
cat gpurun_out/${TAG}_columns_ab.txt
