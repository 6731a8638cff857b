#!/bin/bash
# Leaf blocks taller than wide sorted by y ("leaf_ysort") on / off: bench.py's step on one box.   tools/ysort_ab.sh <tag> "Base Query" ...
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
: > gpurun_out/${TAG}_ysort_ab.txt
for p in "$@"; do
  set -- $p
  for c in 0 1; do
    RJ_LEAF_YSORT=$c timeout -k 10 300 python3 bench.py --base $1 --query $2 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/${TAG}_yab_$1_$2_$c.json 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ro=[d['roofline'],d['roofline_other']]
l=[r for r in ro if r['kernel'].startswith('k_lsi')][0]; p=[r for r in ro if r['kernel'].startswith('k_pip')][0]
print(json.dumps({'pair':'$1 x $2','ysort':$c,'ms_per_step':d['ms_per_step'],'pipelined':d.get('ms_per_step_pipelined'),'schedule':d['config']['kernel_schedule'],
  'lsi':[l['kernel'],l['kernel_ms'],l.get('kernel_ms_alone')],'pip':[p['kernel'],p['kernel_ms'],p.get('kernel_ms_alone')],
  'build_ms':d['build_index_ms'],'digest':d['result_digest']['pairs']}))" >> gpurun_out/${TAG}_ysort_ab.txt
  done
done
cat gpurun_out/${TAG}_ysort_ab.txt
