#!/bin/bash
# bench.py's step under two settings of ONE environment knob of the library, on one box:
#   tools/env_ab.sh <tag> <VAR> <a> <b> "Base Query" ...     -> gpurun_out/<tag>_<VAR>_ab.txt
TAG=${1:?tag}; VAR=${2:?variable}; A=${3:?value a}; B=${4:?value b}; shift 4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
OUT=gpurun_out/${TAG}_${VAR}_ab.txt
: > $OUT
for p in "$@"; do
  set -- $p
  for v in $A $B; do
    env $VAR=$v timeout -k 10 300 python3 bench.py --base $1 --query $2 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/${TAG}_${VAR}_$1_$2_$v.json 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ro=[d['roofline'],d['roofline_other']]
l=[r for r in ro if r['kernel'].startswith('k_lsi')][0]; p=[r for r in ro if r['kernel'].startswith('k_pip')][0]
print(json.dumps({'pair':'$1 x $2','$VAR':'$v','ms_per_step':d['ms_per_step'],'pipelined':d.get('ms_per_step_pipelined'),'schedule':d['config']['kernel_schedule'],
  'lsi':[l['kernel'],l['kernel_ms'],l.get('kernel_ms_alone')],'pip':[p['kernel'],p['kernel_ms'],p.get('kernel_ms_alone')],'query_ms':p.get('query_ms'),
  'build_ms':d['build_index_ms'],'digest':d['result_digest']['pip_eids']}))" >> $OUT
  done
done
cat $OUT
