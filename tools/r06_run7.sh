#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for pair in "WaterBodies BlockGroup" "LakesNA ParksNA"; do
set -- $pair
for cap in 24 32 48 64; do
  RJ_BENCH_DEBUG_OPTS=run_cap=$cap timeout -k 10 300 python3 bench.py --base $1 --query $2 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/r06h_$1_cap$cap.json 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ro=[d['roofline'],d['roofline_other']]
l=[r for r in ro if r['kernel'].startswith('k_lsi')][0]; p=[r for r in ro if r['kernel'].startswith('k_pip')][0]
print(json.dumps({'pair':'$1 x $2','run_cap':$cap,'ms_per_step':d['ms_per_step'],'pipelined':d.get('ms_per_step_pipelined'),'sched':d['config']['kernel_schedule'],'lsi':[l['kernel'],l['kernel_ms'],l.get('kernel_ms_alone')],'pip':[p['kernel'],p['kernel_ms'],p.get('kernel_ms_alone')],'slots':d['index_slots_per_segment'],'build':d['build_index_ms']}))" | tee -a gpurun_out/r06h_runcap.txt
done
done
