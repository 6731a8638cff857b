#!/bin/bash
# k_pip_strip's scan unrolled by U = 1 / 2 / 4 entries per trip (RJ_STRIP_UNROLL), on the two ring-shaped pairs: the first
# pass alone, the PIP query alone, the step.  Run ON the GPU box; writes gpurun_out/<tag>_strip_unroll.txt
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R && mkdir -p gpurun_out
for pair in "WaterBodiesLike BlockGroup" "LakesLike ParksLike"; do
  set -- $pair
  for U in 1 2 4; do
    RJ_STRIP_UNROLL=$U python3 bench.py --base $1 --query $2 --steps 10 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
r=d['roofline'] if d['roofline']['kernel'].startswith('k_pip') else d['roofline_other']
print('$1 x $2 U=$U', 'step', d['ms_per_step'], 'pipelined', d.get('ms_per_step_pipelined'), r['kernel'], 'in step', r['kernel_ms'], 'alone', r.get('kernel_ms_alone'), 'pip query alone', r.get('query_ms_alone'), 'build', d['build_index_ms'])
"
  done
done | tee gpurun_out/${TAG}_strip_unroll.txt
