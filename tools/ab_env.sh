#!/bin/bash
# usage: ab_env.sh "ENV_A" "ENV_B" rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
 for cfg in "$1" "$2"; do
  env $cfg python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$cfg', 'step', d['ms_per_step'], 'pipelined', d.get('ms_per_step_pipelined'), 'walk', d['roofline']['kernel_ms'], 'lsi', d['roofline_other']['kernel_ms'], 'points', d['lsi_points_ms'])
"
 done
done
