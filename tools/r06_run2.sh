#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python -m pytest tests/test_gpu_strip.py tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r06b_tests.log 2>&1; tail -3 gpurun_out/r06b_tests.log
tools/fetch_calibrate > gpurun_out/r06_fetchcal_plain.json 2> gpurun_out/r06_fetchcal_plain.err; cat gpurun_out/r06_fetchcal_plain.json
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r06_fetchcal -o g -- $R/tools/fetch_calibrate > $R/gpurun_out/r06_fetchcal_run.json 2> $R/gpurun_out/r06_fetchcal.err )
cp gpurun_out/r06_fetchcal_run.json gpurun_out/r06_fetchcal/run.json
python3 tools/fetch_calibrate.py gpurun_out/r06_fetchcal gpurun_out/r06_fetch_calibration.json | tee gpurun_out/r06_fetch_calibration.txt
for cap in 16 24 32 48; do
  RJ_BENCH_DEBUG_OPTS=run_cap=$cap timeout -k 10 300 python3 bench.py --base WaterBodies --query BlockGroup --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/r06b_wb_cap$cap.json 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'run_cap':$cap,'ms_per_step':d['ms_per_step'],'pipelined':d.get('ms_per_step_pipelined'),'sched':d['config']['kernel_schedule'],'roofline':[d['roofline']['kernel'],d['roofline']['kernel_ms'],d['roofline'].get('kernel_ms_alone')],'other':[d['roofline_other']['kernel'],d['roofline_other']['kernel_ms'],d['roofline_other'].get('kernel_ms_alone')],'slots':d['index_slots_per_segment'],'build':d['build_index_ms']}))" | tee -a gpurun_out/r06b_wb_runcap.txt
done
( time python3 bench.py --base WaterBodiesLike --query BlockGroup --steps 5 --warmup 5 --detail gpurun_out/r06b_wbl.json 2>/dev/null ) 2>&1 | tail -c 1500
