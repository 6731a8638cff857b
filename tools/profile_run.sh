#!/bin/bash
# Regenerate the judged artefacts under profiles/ for the current tree (run ON the GPU box):
#   tools/profile_run.sh r01_h
# writes gpurun_out/<tag>_* (raw) -- afterwards, back in the authoring container, run
#   python tools/pmc_summary.py <tag> gpurun_out/<tag>_fetch gpurun_out/<tag>_write
#   cp gpurun_out/<tag>_kt/*kernel_stats.csv profiles/<tag>_kernel_stats.csv
#   cp gpurun_out/<tag>_bench.json profiles/<tag>_bench.json
# Counter passes are separate runs with --kernel-trace only (never combined with other traces); rocprofv3
# serialises the kernels of a counter pass, so they run with --serial-kernels: the counters describe each
# kernel alone on its full grid, whatever schedule the timed steps of a normal run settle on.
set -e -o pipefail
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
# (the driver's flags; the four schedule trials are warm-up steps, every timed step runs the settled schedule;
#  tools/trace_tables.py splits the per-kernel averages by grid size = by regime)
if [[ "${PASSES:-headline}" == "headline" ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -o g -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -o g -- python3 $R/bench.py --no-cpu-baseline --no-secondary --serial-kernels > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -o g -- python3 $R/bench.py --no-cpu-baseline --no-secondary --serial-kernels > $R/gpurun_out/${TAG}_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq1 -o g -- python3 $R/bench.py --no-cpu-baseline --no-secondary --serial-kernels --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_sq1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq2 -o g -- python3 $R/bench.py --no-cpu-baseline --no-secondary --serial-kernels --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_sq2.log 2>&1
cd $R
python3 tools/trace_tables.py gpurun_out/${TAG}_kt/g_kernel_trace.csv --regimes > gpurun_out/${TAG}_kernel_regimes.csv
python3 tools/trace_tables.py gpurun_out/${TAG}_kt/g_kernel_trace.csv --steps 3 --skip 4 > gpurun_out/${TAG}_step_trace.txt
# (the one stdout line = what the driver records; the full record -- secondaries, plan, checks -- is the --detail file)
timeout -k 10 500 python3 bench.py --check --steps 20 --warmup 5 --detail gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_bench_stderr.txt | grep "^{" > gpurun_out/${TAG}_bench_line.json
wc -c gpurun_out/${TAG}_bench_line.json; tail -c 300 gpurun_out/${TAG}_bench_line.json || true
fi
# Round 5: sections of counter passes beside the headline's full grids (tools/regime_probe.py: the profiler serialises
# kernels, so a regime is profiled kernel by kernel on that regime's grids).  "shared" = the headline's kernels on the
# grids of the shared schedule (k_lsi2 on 512 blocks, k_pip_walk2 on 1 536); the other pairs of the bench line (ring-shaped
# and the default run's secondaries) on their full grids.
cd /tmp
section() {  # name, regime_probe arguments ...
  local NAME=$1; shift
  for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq1:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "sq2:SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    timeout -k 10 300 rocprofv3 --pmc ${P#*:} --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${NAME}_${P%%:*} -o g -- python3 $R/tools/regime_probe.py "$@" > $R/gpurun_out/${TAG}_${NAME}_${P%%:*}.log 2>&1
  done
  echo "section $NAME done"
}
# (a section named <Base>_<Query> is that pair on its full grids; "shared" as above)
for S in ${SECTIONS:-shared WaterBodiesLike_BlockGroup LakesLike_ParksLike USCounty_NestedBlockGroup WaterBodies_BlockGroup USCounty_CrossingZipcode}; do
  if [[ "$S" == "shared" ]]; then section shared --lsi-blocks 512 --pip-blocks 1536
  elif [[ "$S" == *_* ]]; then section $S --base ${S%%_*} --query ${S#*_}; fi
done
cd $R
