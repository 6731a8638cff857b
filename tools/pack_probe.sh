set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_strip.py tests/test_gpu_leaf_order.py tests/test_gpu_stitch.py -x -q 2>&1 | tail -4 || exit 1
O=gpurun_out/r04_d_leaf_order.txt; : > $O
for pair in "WaterBodies BlockGroup 1:0" "WaterBodiesLike BlockGroup 1:0" "Gaussian5M Gaussian1M 1:0" "USCounty BlockGroup 1:0,1:0:0:0:1"; do
  set -- $pair
  timeout -k 10 400 python tools/leaf_order_probe.py --base $1 --query $2 --reps 3 --variants $3 >> $O 2>> gpurun_out/r04_d_leaf_order.err
  echo "--- $1 $2 rc=$?" >> $O
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_d_leaf_order.txt"):
    if l.startswith("{"):
        d=json.loads(l)
        if "variant" in d: print(d["pair"], d["variant"], "slots", d["slots_per_segment"], "used", d["used"], "build", d["first_build_ms"], d["build_ms"], "lsi", d["lsi_ms"], "pip", d["pip_query_ms"], "walk", d["pip_walk_ms"], "single", d["pip_single_kernel_ms"], d["pip_per_group"]["leaf_blocks"], d["lsi_per_group"]["leaf_blocks"])
        else: print(d)
    else: print(l.strip())
PY
