#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box (box-to-box spread is 2-3 %, more than most kernel changes): bench.py's
headline step, alternately with each .so (RAYJOIN_AMD_LIB), `rounds` times; prints the medians side by side.
usage (on the GPU box): tools/ab_bench.py rayjoin_amd/variants/librj_base.so rayjoin_amd/librayjoin_amd.so [rounds] [extra bench flags ...]"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
keys = {"step": lambda d: d["ms_per_step"], "pipelined": lambda d: d.get("ms_per_step_pipelined"),
        "dom_in_step": lambda d: d["roofline"]["kernel_ms"], "dom_alone": lambda d: d["roofline"].get("kernel_ms_alone"),
        "other_in_step": lambda d: d["roofline_other"]["kernel_ms"], "lsi_wall": lambda d: d["lsi_ms"], "pip_wall": lambda d: d["pip_ms"]}
res = [{k: [] for k in keys}, {k: [] for k in keys}]
for r in range(rounds):
    for i, lib in enumerate(libs):
        env = dict(os.environ, RAYJOIN_AMD_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-secondary", "--no-cpu-baseline"] + extra,
                             env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print("run failed:", out.stderr[-2000:]); sys.exit(1)
        d = json.loads(line[0])
        for k, f in keys.items():
            v = f(d)
            if v is not None:
                res[i][k].append(v)
        print("round %d %s: step %.4f pipelined %s %s in step %.4f" % (r, "AB"[i], d["ms_per_step"], d.get("ms_per_step_pipelined"), d["roofline"]["kernel"], d["roofline"]["kernel_ms"]), flush=True)
print(json.dumps({"A": libs[0], "B": libs[1], "rounds": rounds,
                  "median": {k: [round(statistics.median(res[i][k]), 4) if res[i][k] else None for i in (0, 1)] for k in keys}}))
