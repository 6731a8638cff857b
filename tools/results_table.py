#!/usr/bin/env python3
"""DESIGN.md section 6's table from profiles/<tag>_bench*.json (run after tools/round_artifacts.sh): prints the rows."""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = [("%s_bench.json", "USCounty ⋈ BlockGroup (headline)"), ("%s_bench_USCounty_Zipcode.json", "USCounty ⋈ Zipcode"),
        ("%s_bench_USCounty_NestedBlockGroup.json", "USCounty ⋈ NestedBlockGroup"), ("%s_bench_WaterBodies_BlockGroup.json", "WaterBodies ⋈ BlockGroup (lattices)"),
        ("%s_bench_LakesNA_ParksNA.json", "LakesNA ⋈ ParksNA (lattices)"), ("%s_bench_Gaussian5M_Gaussian1M.json", "Gaussian5M ⋈ Gaussian1M"),
        ("%s_bench_WaterBodiesLike_BlockGroup.json", "WaterBodiesLike ⋈ BlockGroup"), ("%s_bench_LakesLike_ParksLike.json", "LakesLike ⋈ ParksLike"),
        ("%s_bench_BlockGroup_WaterBodiesLike.json", "BlockGroup ⋈ WaterBodiesLike")]
for f, name in rows:
    d = json.load(open(os.path.join(R, f % tag)))
    d = d.get("headline", d)   # (round 6: the --detail file; before: the single line)
    sched = d["config"]["kernel_schedule"]
    sched = ("shared " + sched[sched.index("(") + 1:sched.index(" blocks")]) if "share" in sched else ("turns" if "then" in sched else "full grids")
    cb = d.get("cpu_baseline") or {}
    print("| %s | %s | **%.3f** | %.3f | %s | `%s` | %.1f / %.2f | %.3f | %.1f M/s | pip alone %.3f lsi alone %.3f |" % (
        name, format(d["intersections"], ",").replace(",", " "), d["ms_per_step"], d.get("ms_per_step_pipelined") or 0, sched,
        d["config"]["pip_passes"].split(" ")[0], d["build_index_ms"], d.get("rebuild_index_ms") or 0, d.get("index_slots_per_segment") or 0,
        cb.get("value") or 0, d.get("pip_ms") or 0, d.get("lsi_ms") or 0))
