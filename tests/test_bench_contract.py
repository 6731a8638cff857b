"""The bench line committed under profiles/ carries every field of the bench contract."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))[-1]


def _load(path=None):
    """the full record of the latest committed default run: since round 6 the file bench.py writes beside its one stdout line
    ({"headline": ..., "secondary": [...]}); before, the single line itself with the secondaries inside"""
    d = json.load(open(path or _latest_bench()))
    if "headline" in d:
        d = dict(d["headline"], secondary=d["secondary"])
    return d


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)   # (imports numpy only: torch and the library are imported inside main())
    return m


def test_the_stdout_line_is_the_headline_alone_and_small(capsys, tmp_path):
    """Round 5's single line had grown to 20.9 KB and the driver's record (the last 8 KB of stdout) lost the headline's
    ms_per_step, roofline and cpu_baseline.  The line bench.py builds from the committed numbers must stay under 6 000 bytes,
    be the LAST line of stdout -- the only one -- and carry the contract fields; the secondaries go to stderr before it."""
    b = _bench_module()
    d = _load()
    sec = d.pop("secondary")
    b.emit(d, sec, str(tmp_path / "detail.json"))
    cap = capsys.readouterr()
    lines = cap.out.strip().split("\n")
    assert len(lines) == 1 and len(lines[0].encode()) <= b.HEADLINE_MAX_BYTES <= 6000
    h = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "ms_per_step_pipelined", "ranks", "detail"):
        assert k in h, k
    for k in ("kernel", "achieved", "peak", "frac", "traffic", "algorithmic_bytes", "kernel_ms", "traffic_frac", "limiter", "pmc_source"):
        assert k in h["roofline"], k
    # nothing nested deeper than one level below the line (config, rooflines, cpu_baseline, ranks, the per-pair summary)
    assert not any(isinstance(v2, dict) for k, v in h.items() if isinstance(v, dict) and k != "secondary" for v2 in v.values())
    assert set(h["secondary"]) == {s["metric"].split(", ", 1)[1] for s in sec}
    err = [json.loads(l) for l in cap.err.strip().split("\n") if l.startswith("{")]
    assert [e["metric"] for e in err] == [s["metric"] for s in sec] and all(e["line"] == "secondary" and "roofline" in e and "cpu_baseline" in e for e in err)
    full = json.load(open(tmp_path / "detail.json"))
    assert full["headline"]["plan"] and len(full["secondary"]) == len(sec)
    # the committed copy of the printed line, when the round has one
    p = _latest_bench().replace("_bench.json", "_bench_line.json")
    if os.path.exists(p):
        raw = open(p).read().strip()
        assert "\n" not in raw and len(raw.encode()) <= 6000 and json.loads(raw)["roofline"]["frac"] > 0


def test_committed_bench_line_has_the_contract_fields():
    d = _load()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # round 3: the dominant kernel is the PIP walk; the line says what the number is (algorithmic bytes vs moved bytes)
    assert r["kernel"] in ("k_pip_walk", "k_pip_walk2", "k_pip_strip", "k_lsi", "k_lsi2", "k_lsix", "k_lsi2x") and "query_ms" in (r if r["kernel"].startswith("k_pip_") else d["roofline_other"])
    if r.get("traffic"):
        assert 0 < r["traffic_frac"] < 1 and r["limiter"] in ("valu-issue", "dependent-load latency", "hbm-traffic") and 0 < r["limiter_frac"] <= 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    assert all(d["checks"].values())
    # value is whole-step throughput of the query map's segments
    n_s = 28793160
    assert abs(d["value"] - n_s / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    # the harder pairs ride in the same line, each with its own roofline and CPU baseline
    assert [s["metric"].split(", ")[1] for s in d["secondary"]][:2] == ["USCounty |><| NestedBlockGroup", "WaterBodies |><| BlockGroup"]
    if "r04" in os.path.basename(_latest_bench()) or len(d["secondary"]) > 2:   # round 4: a lake-shaped base map rides along
        assert d["secondary"][2]["metric"].split(", ")[1] == "WaterBodiesLike |><| BlockGroup"
        assert d["build_index_ms"] < 10 and d["rebuild_index_ms"] < d["build_index_ms"]   # the FIRST build of the map, on the device
        assert d["pip_caller_array"]["equals_map_owned_results"] is True and d["pip_caller_array"]["vs_map_owned"] < 1.06
    for s in d["secondary"]:
        assert s["value"] > 0 and "roofline" in s and "cpu_baseline" in s and s["config"]["schedule_settled_before_timing"] is True
    if len(d["secondary"]) > 3:   # round 5: the pair with the published intersection density, the handle's plan, the ranks that ran
        z = d["secondary"][3]
        assert z["metric"].split(", ")[1] == "USCounty |><| CrossingZipcode"
        assert 0.030 < z["intersections_per_query_segment"] < 0.040   # County x Zipcode in the reference's log: 0.0351
        assert d["plan"]["schedule"]["settled"] and d["plan"]["lsi"]["kernel"].startswith("k_lsi") and d["plan"]["pip"]["passes"] in (1, 3)
        assert d["ranks"]["world_size"] == d["n_gpus"] == 1
        ring = d["secondary"][2]["roofline"]
        assert ring["kernel"] == "k_pip_strip" and ring["traffic"] and ring["traffic"] > ring["algorithmic_bytes"]   # its own section of traffic.json
        assert d["roofline"]["on_its_share_of_the_chip"]["valu_per_query"] > 0


def test_no_schedule_trials_inside_the_timed_region():
    """Under the driver's flags (--steps 20 --warmup 5) the kernel schedule must be settled before the first timed
    step: the committed line of exactly that command says so, and this fails if it ever does not."""
    d = _load()
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["schedule_settled_before_timing"] is True
    assert "undecided" not in d["config"]["kernel_schedule"]


def test_traffic_file_matches_the_kernels_bench_reports():
    """profiles/traffic.json carries the PMC evidence bench.py quotes AND the hash of the kernel
    sources it was measured on; bench.py ignores it when that hash is not the tree's."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    lsi_k, walk_k = ("k_lsi2" if "k_lsi2" in t["traffic"] else "k_lsi"), ("k_pip_walk2" if "k_pip_walk2" in t["traffic"] else "k_pip_walk")  # (the headline's tree has the second order)
    assert {lsi_k, walk_k, "k_pip_exact"} <= set(t["traffic"]) and all(v > 0 for v in t["traffic"].values())
    assert len(t["kernel_source_hash"]) == 16 and t["tag"].startswith("r")
    for k in (lsi_k, walk_k):
        assert t["sq"][k]["SQ_ACTIVE_INST_VALU"] > 0 and t["sq"][k]["GRBM_GUI_ACTIVE"] > 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "kernel_source_hash" in src and "stale" in src


def test_gpus_flag_and_world_size_must_agree():
    """A launcher that started WORLD_SIZE ranks under a command line that says --gpus N is refused before anything
    touches the GPU (runs here, without one): the line's n_gpus can only be what both say."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and not r.stdout.strip()
    src = open(os.path.join(ROOT, "bench.py")).read()
    # N > 1 without a launcher: the ranks are a CHILD process started before torch is imported (no exec from a GPU process)
    assert src.index("sys.exit(launch_ranks(args))") < src.index("import torch\n    import torch.distributed as dist")
