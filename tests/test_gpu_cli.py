"""query_exec end to end on a GPU: the reference's phases, timing format and result counts."""
import os
import re
import subprocess

import numpy as np
import pytest

from rayjoin_amd import maps, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rayjoin_amd", "query_exec")


def test_query_exec_lsi_and_pip(oracle, tmp_path):
    g0, g1 = synth.lattice_map(7, 100, 51), synth.lattice_map(15, 44, 52)
    p0, p1 = str(tmp_path / "a.cdb"), str(tmp_path / "b.cdb")
    maps.write_cdb(p0, g0, "%.9f")
    maps.write_cdb(p1, g1, "%.9f")
    ctx = maps.Context([maps.read_cdb(p0), maps.read_cdb(p1)]).load()
    m0 = oracle.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = oracle.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
    want = oracle.lsi_grid(m0, m1, 512)
    out = str(tmp_path / "pairs.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "lsi", "-mode", "lbvh", "-xsect_factor", "0.5",
                        "-warmup", "1", "-repeat", "2", "-output", out, "-v", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert int(re.search(r"Intersections: (\d+)", r.stderr).group(1)) == len(want)
    assert "Timing results:" in r.stderr
    for phase in ("Read map 0", "Read map 1", "Load Data", "Build Index", "Warmup", "Query"):
        assert re.search(r" - %s: [0-9.e+-]+ ms" % phase, r.stderr), phase
    got = np.loadtxt(out, dtype=np.int64).reshape(-1, 4)
    assert np.array_equal(got[:, :2], want["eid"].astype(np.int64))
    assert np.array_equal(got[:, 2], want["x_num"]) and np.array_equal(got[:, 3], want["y_num"])
    # PIP: every vertex of map 1
    outp = str(tmp_path / "pip.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "pip", "-mode", "lbvh", "-warmup", "1",
                        "-repeat", "1", "-output", outp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    eids = oracle.pip_grid(m0, 0, ctx.maps[1].pts, 512)
    got = np.loadtxt(outp, dtype=np.int64).reshape(-1, 2)
    assert np.array_equal(got[:, 0].astype(np.uint32), eids)
    assert np.array_equal(got[:, 1].astype(np.int32), m0.face_ids(eids))
    # -mode=grid: the same answers from the device-side uniform grid (the three-way comparison)
    outg = str(tmp_path / "pairs_grid.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "lsi", "-mode", "grid", "-grid_size", "512",
                        "-xsect_factor", "0.5", "-warmup", "1", "-repeat", "1", "-output", outg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(outg).read() == open(out).read()
    outpg = str(tmp_path / "pip_grid.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "pip", "-mode", "grid", "-grid_size", "512",
                        "-warmup", "0", "-repeat", "1", "-output", outpg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(outpg).read() == open(outp).read()
    # -mode=amd: the adapter classes of INTEGRATION.md section 2 (host/lsi_amd.h: LSIAMD / PIPAMD, subclasses of the operator
    # interfaces written against the C ABI alone), compiled into the binary -- same files as -mode=lbvh, byte for byte
    outa = str(tmp_path / "pairs_amd.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "lsi", "-mode", "amd", "-xsect_factor", "0.5",
                        "-warmup", "1", "-repeat", "2", "-output", outa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert int(re.search(r"Intersections: (\d+)", r.stderr).group(1)) == len(want)
    assert open(outa).read() == open(out).read()
    outpa = str(tmp_path / "pip_amd.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "pip", "-mode", "amd", "-warmup", "1",
                        "-repeat", "1", "-output", outpa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(outpa).read() == open(outp).read()
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "lsi", "-mode", "amd", "-xsect_factor", "0.000001",
                        "-warmup", "0", "-repeat", "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "overflow" in r.stderr
    # generated workloads (no -poly2) run and overflow is a reported error, not UB
    r = subprocess.run([EXE, "-poly1", p0, "-query", "lsi", "-mode", "lbvh", "-gen_n", "2000", "-gen_t", "5",
                        "-seed", "3", "-warmup", "0", "-repeat", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "Generate Workloads" in r.stderr
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "lsi", "-mode", "lbvh", "-xsect_factor", "0.000001",
                        "-warmup", "0", "-repeat", "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "overflow" in r.stderr


def test_native_rccl_allgatherv_single_rank(oracle, tmp_path):
    """The C-ABI's RCCL path with a 1-rank communicator (all a 1-GPU box allows): counts
    all-gather + exact-slice exchange reduce to the identity; query_exec drives it end to end."""
    from rayjoin_amd import _capi, ops
    ctx = maps.Context([synth.lattice_map(7, 100, 51), synth.lattice_map(15, 44, 52)]).load()
    d = ops.DeviceContext(ctx).LoadToDevice()
    d.BuildIndex(0)
    h = d.handle
    h.comm_init(1, 0, _capi.Handle.comm_unique_id())
    lsi = ops.LSILBVH(d)
    lsi.Init(100000)
    n = lsi.Query(1)
    out = h.alloc(8 * 100000)
    total, counts = h.allgather_pairs(lsi.queue, n, out, 100000)
    assert total == n and counts == [n]
    assert np.array_equal(out.to_host(np.uint32, 2 * n), lsi.queue.to_host(np.uint32, 2 * n))
    ids = h.alloc(4 * 1000)
    src = h.alloc(4 * 1000).from_host(np.arange(1000, dtype=np.uint32))
    total, counts = h.allgather_u32(src, 1000, ids, 1000)
    assert total == 1000 and np.array_equal(ids.to_host(np.uint32), np.arange(1000, dtype=np.uint32))
    with pytest.raises(_capi.QueueOverflow):
        h.allgather_u32(src, 1000, ids, 10)
    h.comm_destroy()
    d.close()
    p0, p1 = str(tmp_path / "a.cdb"), str(tmp_path / "b.cdb")
    maps.write_cdb(p0, synth.lattice_map(7, 100, 51), "%.9f")
    maps.write_cdb(p1, synth.lattice_map(15, 44, 52), "%.9f")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "lsi", "-mode", "lbvh", "-xsect_factor", "0.5",
                        "-warmup", "0", "-repeat", "1", "-nranks", "1", "-rank", "0", "-comm_file", str(tmp_path / "id")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert re.search(r"all ranks (\d+)", r.stderr)


def test_query_exec_pip_check_and_profile(tmp_path):
    """-check (default on, like src/flags.cc:9): RunPIPQuery re-runs the points through the device grid
    and compares by edge ENDPOINTS where eids differ (run_query.cu:22-99) -- a map with every chain
    stored twice has equal-geometry edges with different eids, which must still pass.  -profile prints
    the index build stages (deps/lbvh/lbvh/bvh.cuh:464-474)."""
    g0, g1 = synth.lattice_map(7, 60, 61), synth.lattice_map(15, 30, 62)
    # duplicate every chain of the base map: same coordinates, different eids
    dup = maps.PlanarGraph(np.concatenate([g0.chains, g0.chains]),
                           np.concatenate([g0.row_index[:-1], g0.row_index + g0.n_points]).astype(np.uint32),
                           np.concatenate([g0.points, g0.points]))
    p0, p1 = str(tmp_path / "a.cdb"), str(tmp_path / "b.cdb")
    maps.write_cdb(p0, dup, "%.9f")
    maps.write_cdb(p1, g1, "%.9f")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "pip", "-mode", "lbvh", "-warmup", "0", "-repeat", "1",
                        "-grid_size", "256", "-profile"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Checking point in polygon" in r.stderr and "Map: 0 passed check" in r.stderr
    assert re.search(r" - Check: [0-9.e+-]+ ms", r.stderr)
    assert "LBVH Profiling result:" in r.stdout and re.search(r"Radix sort: [0-9.]+", r.stdout) and re.search(r"Total: [0-9.]+", r.stdout)
    # -nocheck skips the phase; -mode=grid never checks itself (run_query.cu:452)
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "pip", "-mode", "lbvh", "-warmup", "0", "-repeat", "1",
                        "-nocheck"], capture_output=True, text=True)
    assert r.returncode == 0 and "passed check" not in r.stderr and " - Check:" not in r.stderr
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-query", "pip", "-mode", "grid", "-grid_size", "256", "-warmup", "0",
                        "-repeat", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "passed check" not in r.stderr
    # generated points (no -poly2) are checked too
    r = subprocess.run([EXE, "-poly1", p0, "-query", "pip", "-mode", "lbvh", "-gen_n", "5000", "-seed", "4", "-warmup", "0",
                        "-repeat", "1", "-grid_size", "256"], capture_output=True, text=True)
    assert r.returncode == 0 and "Map: 0 passed check" in r.stderr
