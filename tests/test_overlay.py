"""Overlay row (BASELINE.json configs[3]): polyover_exec on the HIP path against the
oracle pipeline (-mode=grid semantics) -- the pattern of the reference's own test/test_overlay.sh:
diff the output CDB with an answer file."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rayjoin_amd import _capi, maps, ops, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import overlay_ref  # noqa: E402

D = os.path.join(ROOT, "tests", "golden", "sample_pair")
EXE = os.path.join(ROOT, "rayjoin_amd", "polyover_exec")


def test_oracle_pipeline_reproduces_overlay_answer(oracle, tmp_path):
    ctx = maps.Context([maps.read_cdb(os.path.join(D, "map0.cdb")), maps.read_cdb(os.path.join(D, "map1.cdb"))]).load()
    out = str(tmp_path / "ovl.txt")
    (nch, nfc), xs, pip = overlay_ref.oracle_overlay(oracle, ctx, out)
    assert nch > 100 and nfc > 20
    assert open(out).read() == open(os.path.join(D, "overlay_answer.txt")).read()
    # the answer is a loadable CDB itself
    g = maps.read_cdb(out)
    assert g.n_chains == nch
    for im in range(2):  # ordering contract of the per-map records
        e = xs[im]["eid"][:, im].astype(np.int64)
        assert (np.diff(e) >= 0).all()
        last = np.r_[e[1:] != e[:-1], True]
        assert (xs[im]["mid_point_polygon_id"][last] == -1).all()
        assert (xs[im]["mid_point_polygon_id"][~last] >= 0).all()


def _writer_twin():
    """tests/hosttwin/output_chain_twin.cc: the product's output-map writer (host/output_chain.h) driven without a GPU"""
    src = os.path.join(ROOT, "tests", "hosttwin", "output_chain_twin.cc")
    hdr = os.path.join(ROOT, "rayjoin_amd", "host")
    out = os.path.join(ROOT, "tests", "hosttwin", "_build", "output_chain_twin")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(hdr, "output_chain.h")), os.path.getmtime(os.path.join(hdr, "planar_graph.h")))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-pthread", "-I", hdr, "-o", out, src,
                               "-L", os.path.join(ROOT, "rayjoin_amd"), "-lrayjoin_amd", "-Wl,-rpath," + os.path.join(ROOT, "rayjoin_amd"),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return out


@pytest.mark.parametrize("pair", ["sample", "lattice", "rings"])
def test_product_writer_reproduces_the_oracle_pipelines_file(oracle, tmp_path, pair):
    """The C++ writer that polyover_exec uses -- chains cut into pieces, pieces labelled, faces and points numbered --
    fed the oracle pipeline's records and vertex faces must write the oracle pipeline's file byte for byte (on the sample
    pair that is the committed overlay answer).  No GPU: this is the host pass alone."""
    if pair == "sample":
        p0, p1 = os.path.join(D, "map0.cdb"), os.path.join(D, "map1.cdb")
    else:
        g0, g1 = ((synth.lattice_map(5, 40, 61), synth.lattice_map(9, 22, 62)) if pair == "lattice"
                  else (synth.ring_map(60, 900, seed=63), synth.lattice_map(6, 30, 64)))
        p0, p1 = str(tmp_path / "a.cdb"), str(tmp_path / "b.cdb")
        maps.write_cdb(p0, g0, "%.9f")
        maps.write_cdb(p1, g1, "%.9f")
    ctx = maps.Context([maps.read_cdb(p0), maps.read_cdb(p1)]).load()
    want = str(tmp_path / "want.txt")
    (nch, nfc), xs, pip = overlay_ref.oracle_overlay(oracle, ctx, want, 256)
    files = []
    for im in range(2):
        fx, fp = str(tmp_path / ("xs%d.bin" % im)), str(tmp_path / ("pip%d.bin" % im))
        np.ascontiguousarray(xs[im]).tofile(fx)
        np.ascontiguousarray(pip[im], dtype=np.int32).tofile(fp)
        files += [fx, fp]
    got = str(tmp_path / "got.txt")
    r = subprocess.run([_writer_twin(), p0, p1, files[0], files[2], files[1], files[3], got], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == [str(nch), str(nfc)]
    assert open(got).read() == open(want).read()
    if pair == "sample":
        assert open(got).read() == open(os.path.join(D, "overlay_answer.txt")).read()


def test_polyover_cli_errors():
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    r = subprocess.run([EXE, "-poly1", "a", "-poly2", "b", "-mode", "rt"], capture_output=True, text=True)
    assert r.returncode == 3 and "lbvh" in r.stderr
    r = subprocess.run([EXE, "-poly1", "a", "-mode", "lbvh"], capture_output=True, text=True)
    assert r.returncode == 2


@pytest.mark.gpu
def test_overlay_edge_xsects_parity(oracle):
    """Edges with several intersections each: map 0 has long edges, map 1 a fine lattice."""
    ctx = maps.Context([synth.lattice_map(3, 90, 61), synth.lattice_map(400, 1, 62)]).load()
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    dctx.BuildIndex(1)
    h = dctx.handle
    m0 = oracle.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = oracle.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
    want_pairs = oracle.lsi_grid(m0, m1, 1024)["eid"]
    lsi = ops.LSILBVH(dctx)
    lsi.Init(4 * len(want_pairs))
    n = lsi.Query(0)  # the overlay's direction: map 0 queries the LBVH of map 1
    assert n == len(want_pairs) > 2000
    for im in range(2):
        want = oracle.overlay_edge_xsects(m0, m1, im, want_pairs, 1024)
        out = h.alloc(48 * n)
        h.overlay_edge_xsects(im, lsi.queue, n, out)
        got = out.to_host(_capi.XSECT_DTYPE, n)
        multi = (np.diff(want["eid"][:, im].astype(np.int64)) == 0).sum()
        assert multi > 200 or im == 1  # many edges of map 0 carry several intersections
        for f in ("x_num", "x_den", "y_num", "y_den", "eid", "mid_point_polygon_id"):
            assert np.array_equal(got[f], want[f]), (im, f)
    dctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pair", ["sample", "multi"])
def test_polyover_exec_matches_oracle_pipeline(oracle, tmp_path, pair):
    if pair == "sample":
        p0, p1 = os.path.join(D, "map0.cdb"), os.path.join(D, "map1.cdb")
    else:
        p0, p1 = str(tmp_path / "a.cdb"), str(tmp_path / "b.cdb")
        maps.write_cdb(p0, synth.lattice_map(3, 90, 61), "%.9f")
        maps.write_cdb(p1, synth.lattice_map(120, 2, 62), "%.9f")
    ctx = maps.Context([maps.read_cdb(p0), maps.read_cdb(p1)]).load()
    want_path = str(tmp_path / "want.txt")
    overlay_ref.oracle_overlay(oracle, ctx, want_path, 512)
    got_path = str(tmp_path / "got.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-mode", "lbvh", "-output", got_path, "-xsect_factor", "1.0",
                        "-check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for phase in ("Build Index", "Intersection edges", "Map 0: Locate vertices in other map",
                  "Map 1: Locate vertices in other map", "Computer output polygons", "Check result", "Write to file"):
        assert " - %s: " % phase in r.stderr, phase
    assert "LSI passed check" in r.stderr and "Map 1: PIP passed check" in r.stderr  # vs the device-side grid
    assert open(got_path).read() == open(want_path).read()
    # the reference's own test (test/test_overlay.sh) diffs the output of two modes: same here
    grid_path = str(tmp_path / "got_grid.txt")
    r = subprocess.run([EXE, "-poly1", p0, "-poly2", p1, "-mode", "grid", "-grid_size", "512", "-output", grid_path,
                        "-xsect_factor", "1.0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(grid_path).read() == open(want_path).read()
    if pair == "sample":
        assert open(got_path).read() == open(os.path.join(D, "overlay_answer.txt")).read()
