"""Host C++ (query_exec) checks that need no GPU: CLI surface, error behaviour, and the CDB
loader / .bin cache being byte-compatible between the C++ and Python hosts."""
import os
import subprocess

import numpy as np
import pytest

from rayjoin_amd import maps, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rayjoin_amd", "query_exec")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "rayjoin_amd", "host")])
    return EXE


def test_usage_and_flag_errors(exe):
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr  # query.cc:13-16
    r = subprocess.run([exe, "-poly1", "x.cdb", "-query", "lsi", "-mode", "rt"], capture_output=True, text=True)
    assert r.returncode == 3 and "no" in r.stderr and "lbvh" in r.stderr
    # -mode=grid is a device mode too (rj_grid.hip); it gets as far as the missing file
    r = subprocess.run([exe, "-poly1", "/nonexistent.cdb", "-query", "lsi", "-mode", "grid", "-grid_size", "64"],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "Cannot open file" in r.stderr
    r = subprocess.run([exe, "-poly1", "x.cdb", "-query", "lsi", "-mode", "quadtree"], capture_output=True, text=True)
    assert r.returncode == 3 and "Invalid index type" in r.stderr
    r = subprocess.run([exe, "-poly1=x.cdb", "-query=lsi", "-mode=lbvh", "-bogus=1"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown command line flag" in r.stderr
    r = subprocess.run([exe, "-poly1", "x.cdb", "-query", "nn", "-mode", "lbvh"], capture_output=True, text=True)
    assert r.returncode == 2 and "Invalid query" in r.stderr
    # RT-only flags of the reference are accepted and ignored
    r = subprocess.run([exe, "-poly1", "/nonexistent.cdb", "-query", "lsi", "-mode", "lbvh", "-ag", "2", "-win", "8",
                        "-nocheck", "-fau"], capture_output=True, text=True)
    assert r.returncode == 3 and "Cannot open file" in r.stderr


def test_cdb_text_roundtrip_and_rules(tmp_path):
    g = synth.lattice_map(3, 4, 7)
    p = str(tmp_path / "m.cdb")
    maps.write_cdb(p, g, "%.17g")
    g2 = maps.read_cdb(p)
    assert np.array_equal(g2.points, g.points) and np.array_equal(g2.row_index, g.row_index)
    assert np.array_equal(g2.chains, g.chains) and g2.bb == g.bb
    assert g2.n_edges == g.n_points - g.n_chains
    # comments / blank lines skipped (planar_graph.h:58-60)
    txt = open(p).read()
    open(p, "w").write("# header\n\n% other\n" + txt)
    assert np.array_equal(maps.read_cdb(p).points, g.points)
    for bad in ("0 1 0 0 1 2\n1.0 1.0\n",                    # np < 2          (:71)
                "0 2 0 1 1 2\n1.0 1.0\n1.0 1.0\n",           # repeated point  (:85)
                "0 3 0 2 1 2\n1.0 1.0\n2.0 2.0\n",           # incomplete      (:105)
                "0 2 0 1 1 2\n1.0 abc\n2.0 2.0\n"):          # unparsable      (:100)
        open(p, "w").write(bad)
        with pytest.raises(maps.CDBFormatError):
            maps.read_cdb(p)


def test_bin_cache_cpp_python_compatible(exe, tmp_path):
    """query_exec parses + serialises the map before it needs the GPU (it then fails loudly on
    rj_create when there is none); the .bin it wrote must equal what the Python host writes."""
    g = synth.lattice_map(4, 5, 9)
    p = str(tmp_path / "base.cdb")
    maps.write_cdb(p, g, "%.9f")
    gtxt = maps.read_cdb(p)
    ser = str(tmp_path / "ser")
    r = subprocess.run([exe, "-poly1", p, "-poly2", p, "-query", "lsi", "-mode", "lbvh", "-serialize", ser,
                        "-warmup", "0", "-repeat", "1"], capture_output=True, text=True)
    binp = os.path.join(ser, p.replace("/", "-") + ".bin")
    assert os.path.exists(binp), r.stderr
    gb = maps.deserialize_bin(binp)
    assert np.array_equal(gb.points, gtxt.points) and np.array_equal(gb.row_index, gtxt.row_index)
    assert np.array_equal(gb.chains, gtxt.chains) and gb.bb == gtxt.bb
    py = str(tmp_path / "py.bin")
    maps.serialize_bin(gtxt, py)
    assert open(py, "rb").read() == open(binp, "rb").read()
    assert maps.load_from(p, ser).n_points == g.n_points  # picks up the cache
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 3 and "rj_create failed" in r.stderr  # loud, no fallback


def _check_map_sample(g, s, rate):
    assert np.array_equal(s.chains, g.chains)  # every chain survives (topology preserved)
    for ic in range(g.n_chains):
        b, e = int(g.row_index[ic]), int(g.row_index[ic + 1])
        sb, se = int(s.row_index[ic]), int(s.row_index[ic + 1])
        want = e - b if e - b <= 2 else max(2, int((e - b - 1) * np.float32(rate))) + 1
        assert se - sb == want
        assert np.array_equal(s.points[sb], g.points[b]) and np.array_equal(s.points[se - 1], g.points[e - 1])
        # an ordered subsequence of the original chain
        orig = {tuple(p): k for k, p in enumerate(g.points[b:e])}
        idx = [orig[tuple(p)] for p in s.points[sb:se]]
        assert idx == sorted(set(idx))


def _check_edge_sample(g, s, rate):
    assert s.n_chains <= g.n_chains and list(s.chains[:, 0]) == list(range(s.n_chains))
    n_picked = int(g.n_edges * np.float32(rate))
    assert n_picked / 2 <= s.n_edges <= 2 * n_picked  # re-packing bridges gaps inside a chain
    allp = {tuple(p) for p in g.points}
    assert all(tuple(p) in allp for p in s.points)
    by_lr = {(int(c[3]), int(c[4])) for c in g.chains}
    assert all((int(c[3]), int(c[4])) in by_lr for c in s.chains)


@pytest.mark.parametrize("kind", ["map", "edges"])
def test_samplers(exe, tmp_path, kind):
    """-sample map|edges (planar_graph.h:255-399): C++ host (std::mt19937, like the reference) and
    Python host obey the same rules; the C++ draw is reproducible for a fixed -seed."""
    g = maps.read_cdb(_write(tmp_path, synth.lattice_map(5, 9, 13)))
    check = _check_map_sample if kind == "map" else _check_edge_sample
    py = (maps.sample_map_from if kind == "map" else maps.sample_edges_from)(g, 0.4, seed=5)
    check(g, py, 0.4)
    outs = []
    for rep in range(2):
        out = str(tmp_path / ("s%d.bin" % rep))
        r = subprocess.run([exe, "-poly1", str(tmp_path / "m.cdb"), "-query", "pip", "-mode", "lbvh", "-sample", kind,
                            "-sample_map_id", "0", "-sample_rate", "0.4", "-seed", "5", "-sample_output", out,
                            "-v", "1", "-warmup", "0", "-repeat", "1"], capture_output=True, text=True)
        assert os.path.exists(out), r.stderr
        assert ("Map is sampled" if kind == "map" else "Edges are sampled") in r.stderr
        outs.append(open(out, "rb").read())
        check(g, maps.deserialize_bin(out), 0.4)
    assert outs[0] == outs[1]
    r = subprocess.run([exe, "-poly1", str(tmp_path / "m.cdb"), "-query", "pip", "-mode", "lbvh", "-sample", "bogus",
                        "-sample_map_id", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid sample option" in r.stderr


def _write(tmp_path, g):
    p = str(tmp_path / "m.cdb")
    maps.write_cdb(p, g, "%.9f")
    return p
