"""the chunked parallel CDB text parser (rayjoin_amd/host/planar_graph.h,
read_pgraph_parallel) must be indistinguishable from the reference's serial loop
(src/map/planar_graph.h:42-126, restated as read_pgraph_serial) and from the Python loader: same
PlanarGraph bytes on well-formed files -- including the odd-but-legal spellings an istream accepts --
and the same error, naming the same line, on malformed ones."""
import os
import subprocess

import numpy as np
import pytest

from rayjoin_amd import maps, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "hosttwin", "_build", "cdb_tool")


@pytest.fixture(scope="module")
def tool():
    src = os.path.join(ROOT, "tools", "cdb_tool.cc")
    hdr = os.path.join(ROOT, "rayjoin_amd", "host", "planar_graph.h")
    os.makedirs(os.path.dirname(TOOL), exist_ok=True)
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.dirname(hdr), "-o", TOOL, src])
    return TOOL


def _parse(tool, path, threads, out=None):
    cmd = [tool, "parse", path, str(threads)] + ([out] if out else [])
    return subprocess.run(cmd, capture_output=True, text=True)


def test_parallel_equals_serial_equals_python(tool, tmp_path):
    g = synth.lattice_map(40, 37, 5)
    p = str(tmp_path / "m.cdb")
    maps.write_cdb(p, g, "%.9f")
    outs = {}
    for t in (1, 2, 3, 7, 16):
        o = str(tmp_path / ("o%d.bin" % t))
        r = _parse(tool, p, t, o)
        assert r.returncode == 0, r.stderr
        outs[t] = open(o, "rb").read()
    assert all(v == outs[1] for v in outs.values())
    py = str(tmp_path / "py.bin")
    maps.serialize_bin(maps.read_cdb(p), py)
    assert open(py, "rb").read() == outs[1]


ODD_BUT_LEGAL = """# a comment
% another

0 3 0 2 1 2
+1.5 2.5
 1e1\t-2.E0
.5 5. trailing tokens are ignored
1 2 7 8 0 0 extra
-0 7abc
3e0 4
"""


def test_odd_spellings_go_through_the_same_extraction(tool, tmp_path):
    p = str(tmp_path / "odd.cdb")
    open(p, "w").write(ODD_BUT_LEGAL)
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    r1, r4 = _parse(tool, p, 1, a), _parse(tool, p, 4, b)
    assert r1.returncode == 0 and r4.returncode == 0, (r1.stderr, r4.stderr)
    assert open(a, "rb").read() == open(b, "rb").read()
    g = maps.deserialize_bin(a)
    assert g.n_chains == 2 and g.n_points == 5
    assert g.points[0].tolist() == [1.5, 2.5] and g.points[1].tolist() == [10.0, -2.0] and g.points[2].tolist() == [0.5, 5.0]
    assert g.points[3].tolist() == [0.0, 7.0] and np.signbit(g.points[3][0])  # "-0", and "7abc" reads as 7
    assert g.points[4].tolist() == [3.0, 4.0]


BAD = {
    "bad point": "0 3 0 2 1 2\n1 1\nx y\n3 3\n",
    "comma": "0 2 0 1 1 2\n1,1\n2 2\n",
    "repeated point": "0 3 0 2 1 2\n1 1\n1 1\n3 3\n",
    "np < 2": "0 2 0 1 1 2\n1 1\n2 2\n1 1 0 0 1 2\n5 5\n",
    "short header": "0 2 0 1 1 2\n1 1\n2 2\n1 2 0\n5 5\n6 6\n",
    "incomplete": "0 2 0 1 1 2\n1 1\n2 2\n1 3 0 2 1 2\n5 5\n6 6\n",
    "huge np": "0 2 0 1 1 2\n1 1\n2 2\n1 999999999999 0 2 1 2\n5 5\n6 6\n2 2 0 1 0 0\n7 7\n8 8\n",
    "crlf": "0 2 0 1 1 2\r\n1 1\r\n\r\n2 2\r\n",
    "two bad lines": "0 3 0 2 1 2\n1 1\n2 2\n3 3\n1 3 0 2 1 2\n5 5\nbad\n6 6\n2 2 0 1 0 0\n7 7\nworse\n",
    "hex float": "0 2 0 1 1 2\n0x10 1\n2 2\n",  # an istream reads 0, then cannot read "x10"
    "overflowing number": "0 2 0 1 1 2\n1e999 1\n2 2\n",
    "int overflow in header": "99999999999999999999 2 0 1 1 2\n1 1\n2 2\n",
}


@pytest.mark.parametrize("name", sorted(BAD))
def test_malformed_files_fail_identically(tool, tmp_path, name):
    # bury the defect in enough well-formed chains that it falls into different threads' ranges
    good = "".join("%d 3 0 2 1 2\n%d 0\n%d 1\n%d 2\n" % (i, i, i, i) for i in range(300))
    for text in (BAD[name], good + BAD[name] + good.replace(" 3 0 2", " 3 0 2")):
        p = str(tmp_path / "bad.cdb")
        open(p, "w").write(text)
        r1 = _parse(tool, p, 1)
        for t in (2, 5, 16):
            rt = _parse(tool, p, t)
            assert (rt.returncode, rt.stderr) == (r1.returncode, r1.stderr), (name, t)
        assert r1.returncode == 3 and "FATAL" in r1.stderr, name
        with pytest.raises(maps.CDBFormatError):  # the Python loader rejects the same files
            maps.read_cdb(p)


def test_loader_picks_the_parallel_form_for_large_files(tmp_path):
    """read_pgraph itself: parallel above RAYJOIN_LOADER_MIN_BYTES, and query_exec-visible results
    do not depend on it (same .bin through the public loader)."""
    exe = os.path.join(ROOT, "rayjoin_amd", "query_exec")
    if not os.path.exists(exe):
        pytest.skip("query_exec not built")
    g = synth.lattice_map(12, 20, 8)
    p = str(tmp_path / "m.cdb")
    maps.write_cdb(p, g, "%.9f")
    outs = []
    for env in ({"RAYJOIN_LOADER_THREADS": "1"}, {"RAYJOIN_LOADER_THREADS": "6", "RAYJOIN_LOADER_MIN_BYTES": "1"}):
        o = str(tmp_path / ("s%d.bin" % len(outs)))
        r = subprocess.run([exe, "-poly1", p, "-query", "lsi", "-mode", "lbvh", "-sample", "map", "-sample_map_id", "0",
                            "-sample_rate", "1", "-seed", "1", "-sample_output", o, "-device", "99"],
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert os.path.exists(o), r.stderr  # (the run itself stops at rj_create: no GPU here; the loader ran before)
        outs.append(open(o, "rb").read())
    assert outs[0] == outs[1]
