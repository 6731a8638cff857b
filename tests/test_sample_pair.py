"""BASELINE.json configs[0]: the sample CDB pair end to end (the reference's own sample pair is
missing from the snapshot; tests/golden/sample_pair/ plays its role, like test/test_overlay.sh's
dataset + answer file).  CPU: loader + oracle reproduce the committed answers.  GPU: query_exec
-mode=lbvh writes exactly the answer files."""
import os
import subprocess

import numpy as np
import pytest

from rayjoin_amd import maps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "golden", "sample_pair")
EXE = os.path.join(ROOT, "rayjoin_amd", "query_exec")


def _ctx():
    return maps.Context([maps.read_cdb(os.path.join(D, "map0.cdb")), maps.read_cdb(os.path.join(D, "map1.cdb"))]).load()


def test_oracle_reproduces_committed_answers(oracle):
    ctx = _ctx()
    m0 = oracle.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = oracle.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
    want = np.loadtxt(os.path.join(D, "lsi_answer.txt"), dtype=np.int64).reshape(-1, 4)
    xs = oracle.lsi_grid(m0, m1, 2048)  # -grid_size default, src/flags.cc:6
    assert len(want) > 100
    assert np.array_equal(xs["eid"].astype(np.int64), want[:, :2])
    assert np.array_equal(xs["x_num"], want[:, 2]) and np.array_equal(xs["y_num"], want[:, 3])
    for g in (64, 15000 // 8):  # results do not depend on the grid resolution
        assert np.array_equal(oracle.lsi_grid(m0, m1, g)["eid"].astype(np.int64), want[:, :2])
    wp = np.loadtxt(os.path.join(D, "pip_answer.txt"), dtype=np.int64).reshape(-1, 2)
    eids = oracle.pip_grid(m0, 0, ctx.maps[1].pts, 2048)
    assert np.array_equal(eids.astype(np.int64), wp[:, 0])
    assert np.array_equal(m0.face_ids(eids).astype(np.int64), wp[:, 1])


@pytest.mark.gpu
def test_query_exec_matches_answer_files(tmp_path):
    for query, ans, cols in (("lsi", "lsi_answer.txt", 4), ("pip", "pip_answer.txt", 2)):
        out = str(tmp_path / (query + ".txt"))
        r = subprocess.run([EXE, "-poly1", os.path.join(D, "map0.cdb"), "-poly2", os.path.join(D, "map1.cdb"),
                            "-query", query, "-mode", "lbvh", "-warmup", "1", "-repeat", "1", "-output", out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.loadtxt(out, dtype=np.int64).reshape(-1, cols)
        want = np.loadtxt(os.path.join(D, ans), dtype=np.int64).reshape(-1, cols)
        assert np.array_equal(got, want), query
