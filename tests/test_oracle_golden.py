"""Pin the CPU oracle (oracle/rj_oracle.c) against vectors produced by the reference's own
predicate headers (tests/golden/lsi_ref_vectors.json) and the known answers of SURVEY.md 8c."""
import numpy as np
import pytest


def test_survey_known_answers(oracle):
    # outputs observed from the reference's lsi.h compiled on the host
    r = oracle.intersect_point([0, 0, 10, 10], [0, 10, 10, 0])
    assert r["x"] == (5, 1) and r["y"] == (5, 1)
    r = oracle.intersect_point([0, 0, 10, 7], [0, 10, 10, 0])
    assert r["x"] == (100, 17) and r["y"] == (70, 17) and r["stored"] == (5, 4)
    # T-junction: operand order matters (simulation of simplicity)
    assert oracle.intersect_test([0, 0, 10, 0], [5, 0, 5, 7]) == 0
    assert oracle.intersect_test([5, 0, 5, 7], [0, 0, 10, 0]) == 1
    assert oracle.XSECT_DTYPE.itemsize == 48


def test_golden_pairs(oracle, golden):
    g = golden["gsize"]
    n_hit = 0
    for rec in golden["pairs"]:
        assert oracle.intersect_test(rec["s1"], rec["s2"]) == rec["hit"], rec
        r = oracle.intersect_point(rec["s1"], rec["s2"], g)
        assert (r is not None) == bool(rec["hit"])
        if r:
            n_hit += 1
            assert [str(r["x"][0]), str(r["x"][1])] == rec["x"], rec
            assert [str(r["y"][0]), str(r["y"][1])] == rec["y"], rec
            assert list(r["stored"]) == rec["stored"], rec
            assert list(r["cell"]) == rec["cell"], rec
    assert n_hit > 100


def test_golden_cells(oracle, golden):
    L = oracle.lib()
    for rec in golden["cell_of_int"]:
        assert L.rjo_cell_of_int(rec["g"], rec["v"]) == rec["cell"]
    for rec in golden["cell_of_double"]:
        assert L.rjo_cell_of_double(rec["g"], rec["v"]) == rec["cell"]


def test_live_reference_agrees_when_available(oracle):
    """When oracle/_ref is present (authoring container, or shipped prebuilt), fuzz the oracle
    against the reference headers directly."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref/liblsi_ref.so not built")
    rng = np.random.default_rng(99)
    for span in (3, 1 << 20, 1 << 38):
        segs = rng.integers(-span, span + 1, size=(4000, 8))
        for r in segs:
            a, b = r[:4], r[4:]
            if (a[0] == a[2] and a[1] == a[3]) or (b[0] == b[2] and b[1] == b[3]):
                continue
            assert oracle.intersect_test(a, b) == oracle.intersect_test(a, b, "ref")
            assert oracle.intersect_point(a, b, 2048) == oracle.intersect_point(a, b, 2048, "ref")
