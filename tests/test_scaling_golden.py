"""Row a1 pinned by the reference itself: Scaling<double,int64_t,17> (src/map/scaling.h:32-136)
compiled in place on the host produced tests/golden/scaling_ref_vectors.json (generator:
tests/golden/make_scaling_ref_vectors.py); the oracle, the Python host (rayjoin_amd/maps.py) and the
C++ host (rayjoin_amd/host/context.h) must reproduce every scaled integer and every unscaled double
bit for bit.  The vectors are the HOST build of the class (separate multiply and add); DESIGN.md
section 2 records what that means for the reference's GPU lambda."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from rayjoin_amd import maps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(ROOT, "tests", "golden", "scaling_ref_vectors.json")) as f:
        g = json.load(f)
    fh = float.fromhex
    for s in g["sets"]:
        s["bb"] = np.array([fh(v) for v in s["bb"]])
        s["xy"] = np.array([[fh(a), fh(b)] for a, b in s["xy"]])
        s["scaled"] = np.array(s["scaled"], dtype=np.int64)
        s["ints"] = np.array(s["ints"], dtype=np.int64)
        s["unscaled"] = np.array([[fh(a), fh(b)] for a, b in s["unscaled"]])
    return g


def _same_bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), np.asarray(b, dtype=np.float64).view(np.uint64))


def test_constants(vec):
    assert vec["internal_min"] == maps.INTERNAL_MIN == -(1 << 46)
    assert vec["internal_max"] == maps.INTERNAL_MAX == (1 << 46) - 1
    assert vec["internal_range"] == maps.INTERNAL_RANGE and vec["sizeof_scaling"] == 88
    assert len(vec["sets"]) >= 7 and sum(len(s["xy"]) for s in vec["sets"]) > 1000


def test_python_host_scaling_is_the_references(vec):
    for s in vec["sets"]:
        sc = maps.Scaling(tuple(s["bb"]))
        assert np.array_equal(sc.scale(s["xy"]), s["scaled"])
        assert _same_bits(sc.unscale(s["ints"]), s["unscaled"])


def test_oracle_scaling_is_the_references(oracle, vec):
    for s in vec["sets"]:
        sc = oracle.make_scaling(*s["bb"])
        assert np.array_equal(oracle.scale_points(sc, s["xy"]), s["scaled"])
        assert _same_bits(oracle.unscale_points(sc, s["ints"]), s["unscaled"])


def test_cpp_host_scaling_is_the_references(vec):
    src = os.path.join(ROOT, "tests", "hosttwin", "host_scaling.cc")
    out = os.path.join(ROOT, "tests", "hosttwin", "_build", "libhostscaling.so")
    hdr = os.path.join(ROOT, "rayjoin_amd", "host", "context.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-I", os.path.dirname(hdr), "-o", out, src])
    L = C.CDLL(out)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    L.host_scale_points.argtypes = [f64, f64, C.c_uint64, i64]
    L.host_unscale_points.argtypes = [f64, i64, C.c_uint64, f64]
    L.host_scaling_consts.argtypes = [i64]
    c = np.zeros(3, dtype=np.int64)
    L.host_scaling_consts(c)
    assert c.tolist() == [vec["internal_min"], vec["internal_max"], vec["internal_range"]]
    for s in vec["sets"]:
        got = np.zeros(s["xy"].shape, dtype=np.int64)
        L.host_scale_points(np.ascontiguousarray(s["bb"]), np.ascontiguousarray(s["xy"]), len(s["xy"]), got)
        assert np.array_equal(got, s["scaled"])
        back = np.zeros(s["ints"].shape, dtype=np.float64)
        L.host_unscale_points(np.ascontiguousarray(s["bb"]), np.ascontiguousarray(s["ints"]), len(s["ints"]), back)
        assert _same_bits(back, s["unscaled"])


def test_live_reference_scaling_when_available(oracle):
    R = oracle.ref_lib()
    if R is None or not hasattr(R, "ref_scale_points"):
        pytest.skip("oracle/_ref/liblsi_ref.so (with Scaling) not built")
    rng = np.random.default_rng(5)
    for _ in range(20):
        lo = rng.uniform(-1e4, 1e4, 2)
        ext = 10.0 ** rng.uniform(-3, 4, 2)
        bb = np.array([lo[0], lo[1], lo[0] + ext[0], lo[1] + ext[1]])
        xy = np.ascontiguousarray(np.stack([rng.uniform(bb[0], bb[2], 2000), rng.uniform(bb[1], bb[3], 2000)], 1))
        want = np.zeros(xy.shape, dtype=np.int64)
        R.ref_scale_points(bb, xy, len(xy), want)
        assert np.array_equal(maps.Scaling(tuple(bb)).scale(xy), want)
        assert np.array_equal(oracle.scale_points(oracle.make_scaling(*bb), xy), want)


def test_fused_scaling_is_one_exact_fma(vec):
    """fused=True (rj_scale_points / maps.Scaling(bb, fused=True) / query_exec -scale_fma): x*rx + dx
    rounded ONCE -- checked against exact rational arithmetic -- which is what nvcc's -fmad=true makes
    of the reference's device lambda (src/map/map.h:171-180); it differs from the unfused golden values
    by one unit for a fraction of a percent of the points, never by more."""
    from fractions import Fraction
    from rayjoin_amd import _capi
    s = vec["sets"][0]
    bb, xy = tuple(s["bb"]), s["xy"]
    sc = maps.Scaling(bb)
    assert np.array_equal(_capi.scale_points(bb, xy, fused=False), s["scaled"])
    got = _capi.scale_points(bb, xy, fused=True)
    assert np.array_equal(maps.Scaling(bb, fused=True).scale(xy), got)
    for (x, y), (gx, gy) in zip(xy[:300], got[:300]):
        wx = float(Fraction(float(x)) * Fraction(float(sc.rx)) + Fraction(float(sc.dx)))  # correctly rounded
        wy = float(Fraction(float(y)) * Fraction(float(sc.ry)) + Fraction(float(sc.dy)))
        assert (int(wx), int(wy)) == (int(gx), int(gy))
    rng = np.random.default_rng(9)
    big = np.stack([rng.uniform(bb[0], bb[2], 100000), rng.uniform(bb[1], bb[3], 100000)], 1)
    a, b = _capi.scale_points(bb, big, False), _capi.scale_points(bb, big, True)
    assert np.abs(a - b).max() == 1 and 0.0005 < (a != b).any(axis=1).mean() < 0.02
