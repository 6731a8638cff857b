"""The device predicate source (rayjoin_amd/csrc/rj_predicates.h) compiled for the host -- test
only -- against the reference's golden vectors and the oracle: the sign-only intersect test
(edge_side), the intersection point, and the two-product PIP numerator must be the reference's
functions, not merely agree with them on the maps the GPU tests happen to use."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hosttwin", "twin.cc")
OUT = os.path.join(ROOT, "tests", "hosttwin", "_build", "libtwin.so")


@pytest.fixture(scope="module")
def twin():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdr = os.path.join(ROOT, "rayjoin_amd", "csrc", "rj_predicates.h")
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fwrapv",
                               "-I", os.path.dirname(hdr), "-o", OUT, SRC])
    L = C.CDLL(OUT)
    i64p = C.POINTER(C.c_int64)
    L.twin_lsi_test.argtypes = [i64p, i64p]
    L.twin_lsi_stored.argtypes = [i64p, i64p, i64p]
    L.twin_pip_eval.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.twin_i128_to_double.argtypes = [C.c_int64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.twin_rat_make.argtypes = [C.c_int64, C.c_uint64, C.c_int64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.twin_lsi_stored_fast.argtypes = [i64p, i64p, i64p]
    L.twin_lsi_coord.argtypes = [C.c_int64, C.c_uint64, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, i64p, i64p]
    L.twin_lsi_stored_fuzz.argtypes = [i64p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.twin_pip_better.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_double, C.c_double, C.c_uint32, C.c_int]
    return L


def _p(a):
    return np.ascontiguousarray(a, dtype=np.int64).ctypes.data_as(C.POINTER(C.c_int64))


def test_device_lsi_predicate_on_reference_golden_pairs(twin):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lsi_ref_vectors.json")))
    out = np.zeros(2, dtype=np.int64)
    n_hit = 0
    for rec in gold["pairs"]:
        s1, s2 = np.array(rec["s1"], dtype=np.int64), np.array(rec["s2"], dtype=np.int64)
        assert twin.twin_lsi_test(_p(s1), _p(s2)) == rec["hit"], rec
        if rec["hit"]:
            twin.twin_lsi_stored(_p(s1), _p(s2), _p(out))
            assert out.tolist() == rec["stored"], rec
            n_hit += 1
    assert n_hit > 100


@pytest.mark.parametrize("span,extreme", [(5, False), (40, False), (7, True), (1 << 44, False)])
def test_device_predicates_fuzz_against_oracle(twin, oracle, span, extreme):
    rng = np.random.default_rng(span % 1000 + extreme)
    off = (1 << 46) - span - 1 if extreme else 0
    segs = rng.integers(-span, span + 1, size=(4000, 4), dtype=np.int64) + off
    for k in range(0, len(segs) - 1, 2):
        s1, s2 = segs[k], segs[k + 1]
        if (s1[0] == s1[2] and s1[1] == s1[3]) or (s2[0] == s2[2] and s2[1] == s2[3]):
            continue  # zero-length edges cannot come out of a CDB file (planar_graph.h:100-105)
        assert twin.twin_lsi_test(_p(s1), _p(s2)) == oracle.intersect_test(s1.tolist(), s2.tolist())
    yy, sl = C.c_double(), C.c_double()
    pts = rng.integers(-span, span + 1, size=(2000, 2), dtype=np.int64) + off
    for k in range(2000):
        s = segs[k]
        if s[0] == s[2]:
            continue  # vertical: never a candidate (and its slope is never asked for)
        for q in (0, 1):
            got = twin.twin_pip_eval(_p(s), int(pts[k, 0]), int(pts[k, 1]), q, C.byref(yy), C.byref(sl))
            want, wyy = oracle.pip_single(s.tolist(), pts[k].tolist(), q)
            assert got == want, (s, pts[k], q)
            if got:
                assert yy.value == wyy  # bit-equal doubles


def test_device_pip_total_order(twin):
    # pip.h:73-95 as a total order: lower y wins; ties: slope rule by query map id, then eid
    assert twin.twin_pip_better(1.0, 0.0, 5, 2.0, 0.0, 1, 1) == 1
    assert twin.twin_pip_better(2.0, 0.0, 5, 1.0, 0.0, 1, 0) == 0
    assert twin.twin_pip_better(1.0, 2.0, 5, 1.0, 1.0, 1, 1) == 1 and twin.twin_pip_better(1.0, 2.0, 5, 1.0, 1.0, 1, 0) == 0
    assert twin.twin_pip_better(1.0, 1.0, 5, 1.0, 1.0, 9, 1) == 1 and twin.twin_pip_better(1.0, 1.0, 5, 1.0, 1.0, 9, 0) == 0
    assert twin.twin_pip_better(1.0, 1.0, 9, 1.0, 1.0, 5, 0) == 1


def test_custom_int128_to_double_is_the_compilers(twin):
    """i128_to_double (top 64 bits + sticky) == (double)(__int128), bit for bit: random magnitudes
    of every width up to 127 bits, exact ties at the rounding position, all-ones patterns."""
    rng = np.random.default_rng(12)
    a, b = C.c_double(), C.c_double()
    vals = []
    for bits in range(1, 128):
        for _ in range(40):
            vals.append(int(rng.integers(0, 1 << 62)) << max(0, bits - 62) | int(rng.integers(0, 1 << 62)) if bits > 62
                        else int(rng.integers(0, 1 << bits)))
        top = 1 << (bits - 1)
        vals += [top, top - 1, top + 1, (1 << bits) - 1]
        if bits > 54:  # exact half-way cases and their neighbours at the 53-bit rounding position
            half = 1 << (bits - 54)
            for k in (1, 2, 3):
                base = top + (k << (bits - 53))
                vals += [base + half, base + half - 1, base + half + 1]
    for v in vals:
        for sgn in (1, -1):
            w = (sgn * v) & ((1 << 128) - 1)
            hi = (w >> 64) - (1 << 64) if (w >> 64) >= (1 << 63) else (w >> 64)
            twin.twin_i128_to_double(hi, w & ((1 << 64) - 1), C.byref(a), C.byref(b))
            assert a.value == b.value and np.signbit(a.value) == np.signbit(b.value), (sgn * v, a.value, b.value)


def test_rational_simplify_is_exact(twin):
    """rat_make (Euclid steps with a double-estimated quotient, 64-bit binary GCD, division by a
    modular inverse) == the reference's rational(num, den).simplify() (rational.h:87-90,198-203:
    Euclid gcd with '%', then num = sign(den) num / |g|, den = |den| / |g|) in exact integers:
    map-like magnitudes (2^110 over 2^65), shared factors of every size, powers of two, extremes."""
    import math
    rng = np.random.default_rng(21)
    out = (C.c_uint64 * 4)()
    M = 1 << 128

    def bits(n):
        return int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62) | (int(rng.integers(0, 16)) << 124) if n > 62 \
            else int(rng.integers(0, 1 << max(1, n)))

    cases = []
    for _ in range(3000):
        nb, db, gb = int(rng.integers(1, 127)), int(rng.integers(1, 127)), int(rng.integers(0, 70))
        g = (bits(gb) % (1 << gb) if gb else 1) or 1
        n = (bits(nb) % (1 << nb)) // g * g
        d = ((bits(db) % (1 << db)) // g * g) or g
        if n >= 1 << 126 or d >= 1 << 126:
            continue
        cases.append((n * int(rng.choice([-1, 1])), d * int(rng.choice([-1, 1]))))
    for k in range(1, 126):  # powers of two, near-powers, equal operands, zero numerators
        cases += [(1 << k, 1 << (k // 2)), ((1 << k) - 1, (1 << k) - 1), (0, (1 << k) + 1), (-(1 << k), 3 << (k // 3)),
                  ((1 << k) * 3, (1 << k) * 5), (3 ** 40 * (1 << (k % 60)), -(3 ** 25) * (1 << (k % 50)))]
    cases += [((1 << 127) - 1, 1), (-(1 << 127) + 1, -1), (5, -(1 << 126)), (1 << 110, (1 << 65) + 1)]
    for n, d in cases:
        if abs(n) >= 1 << 127 or abs(d) >= 1 << 127 or d == 0:
            continue
        g = math.gcd(abs(n), abs(d))
        wn, wd = (n if d > 0 else -n) // g, abs(d) // g
        nn, dd = n % M, d % M
        hi = lambda v: (v >> 64) - (1 << 64) if (v >> 64) >= (1 << 63) else (v >> 64)
        twin.twin_rat_make(hi(nn), nn & ((1 << 64) - 1), hi(dd), dd & ((1 << 64) - 1), out)
        gn, gd = (out[0] << 64) | out[1], (out[2] << 64) | out[3]
        gn = gn - M if gn >= 1 << 127 else gn
        assert (gn, gd) == (wn, wd), (n, d)


def _halves(v):
    w = v % (1 << 128)
    hi = w >> 64
    return (hi - (1 << 64) if hi >= 1 << 63 else hi), w & ((1 << 64) - 1)


def test_gcd_free_store_one_coordinate(twin):
    """lsi_coord_fast against rat_make + clamp + (int64_t) double quotient on constructed rationals num / den =
    q + r / D: remainders at and next to 0 and D (where the rounded quotient of the simplified rational can land on the
    neighbouring integer -- the fast path must then either decline or agree), common factors of every size (the gcd
    changes what is rounded), both signs of num and den, clamps that cut below, above and exactly at q."""
    import random
    rng = random.Random(33)
    fast, slow = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
    decided = declined = 0
    for it in range(40000):
        db = rng.randrange(1, 80)
        D = rng.randrange(1 << (db - 1), 1 << db)
        qb = rng.randrange(0, 47)
        q = rng.randrange(-(1 << qb), (1 << qb) + 1)
        kind = it % 8
        if kind == 0:
            r = 0
        elif kind == 1:
            r = min(D - 1, rng.randrange(0, 4))
        elif kind == 2:
            r = max(0, D - 1 - rng.randrange(0, 4))
        elif kind == 3:  # right at the declared margin |num| 2^-51
            r = min(D - 1, max(0, ((abs(q) * D) >> 51) + rng.randrange(-2, 3)))
        elif kind == 4:
            r = max(0, D - 1 - max(0, ((abs(q) * D) >> 51) + rng.randrange(-2, 3)))
        else:
            r = rng.randrange(0, D)
        g = 1 if it % 3 == 0 else rng.randrange(1, 1 << rng.randrange(1, 40))
        num, den = (q * D + r) * g, D * g
        if abs(num) >= 1 << 126 or den >= 1 << 126:
            continue
        if rng.randrange(0, 2):
            num, den = -num, -den
        lo = q - rng.randrange(0, 3) if it % 5 else q + 1 + rng.randrange(0, 5)
        hi = max(lo, q + rng.randrange(0, 3) if it % 7 else q - 1 - rng.randrange(0, 5))
        ok = twin.twin_lsi_coord(*_halves(num), *_halves(den), lo, hi, _p(fast), _p(slow))
        if ok:
            decided += 1
            assert fast[0] == slow[0], (num, den, lo, hi, int(fast[0]), int(slow[0]))
        else:
            declined += 1
    assert decided > 25000 and declined > 500  # both legs exercised


@pytest.mark.parametrize("center_bits,box_bits", [(0, 3), (0, 12), (44, 6), (44, 20), (44, 30), (46, 10), (30, 30)])
def test_gcd_free_store_on_crossing_segments(twin, center_bits, box_bits):
    """lsi_stored_fast against lsi_point + the narrowing store on random predicate-true pairs: lattice-sized boxes (exact
    hits on vertices and T-junctions: r == 0), map-like magnitudes (2^44 with edges of 2^6 .. 2^30), the 2^46 corner."""
    rng = np.random.default_rng(center_bits * 100 + box_bits)
    n = 400000
    c = rng.integers(-(1 << center_bits), (1 << center_bits) + 1, size=(n, 1, 2), dtype=np.int64) if center_bits else np.zeros((n, 1, 2), dtype=np.int64)
    if center_bits == 46:
        c = np.sign(c) * ((1 << 46) - (1 << box_bits) - 2)
    segs = (c + rng.integers(-(1 << box_bits), (1 << box_bits) + 1, size=(n, 4, 2), dtype=np.int64)).reshape(n, 8)
    ok = ((segs[:, 0] != segs[:, 2]) | (segs[:, 1] != segs[:, 3])) & ((segs[:, 4] != segs[:, 6]) | (segs[:, 5] != segs[:, 7]))
    segs = np.ascontiguousarray(segs[ok])
    counts = (C.c_uint64 * 3)()
    twin.twin_lsi_stored_fuzz(_p(segs), len(segs), counts)
    hits, fast, wrong = counts[0], counts[1], counts[2]
    assert hits > 20000 and wrong == 0
    print("center 2^%d box 2^%d: %d hits, %.2f %% decided without the gcd" % (center_bits, box_bits, hits, 100.0 * fast / hits))
    # the point of it: the gcd is the exception (the margin is |v| 2^-51 on either side of an integer: 2 x 2^-5 per
    # coordinate at the 2^46 corner, 2 x 2^-7 at 2^44)
    assert fast > (0.85 if center_bits == 46 else 0.95) * hits, (hits, fast)
