"""The traversal stacks of k_lsi / k_pip (rayjoin_amd/csrc/rj_device.h) must hold the worst case of
every tree rj_build_lbvh accepts: every child of every node overlaps the query group.  This
simulates the two stack disciplines on such a tree (lazily: nodes are (level, index) pairs) and
checks the compiled capacities against the maxima it sees and against the closed forms quoted in
the header.  The reference's own stack is 64 entries, unchecked (deps/lbvh/lbvh/query.cuh:16)."""
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(os.path.dirname(HERE), "rayjoin_amd", "csrc", "rj_device.h")


def _consts():
    src = open(HDR).read()
    env = {}
    for name in ("kMaxLevels", "kMaxTop", "kStackEntries", "kPipStack"):
        m = re.search(r"constexpr int %s = ([^;]+);" % name, src)
        assert m, name
        env[name] = int(eval(m.group(1), {}, env))
    return env


def _simulate(top, k_top, pop_two, max_steps):
    """Depth-first over a complete 64-ary tree with `top` levels above the segments and k_top
    top-level nodes, all boxes overlapping.  Entries are levels only (indices do not matter).
    pop_two = k_lsi (two entries per step, a then b, each pushing its 64 children)."""
    stack = [top] * k_top
    peak = len(stack)
    steps = 0
    while stack and steps < max_steps:
        steps += 1
        popped = [stack.pop()]
        if pop_two and stack:
            popped.append(stack.pop())
        for lvl in popped:
            if lvl > 1:
                stack.extend([lvl - 1] * 64)
            peak = max(peak, len(stack))
    return peak, not stack


@pytest.mark.parametrize("top", [1, 2, 3, 4, 5])
def test_stack_capacities_cover_the_worst_case(top):
    c = _consts()
    assert c["kMaxTop"] == c["kMaxLevels"] - 1 >= 5  # 2^32 segments need levels 1..5
    if top > c["kMaxTop"]:
        pytest.skip("deeper than any accepted tree")
    # small trees run to completion, deep ones for a prefix that contains the first full descent
    # (where the maximum is reached: every later state is a suffix of an earlier one)
    budget = 400_000
    for k_top in (1, 2, 63, 64):
        peak2, done2 = _simulate(top, k_top, True, budget)
        peak1, done1 = _simulate(top, k_top, False, budget)
        assert peak2 <= k_top + 126 * (top - 1) <= c["kStackEntries"]
        assert peak1 <= k_top + 63 * (top - 1) <= c["kPipStack"]
        if top <= 3:
            assert done1 and done2
    # the closed forms are tight for a full top level
    assert _simulate(top, 64, False, budget)[0] == 64 + 63 * (top - 1)
    if top > 1:
        assert _simulate(top, 64, True, budget)[0] == 64 + 126 * (top - 1)


def test_lds_budget_keeps_the_occupancy():
    """k_lsi: stack + 2 x 128 pair buffers per wave; k_pip: 16-byte entries + candidate lists
    (+ the block's four 8-byte chunk ranges).
    160 KiB per CU must hold 7 (k_lsi) and 6 (k_pip) four-wave blocks -- the register-limited
    occupancies in DESIGN.md."""
    c = _consts()
    src = open(os.path.join(os.path.dirname(HDR), "rj_kernels.hip")).read()
    plist = int(re.search(r"constexpr int kPipList = (\d+);", src).group(1))
    lsi_block = 4 * (4 * c["kStackEntries"] + 2 * 128 * 8)
    pip_block = 4 * (16 * c["kPipStack"] + plist * 64 * 4) + 4 * 8
    waves = int(re.search(r"#define RJ_PIP_WAVES (\d+)", src).group(1))
    assert waves == 4 and "__shared__ unsigned long long ranges[kPipWaves];" in src
    assert 7 * lsi_block <= 160 * 1024
    assert 6 * pip_block <= 160 * 1024
