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


def _simulate_partial(top, k_top, pop_two, fanout, rng, steps):
    """The same disciplines with ARBITRARY subsets of the children pushed (what the per-lane culling
    produces): a node of level l > 1 pushes between 0 and `fanout` children.  Returns the peak, which
    must stay below the all-children closed form scaled to this fanout."""
    stack = [top] * k_top
    peak = len(stack)
    for _ in range(steps):
        if not stack:
            break
        popped = [stack.pop()]
        if pop_two and stack:
            popped.append(stack.pop())
        for lvl in popped:
            if lvl > 1:
                stack.extend([lvl - 1] * rng.randint(0, fanout))
            peak = max(peak, len(stack))
    return peak


def _exhaustive_partial(top, k_top, pop_two, fanout):
    """Adversary search (small fanout): the largest stack any sequence of child-subset sizes can reach.
    State = the stack as a tuple of levels; every node of level > 1 may push 0..fanout children."""
    import functools

    @functools.lru_cache(maxsize=None)
    def worst(stack):
        if not stack:
            return 0
        st = list(stack)
        popped = [st.pop()]
        if pop_two and st:
            popped.append(st.pop())
        best = len(st)
        # choose the pushes of the popped entries one after the other
        def rec(i, cur):
            nonlocal best
            if i == len(popped):
                best = max(best, len(cur), worst(tuple(cur)))
                return
            if popped[i] <= 1:
                rec(i + 1, cur)
                return
            for k in range(fanout + 1):
                nxt = cur + [popped[i] - 1] * k
                best = max(best, len(nxt))
                rec(i + 1, nxt)
        rec(0, st)
        return best
    return max(k_top, worst(tuple([top] * k_top)))


@pytest.mark.parametrize("pop_two", [False, True])
def test_partial_child_subsets_stay_below_the_bound(pop_two):
    """The closed forms assume every child is pushed; the kernels push arbitrary SUBSETS (per-lane culling,
    refine masks), and k_lsi's two-pops-per-step discipline then mixes levels on the stack.  The bound
    k_top + (fanout - 1 | 2 fanout - 2) (top - 1) must hold for every subset choice: exhaustively for
    fanout 2 (and 3 on shallow trees), by random search for fanout 3-5 and for the real fanout 64."""
    import random
    per = (lambda f: 2 * f - 2) if pop_two else (lambda f: f - 1)
    for fanout, top, k_top in ((2, 2, 2), (2, 3, 2), (2, 4, 2), (2, 4, 1), (3, 2, 3), (3, 3, 2)):
        assert _exhaustive_partial(top, k_top, pop_two, fanout) <= k_top + per(fanout) * (top - 1)
    rng = random.Random(7)
    for fanout in (3, 4, 5, 64):
        for top in (2, 3, 4, 5):
            bound = fanout + per(fanout) * (top - 1)
            for _ in range(40 if fanout < 64 else 6):
                assert _simulate_partial(top, fanout, pop_two, fanout, rng, 20000) <= bound
    c = _consts()
    assert 64 + per(64) * (c["kMaxTop"] - 1) <= (c["kStackEntries"] if pop_two else c["kPipStack"])


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
    # the walk: a stack cut at kWalkStack entries whatever the height (a group that needs more leaves the walk for k_pip,
    # whose stack IS the worst case) + kWalkList candidate slots per point; eight four-wave blocks of either kernel per CU
    wstack = int(re.search(r"constexpr int kWalkStack = (\d+);", src).group(1))
    wlist = int(re.search(r"#define RJ_WALK_LIST (\d+)", open(HDR).read()).group(1))
    assert 64 < wstack <= c["kPipStack"]
    assert 8 * (4 * (16 * wstack + wlist * 256) + 32) <= 160 * 1024       # k_pip_walk
    assert 8 * (4 * (16 * wstack + 2 * wlist * 256) + 32) <= 160 * 1024   # k_pip_walk2
    assert "if (sp + __popcll(m) > stack_cap) { ovf = true; break; }" in src
