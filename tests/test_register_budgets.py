"""The shared schedule of a step is register arithmetic: per SIMD 6 waves of k_pip_walk2 at 56 VGPRs beside 2 of k_lsi2
at 80 make 496 of 512 (rj_kernels.hip pip_walk2_blocks_beside).  A compiler that hands either
kernel one register more -- or a source change that does -- silently costs a resident block per CU (the headline step
0.79 -> 0.83 ms), so the budgets are pinned here against the resource report of the build
(`make` writes rayjoin_amd/csrc/resource_usage.txt: -Rpass-analysis=kernel-resource-usage)."""
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPORT = os.path.join(os.path.dirname(HERE), "rayjoin_amd", "csrc", "resource_usage.txt")


def _kernels():
    if not os.path.exists(REPORT):
        pytest.skip("no resource report: build the library first (__graft_entry__.build)")
    out, name = {}, None
    for line in open(REPORT):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            out[name][m.group(1).strip()] = int(m.group(2))
    return out


def _one(kernels, fragment):
    hits = [v for k, v in kernels.items() if fragment in k]
    assert len(hits) == 1, (fragment, [k for k in kernels if fragment in k])
    return hits[0]


def test_the_shared_schedules_registers():
    k = _kernels()
    walk2 = _one(k, "k_pip_walk2E")
    lsi2 = _one(k, "k_lsi2E")
    assert walk2["VGPRs"] <= 56 and walk2["ScratchSize"] == 0, walk2
    assert lsi2["VGPRs"] <= 80 and lsi2["ScratchSize"] == 0, lsi2
    assert walk2["TotalSGPRs"] <= 96 and lsi2["TotalSGPRs"] <= 96  # (above 96 the hardware admits a block per CU fewer)
    granule = lambda v: (v + 7) // 8 * 8
    assert 6 * granule(walk2["VGPRs"]) + 2 * granule(lsi2["VGPRs"]) <= 512


def test_no_query_kernel_spills():
    k = _kernels()
    # (k_lsi2x / k_lsix: the LSI bodies instantiated without the second order of steep blocks -- maps of closed rings)
    for frag, vgprs in (("k_pip_walkILb0", 64), ("k_pip_exactE", 96), ("k_lsiILb0", 80), ("k_lsixILb0", 80), ("k_lsi2xE", 80), ("k_lsi_pointsE", 96)):
        r = _one(k, frag)
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= vgprs, (frag, r)
