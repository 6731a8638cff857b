#!/usr/bin/env python3
"""Where does the time of a k_lsi / k_pip launch go?  Runs the diagnostic library built by
tools/timeline/instrument.py on the headline pair (whole query map and a 1/N shard) and prints,
per kernel: the waves' busy span and mean slot occupancy, the groups' duration percentiles, how
long groups take by start time (the tail), own vs stolen (other XCD's part) groups, and the
heaviest groups with their x-extent as a fraction of the map (chain-boundary straddlers)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.environ.get("GRAFT_REPO_ROOT", ROOT)
sys.path.insert(0, ROOT)
os.environ["RAYJOIN_AMD_LIB"] = os.path.join(ROOT, "build_ab", "tl", "librayjoin_tl.so")
from rayjoin_amd import _capi, maps, synth, dist as rjd  # noqa: E402

WAVES, GROUPS = 8192, 1 << 19
M40 = (1 << 40) - 1


def pct(a):
    return [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]


def report(path, ngroups, extent):
    raw = np.fromfile(path, dtype=np.uint64)
    rec = raw[8 * WAVES:].reshape(-1, 2)[: min(ngroups, GROUPS)]
    start = (rec[:, 0] & np.uint64(M40)).astype(np.int64)
    end = ((rec[:, 1] >> np.uint64(24)) & np.uint64(M40)).astype(np.int64)
    wave = ((rec[:, 1] >> np.uint64(8)) & np.uint64(0xFFFF)).astype(np.int64)
    home = ((rec[:, 1] >> np.uint64(4)) & np.uint64(7)).astype(np.int64)
    t0 = start.min()
    gs, gd = (start - t0) / 100.0, (end - start) / 100.0
    span = float((gs + gd).max())
    nw = int(wave.max()) + 1
    w_first = np.full(nw, np.inf); w_last = np.zeros(nw); w_busy = np.zeros(nw); w_n = np.zeros(nw)
    np.minimum.at(w_first, wave, gs); np.maximum.at(w_last, wave, gs + gd); np.add.at(w_busy, wave, gd); np.add.at(w_n, wave, 1)
    live = w_n > 0
    print("    span %.1f us, %d waves with work, groups/wave %s" % (span, int(live.sum()), pct(w_n[live])))
    print("    wave's last group ends at (us)   %s" % pct(w_last[live]))
    print("    mean slot occupancy (sum of group time / waves x span): %.3f" % (gd.sum() / (nw * span)))
    print("    group duration (us)              %s; groups under 1 us (clear of the base map): %d" % (pct(gd), int((gd < 1.0).sum())))
    for lo, hi in ((0, .25), (.25, .5), (.5, .75), (.75, .9), (.9, 1.01)):
        m = (gs >= lo * span) & (gs < hi * span)
        if m.any():
            print("    started in [%.2f, %.2f) of the span: %7d groups, duration median %.1f mean %.1f p99 %.1f"
                  % (lo, hi, int(m.sum()), np.median(gd[m]), gd[m].mean(), np.percentile(gd[m], 99)))
    part = np.minimum(np.arange(len(gd)) * 8 // max(len(gd), 1), 7)
    stolen = part != home
    if stolen.any():
        print("    groups of another XCD's part (stolen): %d, mean %.1f us; own: mean %.1f us" % (int(stolen.sum()), gd[stolen].mean(), gd[~stolen].mean()))
    hv = np.argsort(gd)[-8:][::-1]
    print("    heaviest groups (id, us, x-extent / map width): %s" % [(int(g), round(float(gd[g]), 1), round(float(extent[g]), 3)) for g in hv])
    wide = extent[: len(gd)] > 0.25
    if wide.any():
        print("    groups wider than 1/4 of the map: %d, mean %.1f us (others %.1f us)" % (int(wide.sum()), gd[wide].mean(), gd[~wide].mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
    ap.add_argument("--shards", default="8,1")
    a = ap.parse_args()
    ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
    b, q = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.build_lbvh(0)
    closest = h.alloc(4 * q.n_points)
    cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
    pairs = h.alloc(8 * cap)
    width = float(q.pts[:, 0].max() - q.pts[:, 0].min())
    out = os.path.join(ROOT, "gpurun_out", "timeline")
    os.makedirs(os.path.dirname(out), exist_ok=True)

    def extents(x, lo, hi):
        x = x[lo:hi]
        n = (len(x) + 63) // 64
        pad = np.full(n * 64 - len(x), x[-1], dtype=x.dtype)
        x = np.concatenate([x, pad]).reshape(n, 64)
        return (x.max(1) - x.min(1)) / width

    for shards in [int(s) for s in a.shards.split(",")]:
        sh = rjd.shard_of(q, shards, 0)
        (e0, e1), (p0, p1) = sh["eids"], sh["points"]
        for _ in range(3):
            h.pip_query(0, 1, None, p0, p1 - p0, closest, None)
            h.lsi_query(0, 1, e0, e1, cap, pairs)
        os.environ["RJ_TIMELINE_OUT"] = out
        h.pip_query(0, 1, None, p0, p1 - p0, closest, None); pip_ms = h.last_ms(_capi.RJ_T_PIP_KERNEL)
        h.lsi_query(0, 1, e0, e1, cap, pairs); lsi_ms = h.last_ms(_capi.RJ_T_LSI_KERNEL)
        del os.environ["RJ_TIMELINE_OUT"]
        print("1/%d of the query map: %d points, %d segments" % (shards, p1 - p0, e1 - e0))
        print("  k_pip %.4f ms (instrumented)" % pip_ms)
        report(out + ".pip.bin", (p1 - p0 + 63) // 64, extents(q.pts[:, 0], p0, p1))
        print("  k_lsi %.4f ms (instrumented)" % lsi_ms)
        segx = q.segments()[:, 0] if hasattr(q, "segments") else q.pts[:, 0]
        report(out + ".lsi.bin", (e1 - e0 + 63) // 64, extents(segx, e0, e1))


if __name__ == "__main__":
    main()
