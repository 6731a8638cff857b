#!/usr/bin/env python3
"""The instrumented k_lsi (one segment per lane) on a pair, with the blocks' second order ("leaf_ysort") on and off: per
64-segment group of the query map -- leaf blocks opened, scan steps, exact tests, nodes expanded -- and where a wave's cycles
go (node expansions / leaf scans incl. the dense predicate / group head / scheduler).
usage: lsi_stats_probe.py [--base USCounty --query BlockGroup]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap)
groups = (q.n_edges + 63) // 64
for ysort in (0, 1):
    h.set_option("leaf_ysort", ysort)
    h.build_lbvh(0)
    h.set_option("stats", 0)
    ms = []
    for _ in range(4):
        n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs); ms.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
    h.set_option("stats", 1)
    h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
    s = h.last_stats()
    raw = h.last_stats_raw()
    tot = max(1, s["cyc_total"])
    print(json.dumps({"pair": a.base + " x " + a.query, "leaf_ysort": ysort, "k_lsi2_ms": round(min(ms), 4), "hits": n, "groups_of_64": groups,
                      "per_group": {"leaf_blocks": round(s["leaf_blocks"] / groups, 3), "scan_steps": round(s["leaf_box_tests"] / groups, 3),
                                    "exact_tests": round(s["exact_tests"] / groups, 3), "nodes": round(s["nodes_expanded"] / groups, 3),
                                    "children_refined": round(raw[11] / groups, 3), "children_kept": round(raw[12] / groups, 3)},
                      "steps_per_leaf_block": round(s["leaf_box_tests"] / max(1, s["leaf_blocks"]), 2),
                      "cycles_frac": {"node": round(s["cyc_node"] / tot, 3), "leaf_and_predicate": round(s["cyc_leaf"] / tot, 3),
                                      "head": round(s["cyc_drain"] / tot, 3), "sched": round(s["cyc_sched"] / tot, 3)}}))
