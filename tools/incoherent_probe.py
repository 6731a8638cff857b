#!/usr/bin/env python3
"""How do spatially incoherent query sets (GenerateLSIQueries / GeneratePIPQueries, or a shuffled
point set) behave?  GPU only."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ctx = maps.Context([synth.standin("USCounty"), synth.standin("BlockGroup")]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right)
rng = np.random.default_rng(1)
out = {}
def pip(pts, tag):
    d = h.alloc(16 * len(pts)).from_host(pts); c = h.alloc(4 * len(pts))
    for _ in range(3):
        h.pip_query(0, 1, d, 0, len(pts), c, None)
    out[tag] = round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 3)
    d.free(); c.free()
h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
n = 1 << 22
pip(q.pts[:n], "pip_4M_chain_order_ms")
pip(q.pts[:n][rng.permutation(n)], "pip_4M_shuffled_ms")
pip(synth.generate_pip_queries(ctx.bb, ctx.scaling, n, 3), "pip_4M_uniform_random_ms")
for gen_n, gen_t in ((10000, 0.1), (1 << 20, 0.1), (1 << 20, 2.0)):
    qm = synth.generate_lsi_queries(ctx.bb, ctx.scaling, gen_n, gen_t, 5)
    h.upload_map(1, qm.pts, qm.row_index, qm.left, qm.right)
    cap = 4 * gen_n + 100000
    p = h.alloc(8 * cap)
    for _ in range(3):
        nx = h.lsi_query(0, 1, 0, qm.n_edges, cap, p)
    out["lsi_random_n%d_t%g_ms" % (gen_n, gen_t)] = [round(h.last_ms(_capi.RJ_T_LSI_KERNEL), 3), int(nx)]
    p.free()
print(json.dumps(out, indent=1))
