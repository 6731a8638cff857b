import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth, ops
from oracle import rjoracle as O
O.lib().rjo_set_num_threads(16)
ctx = maps.Context([synth.standin("USCounty"), synth.standin("BlockGroup")]).load()
d = ops.DeviceContext(ctx).LoadToDevice(); d.BuildIndex(0)
m0 = O.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
pts = ctx.maps[1].pts[:1 << 20]
want = O.pip_grid(m0, 0, pts, 2048)
rng = np.random.default_rng(0)
perm = rng.permutation(len(pts))
pip = ops.PIPLBVH(d); pip.Init(len(pts))
pip.Query(1, query_points=pts); a = pip.get_closest_eids()
print("ordered mismatches:", int((a != want).sum()))
pip.Query(1, query_points=pts[perm]); b = pip.get_closest_eids()
bad = np.nonzero(b != want[perm])[0]
print("permuted mismatches:", len(bad), "of", len(pts))
segs = ctx.maps[0].segments()
for i in bad[:12]:
    p = pts[perm][i]; g, w = b[i], want[perm][i]
    print("pt", p, "gpu", g, "want", w, "group", i // 64, "lane", i % 64)
    for e in (g, w):
        if e != 0xFFFFFFFF:
            print("   eid", e, segs[e], O.pip_single(segs[e], p, 1))
print("bad groups:", sorted(set((bad // 64).tolist()))[:20], "lanes:", sorted(set((bad % 64).tolist()))[:64])
