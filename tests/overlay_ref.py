"""Test-side restatement of the reference's overlay host logic (test infrastructure):
WriteOutputChain, src/app/output_chain.h:42-205, driven by the oracle's results.  Independent of
the product's C++ writer (rayjoin_amd/host/output_chain.h) so the two can be diffed."""
import numpy as np


def _fmt6(v):
    return "%.6f" % v  # ofs.setf(fixed), precision(6): output_chain.h:186-187


def write_output_chain(ctx, xsects_sorted_pair, point_in_polygon_pair, path):
    """ctx: maps.Context (planar graphs + scaling); xsects_sorted_pair[im]: records ordered by
    (eid[im], distance) with mid_point_polygon_id; point_in_polygon_pair[im]: face (in the other
    map) of every vertex of map im."""
    sc = ctx.scaling
    chains_out = []  # dicts: points, left, right, other

    def flush(oc):
        pts = oc["points"]
        if pts:
            if oc["left"] * oc["other"] != 0 or oc["right"] * oc["other"] != 0:  # :58-61
                uniq = [pts[0]]
                for p in pts[1:]:  # std::unique: drop consecutive duplicates (:62-66)
                    if p != uniq[-1]:
                        uniq.append(p)
                chains_out.append(dict(points=uniq, left=oc["left"], right=oc["right"], other=oc["other"]))
            oc["points"] = []

    def xsect_point(x):  # AddXsectPoint (:33-39): unscale the stored (truncated) point
        p = sc.unscale(np.array([[x["x_num"], x["y_num"]]], dtype=np.int64))[0]
        return (float(p[0]), float(p[1]))

    for im in range(2):
        xs = xsects_sorted_pair[im]
        pip = point_in_polygon_pair[im]
        g = ctx.planar_graphs[im]
        grouped = {}
        for x in xs:  # :84-88
            grouped.setdefault(int(x["eid"][im]), []).append(x)
        for ic in range(g.n_chains):
            b, e = int(g.row_index[ic]), int(g.row_index[ic + 1])
            oc = dict(points=[], left=int(g.chains[ic, 3]), right=int(g.chains[ic, 4]), other=0)
            for pid in range(b, e):
                oc["other"] = int(pip[pid])
                oc["points"].append((float(g.points[pid, 0]), float(g.points[pid, 1])))
                if pid != e - 1:
                    lst = grouped.get(pid - ic)
                    if lst:
                        oc["points"].append(xsect_point(lst[0]))
                        for k in range(len(lst) - 1):
                            flush(oc)
                            oc["other"] = int(lst[k]["mid_point_polygon_id"])
                            oc["points"].append(xsect_point(lst[k]))
                            oc["points"].append(xsect_point(lst[k + 1]))
                        flush(oc)
                        oc["points"].append(xsect_point(lst[-1]))
            flush(oc)

    face_ids = {}

    def create_polygon(a, b):  # :146-158
        if a == 0 or b == 0:
            return 0
        if (a, b) not in face_ids:
            face_ids[(a, b)] = len(face_ids) + 1
        return face_ids[(a, b)]

    point_ids = {}
    for ch in chains_out:  # :160-185
        o = ch["other"]
        ch["left"] = create_polygon(ch["left"], o) if ch["left"] < o else create_polygon(o, ch["left"])
        ch["right"] = create_polygon(ch["right"], o) if ch["right"] < o else create_polygon(o, ch["right"])
        for p in ch["points"]:
            if p not in point_ids:
                point_ids[p] = len(point_ids)
        ch["first"] = point_ids[ch["points"][0]]
        ch["last"] = point_ids[ch["points"][-1]]
    with open(path, "w") as f:
        for i, ch in enumerate(chains_out):
            f.write("%d %d %d %d %d %d\n" % (i + 1, len(ch["points"]), ch["first"], ch["last"], ch["left"], ch["right"]))
            for p in ch["points"]:
                f.write(_fmt6(p[0]) + " " + _fmt6(p[1]) + "\n")
    return len(chains_out), len(face_ids)


def oracle_overlay(oracle, ctx, path, gsize=2048):
    """The whole polyover pipeline on the CPU oracle (-mode=grid semantics)."""
    m = [oracle.Map(ctx.maps[i].pts, ctx.maps[i].row_index, ctx.maps[i].left, ctx.maps[i].right) for i in range(2)]
    pairs = oracle.lsi_grid(m[0], m[1], gsize)["eid"]
    pip = []
    for im in range(2):  # LocateVerticesInOtherMap(im): vertices of map im in map 1-im
        eids = oracle.pip_grid(m[1 - im], 1 - im, ctx.maps[im].pts, gsize)
        pip.append(m[1 - im].face_ids(eids))
    xs = [oracle.overlay_edge_xsects(m[0], m[1], im, pairs, gsize) for im in range(2)]
    return write_output_chain(ctx, xs, pip, path), xs, pip
