#!/usr/bin/env python3
"""bench.py -- LSI + PIP query throughput of the MI355X-native path (BASELINE.json metric).

A step = one LSI Query (all query-map segments against the indexed base map: queue clear,
traversal + predicate kernel, the 48-byte Intersection record of every hit, count read-back, sync
-- what the reference times, src/run_query.cu:297-303 around src/app/lsi_lbvh.h:27-98)
followed by one PIP Query (every vertex of the query map, src/run_query.cu:346,441-457).
Workload (N=1): BASELINE.json configs[1], USCounty (base, 7.1 M segments) |><| BlockGroup
(query, 28.8 M segments), as synthetic stand-ins of those sizes (the real files are
not obtainable).  Inputs are resident in HBM before the timed region.  Index build is timed
separately and reported, never part of `value`.

N>1 (torchrun, one rank per GPU): the query map is sharded by contiguous chain ranges balanced
by edge count, the base map + LBVH are replicated.  The one real exchange of a step is the RCCL
all-gather-v of the intersection queues (every rank needs the pairs that touch ITS edges of either
map).  PIP results are per-vertex properties of the query map's chains and are consumed by the
owner of that chain range (the overlay uses them per edge of the same map,
src/app/map_overlay_lbvh.h:215-236), so by default they stay sharded; --gather-pip adds the
119 MB all-gather of closest-eid queues to every step.  Total work is fixed: "strong" scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)    # the reference's -repeat=5
    ap.add_argument("--warmup", type=int, default=8)   # (the reference: -warmup=5; the first six steps also pick the kernel schedule)
    ap.add_argument("--base", default="USCounty")
    ap.add_argument("--query", default="BlockGroup")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink both stand-ins (debug only)")
    ap.add_argument("--xsect-factor", type=float, default=0.1, help="queue capacity factor (expr/env.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=float, default=1.0)
    ap.add_argument("--check", action="store_true", help="size-independent result checks after timing")
    ap.add_argument("--gather-pip", action="store_true", help="N>1: also all-gather the PIP result queues every step")
    ap.add_argument("--serial-kernels", action="store_true", help="run the PIP kernel after the LSI kernel instead of beside it")
    ap.add_argument("--emulate-shard", type=int, default=0, metavar="N",
                    help="diagnostic, 1 GPU: time rank 0's shard of an N-way run (no exchange); the line is marked as such")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0, collectives over gloo")
    return ap.parse_args()


def cpu_baseline(args, ctx):
    """CPU restatement of -mode=grid (the oracle, kind 'port') timed beside the GPU numbers, on
    all host cores.  Default sample = the whole workload (about 10-20 s of query work on 16
    cores); --cpu-scale < 1 regenerates both stand-ins at that fraction of the lattice resolution
    (same per-segment geometry, ~scale^2 of the segments).  Query time only, like the GPU side."""
    from oracle import rjoracle as O
    from rayjoin_amd import maps, synth
    # all host cores this process may use, capped: the GPU box is shared
    O.lib().rjo_set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    if args.cpu_scale != 1.0:
        g0 = synth.standin(args.base, args.cpu_scale * args.scale)
        g1 = synth.standin(args.query, args.cpu_scale * args.scale)
        ctx = maps.Context([g0, g1]).load()
    m0 = O.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = O.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
    L = O.lib()
    gsize = 2048  # src/flags.cc:6 default
    t0 = time.perf_counter()
    grid = L.rjo_grid_build(m0.h, m1.h, gsize)
    t_build = time.perf_counter() - t0
    cap = int(0.5 * (m0.ne + m1.ne)) + 1024
    out = np.zeros(cap, dtype=O.XSECT_DTYPE)
    t0 = time.perf_counter()
    n = L.rjo_lsi_grid(m0.h, m1.h, grid, out.ctypes.data, cap)
    t_lsi = time.perf_counter() - t0
    L.rjo_grid_free(grid)
    grid = L.rjo_grid_build(m0.h, None, gsize)
    pts = ctx.maps[1].pts
    res = np.empty(pts.shape[0], dtype=np.uint32)
    t0 = time.perf_counter()
    L.rjo_pip_grid(m0.h, 0, grid, pts, pts.shape[0], res)
    t_pip = time.perf_counter() - t0
    L.rjo_grid_free(grid)
    return {
        "value": round(m1.ne / (t_lsi + t_pip) / 1e6, 4), "unit": "M query segments/s (LSI+PIP)",
        "cores": O.num_threads(), "kind": "port",
        "sample": "%s x %s stand-ins at %.3g of the lattice resolution (1 = the whole workload): %d base / %d query segments, "
                  "%d intersections; grid_size %d; lsi %.1f ms, pip %.1f ms (grid build %.1f ms not counted)"
                  % (args.base, args.query, args.cpu_scale * args.scale, m0.ne, m1.ne, n, gsize,
                     t_lsi * 1e3, t_pip * 1e3, t_build * 1e3),
        "lsi_msegs_per_s": round(m1.ne / t_lsi / 1e6, 4),
        "pip_mpoints_per_s": round(pts.shape[0] / t_pip / 1e6, 4),
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path is the only compute path (no CPU fallback)")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from rayjoin_amd import _capi, maps, synth

    # ---- synthetic workload (identical on every rank: seeded) -------------------------------
    t0 = time.perf_counter()
    g0 = synth.standin(args.base, args.scale)
    g1 = synth.standin(args.query, args.scale)
    ctx = maps.Context([g0, g1]).load()
    base, query = ctx.maps[0], ctx.maps[1]
    t_gen = time.perf_counter() - t0
    n_r, n_s, n_p = base.n_edges, query.n_edges, query.n_points

    h = _capi.Handle(local_rank)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    t_upload = time.perf_counter() - t0
    h.build_lbvh(0)
    h.build_lbvh(0)  # second build = steady-state allocator
    if not args.serial_kernels:
        # LSI and PIP of a step are independent: "auto" tries them one after the other, sharing the chip and beside
        # each other on full grids during the first six steps and keeps the fastest schedule (include/rayjoin_amd.h)
        h.set_option("pip_concurrent", 2)
    build_ms = h.last_ms(_capi.RJ_T_BUILD)

    # ---- shard the query map by chain range -------------------------------------
    from rayjoin_amd import dist as rjd
    sh = rjd.shard_of(query, args.emulate_shard, 0) if (args.emulate_shard and world == 1) else rjd.shard_of(query, world, rank)
    (e0, e1), (p0, p1) = sh["eids"], sh["points"]
    cap = int(args.xsect_factor * (n_r + n_s))  # run_query.cu:226-228
    closest = torch.empty(max(1, p1 - p0), dtype=torch.int32, device=dev)
    faces = torch.empty(max(1, p1 - p0), dtype=torch.int32, device=dev)
    xsects = torch.empty((cap, 6), dtype=torch.int64, device=dev)  # dev::Intersection<int64_t>, 48 B each
    if world > 1:
        # count + pairs leave in one all-gather on a second stream, overlapped with the PIP kernel
        ex = rjd.PairExchange(cap, dev, slot=max(4096, int(0.02 * cap)))
        pairs = ex.pairs
        max_pts = max(b - a for a, b in (rjd.shard_of(query, world, r)["points"] for r in range(world)))
    else:
        pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)

    lsi_ms, pip_ms, pts_ms = [], [], []
    state = {}

    def step(record):
        # everything is enqueued back to back; the step's single host sync is the count read-back
        h.lsi_query_async(0, 1, e0, e1, cap, pairs)
        # once the handle has settled on running the two kernels beside each other, the PIP query -- the longer
        # side, on the handle's second stream -- is issued right behind the LSI query (8-10 us earlier)
        early = h.get_option("pip_schedule") in (1, 2)
        if early:
            h.pip_query(0, 1, None, p0, p1 - p0, closest, faces, sync=False)
        h.lsi_points_async(pairs, cap, xsects)  # the records of this rank's hits (count read on the device)
        if world > 1:
            ex.begin(h)
        if not early:
            h.pip_query(0, 1, None, p0, p1 - p0, closest, faces, sync=False)
        if world > 1:  # RCCL all-gather-v of the intersection queues (rank order, zero-copy views)
            state["pairs_all"], state["cnt_all"] = ex.finish()
            n = state["cnt_all"][rank]
        else:
            n = h.lsi_query_finish(cap)
        h.sync()  # joins the PIP kernel, which runs on the handle's second stream beside the LSI kernel
        if record:
            lsi_ms.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
            pip_ms.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
            pts_ms.append(h.last_ms(_capi.RJ_T_LSI_POINTS))
        if world > 1 and args.gather_pip:  # optional: all-gather of the PIP result queues
            state["ids_all"] = rjd.allgather_point_results(closest, p1 - p0, max_pts)
        state["n"] = n

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([state["n"]], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        n_x = int(tot.item())
    else:
        n_x = state["n"]
    ms_per_step = elapsed * 1e3 / args.steps
    schedule = h.get_option("pip_schedule")  # (read now: a later index build starts the decision again)
    share = (h.get_option("lsi_share_blocks"), h.get_option("pip_share_blocks"))

    # order-independent digest of the step's results, summed over ranks (untimed): lets a test compare
    # an N-rank run with the single-GPU run of the same workload without shipping the results
    def digest():
        n = state["n"]
        pr = pairs[:n].to(torch.int64) & 0xFFFFFFFF
        xs = xsects[:n]
        pc = closest[:p1 - p0].to(torch.int64) & 0xFFFFFFFF
        hit = pc != 0xFFFFFFFF
        v = torch.stack([
            ((pr[:, 0] * 2654435761 + pr[:, 1] * 40503) % 2147483647).sum(),
            ((xs[:, 0] % 1000003) + (xs[:, 2] % 1000033)).sum(),
            hit.sum(), (pc[hit] % 2147483647).sum(), faces[:p1 - p0].to(torch.int64).sum()]).to(torch.int64)
        if world > 1:
            if dist.get_backend() == "gloo":
                c = v.cpu()
                dist.all_reduce(c)
                v = c
            else:
                dist.all_reduce(v)
        names = ("pairs", "points_xy", "pip_hits", "pip_eids", "pip_faces")
        return {k: int(x) for k, x in zip(names, v.tolist())}
    result_digest = digest()

    # phase split (synchronous calls, wall clock), one extra untimed pass
    t0 = time.perf_counter(); h.lsi_query(0, 1, e0, e1, cap, pairs); t_lsi_wall = time.perf_counter() - t0
    t0 = time.perf_counter(); h.pip_query(0, 1, None, p0, p1 - p0, closest, faces); t_pip_wall = time.perf_counter() - t0
    # each kernel with the chip to itself (synchronous queries), median of three
    alone = {"lsi": [], "pip": []}
    for _ in range(3):
        h.lsi_query(0, 1, e0, e1, cap, pairs); alone["lsi"].append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
        h.pip_query(0, 1, None, p0, p1 - p0, closest, faces); alone["pip"].append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
    lsi_alone_ms, pip_alone_ms = float(np.median(alone["lsi"])), float(np.median(alone["pip"]))

    checks = None
    if args.check and rank == 0 and world == 1:
        checks = run_checks(h, torch, dev, n_x, cap, pairs, closest, faces, query, e0, e1)

    if rank == 0:
        lsi_k = float(np.mean(lsi_ms)); pip_k = float(np.mean(pip_ms))
        # ALGORITHMIC bytes: every input element once, every output once
        n_s_loc, n_p_loc = e1 - e0, p1 - p0
        b_lsi = 32 * n_s_loc + 32 * n_r + 8 * state["n"]
        b_pip = 16 * n_p_loc + 32 * n_r + 4 * n_p_loc + 4 * n_p_loc  # + face ids
        # PMC evidence (tools/profile_run.sh + tools/pmc_summary.py): quoted only for the headline
        # workload and only while it was measured on exactly these kernel sources
        traffic, sq, prof_note = {}, {}, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        headline = world == 1 and args.scale == 1.0 and not args.emulate_shard and (args.base, args.query) == ("USCounty", "BlockGroup")
        if os.path.exists(tp) and headline:
            doc = json.load(open(tp))
            if doc.get("kernel_source_hash") == _capi.kernel_source_hash():
                traffic, sq, prof_note = doc.get("traffic", {}), doc.get("sq", {}), "profiles/%s_* (counter passes: each kernel alone on its full grid)" % doc.get("tag")
            else:
                prof_note = "profiles/traffic.json is stale (measured on other kernel sources): traffic not quoted"
        roof = {}
        for name, b, ms, kern in (("lsi", b_lsi, lsi_k, "k_lsi"), ("pip", b_pip, pip_k, "k_pip")):
            ach = b / (ms * 1e-3) / 1e9
            roof[name] = {"bound": "hbm", "kernel": kern, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(kern),
                          "algorithmic_bytes": b, "kernel_ms": round(ms, 4), "pmc_source": prof_note}
            c = sq.get(kern, {})
            if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
                # busy quad-cycles of the VALUs over the quad-cycles 1024 SIMDs have while the kernel runs
                # (GRBM_GUI_ACTIVE sums the 8 XCDs): what actually limits a kernel whose HBM traffic is
                # already the algorithmic minimum
                avail = c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
                roof[name]["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] / avail, 3)
                if c.get("SQ_ACTIVE_INST_SCA"):  # (same normalisation: scalar-issue quad-cycles per SIMD's waves)
                    roof[name]["salu_busy_frac"] = round(c["SQ_ACTIVE_INST_SCA"] / avail, 3)
                if roof[name]["valu_busy_frac"] > 0.6:
                    roof[name]["limiter"] = "valu-issue"
        if schedule in (1, 2):
            # the two kernels ran BESIDE each other in the timed steps: a kernel's duration there is not what it
            # needs alone, and the durations add up to more than the step -- say so, and add the solo figures
            for name, b, alone in (("lsi", b_lsi, lsi_alone_ms), ("pip", b_pip, pip_alone_ms)):
                roof[name]["concurrent_with"] = "k_pip" if name == "lsi" else "k_lsi"
                roof[name]["kernel_ms_alone"] = round(alone, 4)
                roof[name]["frac_alone"] = round(b / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        dom = "lsi" if lsi_k >= pip_k else "pip"
        b_step = b_lsi + b_pip + 56 * state["n"]  # + the 48-byte records and the pairs read back for them
        out = {
            "metric": "LSI+PIP query throughput, %s |><| %s" % (args.base, args.query),
            "value": round(n_s / (ms_per_step * 1e-3) / 1e6, 3), "unit": "M query segments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int128+f64", "data": "synthetic",
            "config": {"workload": "%s(base, %d segs) |><| %s(query, %d segs, %d points), -query=lsi then -query=pip, "
                                   "-mode=lbvh (software LBVH)" % (args.base, n_r, args.query, n_s, n_p),
                       "sharding": ("query map by chain range x%d, base+LBVH replicated, RCCL all-gather-v of LSI pairs%s"
                                    % (world, " and PIP eids" if args.gather_pip else "; PIP results stay with their shard"))
                                   if world > 1 else ("single GPU" if not args.emulate_shard else
                                                      "DIAGNOSTIC: rank 0's shard of a %d-way run on one GPU, value is NOT a job throughput" % args.emulate_shard),
                       "xsect_factor": args.xsect_factor, "queue_capacity": cap, "scale": args.scale,
                       # what rj_set_option("pip_concurrent", 2) settled on for this workload (rank 0)
                       "kernel_schedule": {1: "k_lsi and k_pip share the chip (%d + %d blocks)" % share, 0: "k_lsi, then k_pip",
                                           2: "k_lsi and k_pip beside each other, each on its full grid",
                                           -1: "undecided (fewer than 7 paired steps)"}[schedule]},
            "lsi_ms": round(t_lsi_wall * 1e3, 4), "pip_ms": round(t_pip_wall * 1e3, 4),
            "lsi_msegs_per_s": round(n_s_loc * world / max(t_lsi_wall, 1e-9) / 1e6, 2) if world == 1 else None,
            "pip_mpoints_per_s": round(n_p / max(t_pip_wall, 1e-9) / 1e6, 2) if world == 1 else None,
            "lsi_points_ms": round(float(np.mean(pts_ms)), 4), "result_digest": result_digest,
            "intersections": n_x, "build_index_ms": round(build_ms, 3),
            "host_ms": {"generate": round(t_gen * 1e3, 1), "upload_and_segment_build": round(t_upload * 1e3, 1)},
            "roofline": roof[dom], "roofline_other": roof["pip" if dom == "lsi" else "lsi"],
            # the whole step against the same roofline: all algorithmic bytes of the step over the step's time
            "roofline_step": {"bound": "hbm", "achieved": round(b_step / (ms_per_step * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(b_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b_step},
        }
        if checks is not None:
            out["checks"] = checks
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, ctx)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    h.close()


def run_checks(h, torch, dev, n_x, cap, pairs, closest, faces, query, e0, e1):
    """Size-independent properties at full size (no oracle can run here in seconds):
    role symmetry, shard additivity, sortedness/uniqueness, permutation invariance of PIP."""
    from rayjoin_amd import _capi
    res = {}
    h.sort_pairs(pairs, n_x)
    a = pairs[:n_x].to(torch.int64)
    key = (a[:, 0] & 0xFFFFFFFF) * (1 << 32) + (a[:, 1] & 0xFFFFFFFF)
    res["sorted_unique"] = bool((key[1:] > key[:-1]).all().item()) if n_x > 1 else True
    # role symmetry: index the query map instead; the pair SET must be identical because the
    # predicate is always evaluated as (map-0 edge, map-1 edge)
    h.build_lbvh(1)
    n_r = h.map_num_edges(0)
    p2 = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    n2 = h.lsi_query(1, 0, 0, n_r, cap, p2)
    h.sort_pairs(p2, n2)
    res["role_symmetry"] = bool(n2 == n_x and torch.equal(p2[:n2], pairs[:n_x]))
    # shard additivity
    mid = (e0 + e1) // 2
    pa = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    na = h.lsi_query(0, 1, e0, mid, cap, pa)
    pb = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    nb = h.lsi_query(0, 1, mid, e1, cap, pb)
    both = torch.cat([pa[:na], pb[:nb]])
    h.sort_pairs(both, na + nb)
    res["shard_additivity"] = bool(na + nb == n_x and torch.equal(both, pairs[:n_x]))
    # PIP permutation invariance on a 1 M-point sample
    npts = min(1 << 20, query.n_points)
    pts = torch.from_numpy(query.pts[:npts]).to(dev)
    perm = torch.randperm(npts, device=dev)
    c1 = torch.empty(npts, dtype=torch.int32, device=dev)
    c2 = torch.empty(npts, dtype=torch.int32, device=dev)
    h.pip_query(0, 1, pts, 0, npts, c1, None)
    h.pip_query(0, 1, pts[perm].contiguous(), 0, npts, c2, None)
    res["pip_permutation_invariance"] = bool(torch.equal(c1[perm], c2) and torch.equal(c1, closest[:npts]))
    return res


if __name__ == "__main__":
    main()
