#!/usr/bin/env python3
"""bench.py -- LSI + PIP query throughput of the MI355X-native path (BASELINE.json metric).

A step = one LSI Query (all query-map segments against the indexed base map: queue clear,
traversal + predicate kernel, the 48-byte Intersection record of every hit, count read-back, sync
-- what the reference times, src/run_query.cu:297-303 around src/app/lsi_lbvh.h:27-98)
and one PIP Query (every vertex of the query map, src/run_query.cu:346,441-457: k_pip_walk, then
k_pip_exact over the candidate lists it left; its first blocks locate the few points whose list overflowed).
Workload (N=1): BASELINE.json configs[1], USCounty (base, 7.1 M segments) |><| BlockGroup
(query, 28.8 M segments), as synthetic stand-ins of those sizes (the real files are
not obtainable).  Inputs are resident in HBM before the timed region.  Index build is timed
separately and reported, never part of `value`.  The default single-GPU run also times four
harder pairs (fewer steps) -- USCounty |><| NestedBlockGroup (shared vertices, 11x the
intersections), WaterBodies |><| BlockGroup (BASELINE config 5's maps), a lake-shaped base map and a
query map with the published intersection density -- so the best-case lattice is never the only number.

N>1 (`python bench.py --gpus N` starts its own N ranks; under a launcher -- torch.distributed.run
--nproc-per-node N bench.py [--gpus N] -- it is one of them, one rank per GPU): the query map is sharded by contiguous chain ranges balanced
by edge count, the base map + LBVH are replicated.  The exchanges of a step are the RCCL
all-gather-v of the intersection queues (count + pairs in one collective on a second stream,
overlapped with the PIP kernels) and the all-gather of the PIP result queues (closest eids,
4 bytes per query vertex; it runs while the NEXT step computes, double-buffered, and all of them
complete inside the timed region).  `ms_per_step_pairs_only` times the same steps without the PIP
gather.  Total work is fixed: "strong" scaling.

Prints ONE JSON line on rank 0's stdout: the headline, under 6 000 bytes (contract fields, config, flat
rooflines, cpu_baseline, one short entry per secondary pair).  Each secondary pair's compact line goes
to stderr before it, the full record (plan, checks, instruction rooflines ...) to the --detail file.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0  # measured streaming rate, same guide ("Chip-level parameters"): what `limiter` prices moved bytes against
# the harder pairs that ride in the default line: nested maps (shared vertices), the 4-level WaterBodies lattice, and a
# LAKE-SHAPED base map (2.44 M isolated rings of ~10 edges: the topology the reference's water-body / lake / park
# inputs have and no lattice has -- short rings sharing leaves, a third of the query vertices with nothing above them)
# The CPU baseline runs the WHOLE workload of every pair (round 5 ran the ring pairs at 0.4 of the lattice resolution: the oracle's
# grid PIP -- faithful to src/app/pip_grid.h:48 -- walks every cell above a point that has nothing above it, a third of the
# lattice's vertices against a lake-shaped base map: 36 s at the default grid_size 2048).  A finer grid cuts that walk (4x the
# rows above a miss, 1/16 of the edges in each) and changes no result (grid output is independent of -grid_size:
# tests/test_oracle_maps.py); the paper's own scripts run -grid_size=15000 (expr/env.sh).  The line says which size was used.
CPU_GRID = {("WaterBodiesLike", "BlockGroup"): 8192, ("LakesLike", "ParksLike"): 8192}
# ... and, since round 5, a Zipcode-sized query map that CROSSES the county boundaries at the density of the reference's own
# logs (County x Zipcode: 833 470 intersections / 23.76 M query segments = 3.5 %; the headline's independent
# lattices give 0.65 %): the stand-ins were kind on exactly this count
SECONDARY = (("USCounty", "NestedBlockGroup"), ("WaterBodies", "BlockGroup"), ("WaterBodiesLike", "BlockGroup"), ("USCounty", "CrossingZipcode"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="default: the launcher's WORLD_SIZE, or 1")
    ap.add_argument("--steps", type=int, default=5)    # the reference's -repeat=5
    ap.add_argument("--warmup", type=int, default=5)   # the reference's -warmup=5 (four pairs settle the kernel schedule)
    ap.add_argument("--base", default="USCounty")
    ap.add_argument("--query", default="BlockGroup")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink both stand-ins (debug only)")
    ap.add_argument("--xsect-factor", type=float, default=0.1, help="queue capacity factor (expr/env.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=float, default=1.0)
    ap.add_argument("--check", action="store_true", help="size-independent result checks after timing")
    ap.add_argument("--exact-stream", action="store_true", help="pipelined steps: the PIP query's exact kernel on a stream of its own (\"pip_exact_stream\"; measured: no gain, see DESIGN.md section 5 -- for A/B runs)")
    ap.add_argument("--no-gather-pip", action="store_true", help="N>1: leave the PIP result queues with their shards (round-2 behaviour)")
    ap.add_argument("--gather-pip", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--serial-kernels", action="store_true", help="run the PIP kernels after the LSI kernel instead of beside it")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (nested and WaterBodies pairs)")
    ap.add_argument("--emulate-shard", type=int, default=0, metavar="N",
                    help="diagnostic, 1 GPU: time rank 0's shard of an N-way run (no exchange); the line is marked as such")
    ap.add_argument("--settle-ms", type=float, default=40.0,
                    help="warm-up in TIME: after the W warm-up steps the same untimed steps go on until this much wall clock has passed "
                         "(the GPU's clocks are still climbing 4 ms into a run); 0 = exactly W steps")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the FULL record goes (every field of the headline and of each secondary pair, the handle's plan, "
                         "the checks, the instruction rooflines): stdout carries the compact headline only")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0, collectives over gloo")
    return ap.parse_args()


def cpu_baseline(args, ctx, base_name, query_name):
    """CPU restatement of -mode=grid (the oracle, kind 'port') timed beside the GPU numbers, on
    all host cores.  Default sample = the whole workload (0.1-0.8 s of query work on 64
    threads); --cpu-scale < 1 regenerates both stand-ins at that fraction of the lattice resolution
    (same per-segment geometry, ~scale^2 of the segments).  Query time only, like the GPU side."""
    from oracle import rjoracle as O
    from rayjoin_amd import maps, synth
    # all host cores this process may use, capped: the GPU box is shared
    O.lib().rjo_set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    cpu_scale = args.cpu_scale
    if cpu_scale != 1.0:
        g0 = synth.standin(base_name, cpu_scale * args.scale)
        g1 = synth.standin(query_name, cpu_scale * args.scale)
        ctx = maps.Context([g0, g1]).load()
    m0 = O.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = O.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
    L = O.lib()
    gsize = int(os.environ.get("RJ_BENCH_CPU_GRID", 0)) or CPU_GRID.get((base_name, query_name), 2048)  # 2048: src/flags.cc:6 default
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
                  % (base_name, query_name, cpu_scale * args.scale, m0.ne, m1.ne, n, gsize,
                     t_lsi * 1e3, t_pip * 1e3, t_build * 1e3),
        "lsi_msegs_per_s": round(m1.ne / t_lsi / 1e6, 4),
        "pip_mpoints_per_s": round(pts.shape[0] / t_pip / 1e6, 4),
    }


DEBUG_PHASES = bool(os.environ.get("RJ_BENCH_STEP_TIMES"))


def run_workload(args, env, base_name, query_name, steps, warmup, headline, with_cpu):
    """Times `steps` steps of base |><| query on this rank's shard; rank 0 returns the line (a dict)."""
    torch, dist, dev, world, rank, local_rank = env
    from rayjoin_amd import _capi, maps, synth
    from rayjoin_amd import dist as rjd

    # ---- synthetic workload (identical on every rank: seeded) -------------------------------
    t0 = time.perf_counter()
    ctx = maps.Context([synth.standin(base_name, args.scale), synth.standin(query_name, args.scale)]).load()
    base, query = ctx.maps[0], ctx.maps[1]
    t_gen = time.perf_counter() - t0
    n_r, n_s, n_p = base.n_edges, query.n_edges, query.n_points

    h = _capi.Handle(local_rank)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    for kv in filter(None, os.environ.get("RJ_BENCH_DEBUG_OPTS", "").split(",")):  # (A/B runs: "run_cap=16,strip_shift=16" -> rj_set_debug_option)
        h.set_debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    t0 = time.perf_counter()
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    t_upload = time.perf_counter() - t0
    t0 = time.perf_counter()
    h.build_lbvh(0)  # the first index of a map -- what the reference's protocol runs ("Build Index", run_query.cu): the polyline runs cut on the device, allocations, sort, leaves
    first_build_wall_ms = (time.perf_counter() - t0) * 1e3
    first_build_ms = h.last_ms(_capi.RJ_T_BUILD)   # device timer over the whole call
    build_runs_ms = h.last_ms(_capi.RJ_T_BUILD_RUNS) if h.get_option("leaf_runs0") >= 0 else 0.0
    h.build_lbvh(0)  # second build = what a rebuild costs (runs and buffers kept)
    if not args.serial_kernels:
        # LSI and PIP of a step are independent: "auto" measures taking turns / sharing the chip / full grids beside
        # each other on the first four steps and keeps the fastest schedule (include/rayjoin_amd.h)
        h.set_option("pip_concurrent", 2)
    build_ms = h.last_ms(_capi.RJ_T_BUILD)

    # ---- shard the query map by chain range -------------------------------------
    sh = rjd.shard_of(query, args.emulate_shard, 0) if (args.emulate_shard and world == 1) else rjd.shard_of(query, world, rank)
    (e0, e1), (p0, p1) = sh["eids"], sh["points"]
    cap = int(args.xsect_factor * (n_r + n_s))  # run_query.cu:226-228
    max_pts = max(1, p1 - p0)
    if world > 1:
        max_pts = max(b - a for a, b in (rjd.shard_of(query, world, r)["points"] for r in range(world)))
    gather_pip = world > 1 and not args.no_gather_pip
    # (two result buffers: a step's PIP queue is gathered while the next step fills the other one)
    # (two of every result buffer a later step would overwrite while something still reads it: a step's PIP queue is
    #  gathered while the next step fills the other one, and the pipelined loop launches step k + 1 before it has
    #  looked at step k's count)
    closest2 = [torch.empty(max_pts, dtype=torch.int32, device=dev) for _ in range(2)]
    faces = torch.empty(max_pts, dtype=torch.int32, device=dev)
    faces2 = [faces, torch.empty(max_pts, dtype=torch.int32, device=dev)]
    xsects = torch.empty((cap, 6), dtype=torch.int64, device=dev)  # dev::Intersection<int64_t>, 48 B each
    if world > 1:
        # count + pairs leave in ONE all-gather on the handle's communication stream, overlapped with the PIP kernels:
        # rj_exchange_* of the C ABI (RCCL through the handle's communicators; gloo only in the one-GPU rehearsal)
        ex = rjd.PairExchange(h, cap, dev, slot=max(4096, int(0.02 * cap)))
        pairs2 = ex.pairs
        pg = rjd.PointGather(h, max_pts, dev) if gather_pip else None
    else:
        ex = None
        pairs2 = [torch.empty((cap, 2), dtype=torch.int32, device=dev) for _ in range(2)]
        pg = None
    pairs = pairs2[0]

    lsi_ms, pip_ms, walk_ms, pts_ms = [], [], [], []
    state = {"k": 0}

    phases = []  # debug (RJ_BENCH_STEP_TIMES): host clock at the end of enqueueing / main stream done / all done / timers read

    def read_sample():
        """the stage timers of the last sampled step (its events stay as they are while the following steps record none)"""
        if not state.get("sample_pending"):
            return
        state["sample_pending"] = False
        lsi_ms.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
        pip_ms.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
        pts_ms.append(h.last_ms(_capi.RJ_T_LSI_POINTS))
        if state["two_pass"]:
            walk_ms.append(h.last_ms(_capi.RJ_T_PIP_WALK))

    def step(record, with_gather=True, sample=False):
        t_begin = time.perf_counter()
        closest = closest2[state["k"] % len(closest2)]
        state["k"] += 1
        # the stage timers (HIP events around every stage) are recorded on the sampled steps only -- every fourth timed step,
        # counted back from the last -- and READ later: before the next sampled step overwrites them, or behind the barrier
        # that closes the timed steps (reading them is ~60 us of host time that used to sit inside the sampled step)
        sample = bool(record) and sample
        if sample:
            read_sample()
        if sample != state.get("timers_on", True):
            h.set_option("timers", 1 if sample else 0)
            state["timers_on"] = sample
        # everything is enqueued back to back; the step's single host sync is the count read-back
        early = state.get("early")
        if early is None:
            # once the handle has settled on running the two sides beside each other, the PIP query -- the longer
            # side, on the handle's second stream -- is issued right behind the LSI query (8-10 us earlier); asked before
            # the launch and, once settled (it stays: rj_get_plan), not again -- the call sat between the two launches
            sched = h.get_option("pip_schedule")
            early = sched in (1, 2)
            if sched >= 0:
                state["early"] = early
        h.lsi_query_async(0, 1, e0, e1, cap, pairs)
        # (the query points: the map's own vertices by range, or -- the reference's interface, pip.h:23 -- a caller-owned array)
        qpts, qbeg = (state["caller_pts"], 0) if state.get("caller_pts") is not None else (None, p0)
        if early:
            h.pip_query(0, 1, qpts, qbeg, p1 - p0, closest, faces, sync=False)
        h.lsi_points_async(pairs, cap, xsects)  # the records of this rank's hits (count read on the device)
        if world > 1:
            ex.begin(0)
        if not early:
            h.pip_query(0, 1, qpts, qbeg, p1 - p0, closest, faces, sync=False)
        if world > 1:  # RCCL all-gather-v of the intersection queues (rank order, zero-copy views)
            state["pairs_all"], state["cnt_all"] = ex.finish(0)
            n = state["cnt_all"][rank]
        else:
            t_enq = time.perf_counter()
            n = h.lsi_query_finish(cap)
        t_main = time.perf_counter()
        h.sync()  # joins the PIP kernels, which run on the handle's second stream beside the LSI kernel
        t_all = time.perf_counter()
        if sample:
            state["sample_pending"] = True
        if DEBUG_PHASES and record and world == 1:
            phases.append((t_begin, t_enq, t_main, t_all, time.perf_counter()))
        if pg is not None and with_gather:  # all-gather of this step's PIP result queue, behind the next step's kernels
            pg.begin(closest)
        state["n"] = n
        state["closest"] = closest
        state["last_pairs"] = pairs

    def barrier():
        if pg is not None:
            state["ids_all"] = pg.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The same step WITHOUT a host sync at its end: step j is enqueued into buffer set j % 2, and only then does the
    # host look at step j - 1 -- its count (N = 1: rj_lsi_count_wait, an event) or the gathered heads of its exchange
    # (N > 1: rj_exchange_pairs_finish, the communication stream) -- so the GPU never idles while the host wakes up;
    # step j - 1's PIP queue is gathered behind step j's kernels.  Every step's results are complete and looked at
    # inside the timed region (the closing barrier waits for the last).
    def pipelined_steps(k, with_gather=True):
        # (--exact-stream: the exact kernel of step k's PIP query on its own stream, beside step k + 1's walk --
        #  include/rayjoin_amd.h "pip_exact_stream"; both of the query's outputs are double-buffered for it)
        h.set_option("pip_exact_stream", 1 if args.exact_stream else 0)
        barrier()
        h.set_option("timers", 0)
        t0 = time.perf_counter()
        n = b = 0
        for j in range(k):
            b = j % 2
            closest = closest2[b]
            early = state.get("early")
            if early is None:
                early = h.get_option("pip_schedule") in (1, 2)
            h.lsi_query_async(0, 1, e0, e1, cap, pairs2[b])
            if early:
                h.pip_query(0, 1, None, p0, p1 - p0, closest, faces2[b], sync=False)
            h.lsi_points_async(pairs2[b], cap, xsects)
            if world > 1:
                ex.begin(b)
            else:
                h.lsi_count_async(b)
            if not early:
                h.pip_query(0, 1, None, p0, p1 - p0, closest, faces2[b], sync=False)
            if pg is not None and with_gather:
                pg.begin(closest)   # (waits, on the host, for the gather of step j - 1: its buffer is step j + 1's)
            if j > 0:
                if world > 1:
                    state["pairs_all"], state["cnt_all"] = ex.finish(1 - b)
                    n = state["cnt_all"][rank]
                else:
                    n = h.lsi_count_wait(1 - b, cap)
        b = (k - 1) % 2
        if world > 1:
            state["pairs_all"], state["cnt_all"] = ex.finish(b)
            n = state["cnt_all"][rank]
        else:
            n = h.lsi_count_wait(b, cap)
        h.sync()
        barrier()
        el = time.perf_counter() - t0
        h.set_option("timers", 1)
        h.set_option("pip_exact_stream", 0)
        if b:
            faces.copy_(faces2[1])
        state["timers_on"] = True
        state["n"] = n
        state["closest"] = closest2[b]
        state["last_pairs"] = pairs2[b]
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def timed_pipelined(k, with_gather=True):
        gc.disable()
        try:
            return pipelined_steps(k, with_gather)
        finally:
            gc.enable()

    def timed(k, with_gather=True):
        # (the interpreter's cyclic collector stays out of the timed steps: a full collection that happens to fall into
        #  one -- it walks every container the three workloads' generators left behind -- was a 40 ms step among 2.8 ms ones)
        gc.disable()
        try:
            return timed_steps(k, with_gather)
        finally:
            gc.enable()
            h.set_option("timers", 1)  # (everything outside the timed steps reads its timers)
            state["timers_on"] = True

    def timed_steps(k, with_gather):
        barrier()
        t0 = time.perf_counter()
        marks = []
        for i in range(k):
            step(True, with_gather, sample=(k - 1 - i) % 4 == 0)
            marks.append(time.perf_counter())
        barrier()
        el = time.perf_counter() - t0
        read_sample()
        state["step_ms"] = [(b - a) * 1e3 for a, b in zip([t0] + marks, marks)]  # (this rank's host clock per step)
        if DEBUG_PHASES and rank == 0 and phases:
            ph = np.array(phases[-k:]) * 1e3
            tot = ph[:, 4] - ph[:, 0]
            for row in ph[tot > 2 * np.median(tot)]:  # a stalled step: where did the time go?
                print("slow step: enqueued %.3f, main stream done %.3f, all done %.3f, timers read %.3f ms after its begin"
                      % tuple(row[i] - row[0] for i in (1, 2, 3, 4)), file=sys.stderr)
            print("host phases, mean ms from step begin: enqueued %.3f, main stream done %.3f, all done %.3f, timers read %.3f"
                  % tuple((ph[:, i] - ph[:, 0]).mean() for i in (1, 2, 3, 4)), file=sys.stderr)
        if os.environ.get("RJ_BENCH_STEP_TIMES") and rank == 0:  # debug: each step's wall time, to stderr
            print("step wall ms (%s |><| %s): %s" % (base_name, query_name, " ".join("%.3f" % ((b - a) * 1e3) for a, b in zip([t0] + marks, marks))),
                  file=sys.stderr)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    state["two_pass"] = h.get_option("pip_walk") != 0
    # setup, like the second index build above: one synchronous query of each kind, so that the first PAIR -- whose
    # solo times set the split of the shared schedule -- does not also pay the first touch of every buffer
    h.lsi_query(0, 1, e0, e1, cap, pairs)
    h.pip_query(0, 1, None, p0, p1 - p0, closest2[0], faces)
    gc.collect()  # (before the warm-up steps, not between them and the timed ones: the GPU's clocks drop while the host collects)
    torch.cuda.synchronize()  # (torch's device context comes up at its first call: not inside the barrier that opens the timed steps)
    for _ in range(warmup):
        step(False)
    # Clock settle (round 6): W steps of under a millisecond are 4 ms of GPU work -- the reference's five warm-up queries are
    # 300 ms -- and the chip's clocks are still climbing when the timed steps begin (a fresh box, step by step: 0.79, 0.79,
    # 0.78 ... 0.74, 0.73 ms over the twenty).  Warm-up is meant in TIME: the same untimed steps go on until `--settle-ms` of
    # wall clock have passed since the warm-up began (default 40 ms; 0 = exactly W steps); the line says how many were added.
    def settle():
        """untimed steps until --settle-ms of wall clock have passed (N = 1 only: a step of an N > 1 job holds collectives, every
        rank must run the same number of steps); -> how many"""
        extra = 0
        if warmup > 0 and args.settle_ms > 0 and world == 1:
            t_settle = time.perf_counter() + args.settle_ms * 1e-3
            while time.perf_counter() < t_settle and extra < 2000:
                step(False)
                extra += 1
        return extra
    state["extra_warmup"] = settle()
    # the kernel schedule must be settled before anything is timed: four measured pairs do it, i.e. the fifth
    # warm-up step already runs the chosen schedule; with fewer warm-ups (or --serial-kernels) the line says so
    # (a re-ordered -- spatially shuffled -- query set is never paired: nothing to settle, one kernel after the other)
    never_paired = not args.serial_kernels and h.get_option("pip_schedule_trials") == 0 and h.get_option("pip_schedule") < 0
    settled = args.serial_kernels or never_paired or h.get_option("pip_schedule") >= 0
    elapsed = timed(steps)
    ms_per_step = elapsed * 1e3 / steps
    ms_synced = ms_pipelined = None
    if settled and steps >= 2:
        el_p = timed_pipelined(steps)
        ms_pipelined = el_p * 1e3 / steps
        if world > 1:
            # N > 1: the steps of the job are the pipelined ones (no host sync inside the timed region but the closing one);
            # the synchronised form stays in the line beside it
            ms_synced, ms_per_step, elapsed = ms_per_step, ms_pipelined, el_p
    # (the line's value is the K steps over their total time, as the contract says; the median and the slowest step say
    #  whether one stall -- another tenant of the host, a driver hiccup: seen as one 11-43 ms step among 2.8 ms ones --
    #  is in that total)
    step_median_ms, step_max_ms = float(np.median(state["step_ms"])), float(np.max(state["step_ms"]))
    # ("auto" may have dropped the first PIP pass for this workload: then k_pip is the PIP query)
    state["two_pass"] = state["two_pass"] and h.get_option("pip_last_passes") == 3
    lsi_k = float(np.mean(lsi_ms)); pip_k = float(np.mean(pip_ms)); walk_k = float(np.mean(walk_ms)) if (walk_ms and state["two_pass"]) else None
    ms_pairs_only = None
    if gather_pip:
        ms_pairs_only = (timed_pipelined if ms_synced is not None else timed)(steps, with_gather=False) * 1e3 / steps
    if world > 1:
        tot = torch.tensor([state["n"]], dtype=torch.int64, device=dev)
        if dist.get_backend() == "gloo":
            c = tot.cpu(); dist.all_reduce(c); tot = c
        else:
            dist.all_reduce(tot)
        n_x = int(tot.item())
    else:
        n_x = state["n"]
    state["plan"] = h.get_plan()  # (what the timed steps ran, in the handle's own words; read now: later queries overwrite it)
    schedule = h.get_option("pip_schedule")  # (read now: a later index build starts the decision again)
    state["walk_points"] = h.get_option("pip_last_walk_points")  # (which kernels the timed steps ran: one or two queries per lane)
    state["lsi_segments"] = h.get_option("lsi_last_segments")
    state["columns"] = h.get_option("pip_last_columns")
    share = (h.get_option("lsi_share_blocks"), h.get_option("pip_share_blocks"))
    pip_rest = h.get_option("pip_rest_aux" if schedule in (1, 2) else "pip_rest") if h.get_option("pip_walk") else None
    closest = state["closest"]

    # order-independent digest of the step's results, summed over ranks (untimed): lets a test compare
    # an N-rank run with the single-GPU run of the same workload without shipping the results
    def digest():
        n = state["n"]
        pr = state.get("last_pairs", pairs)[:n].to(torch.int64) & 0xFFFFFFFF
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
    gathered_ok = None
    if gather_pip and state.get("ids_all") is not None:
        # every rank now holds every shard's queue: this rank's slice of the gathered buffer is its own result
        gathered_ok = bool(torch.equal(state["ids_all"][rank, :p1 - p0], closest[:p1 - p0]))

    # The same steps with the PIP query handed a CALLER-OWNED point array (the reference's PIP::Query(Stream&, int,
    # ArrayView<point_t>), src/app/pip.h:23; src/run_query.cu:346,441-443 passes a separate device vector): a copy of
    # this rank's vertices.  After its first query the handle enqueues such a query without a host round trip and pairs
    # it with the LSI query in flight like a map-owned range; the results must be the map-owned ones.
    caller = None
    if world == 1 and p1 > p0:
        own = closest[:p1 - p0].clone()
        cpts = torch.from_numpy(np.ascontiguousarray(query.pts[p0:p1])).to(dev)
        state["caller_pts"] = cpts
        for _ in range(3):   # (first sight: one estimate with a round trip; then the settled path)
            step(False)
        settle()             # (the same warm-up in time as the steps it is compared with: the digest above left the GPU idle)
        el_c = timed(steps)
        same = bool(torch.equal(closest[:p1 - p0], own))
        h.pip_query(0, 1, cpts, 0, p1 - p0, closest, faces)
        pip_caller_ms = h.last_ms(_capi.RJ_T_PIP_KERNEL)
        caller = {"ms_per_step": round(el_c * 1e3 / steps, 4), "ms_per_step_median": round(float(np.median(state["step_ms"])), 4),
                  "vs_map_owned": round(el_c / elapsed, 4), "pip_query_ms_alone": round(pip_caller_ms, 4),
                  "re_ordered": bool(h.get_option("query_last_ordered")), "equals_map_owned_results": same,
                  "pip_schedule": h.get_option("pip_schedule")}
        state["caller_pts"] = None
        del cpts
        step(False)  # (back on the map-owned path for what follows)
        h.set_option("timers", 1)  # (an unsampled step switches the stage timers off: what follows reads them)
        state["timers_on"] = True

    # phase split (synchronous calls, wall clock), one extra untimed pass
    t0 = time.perf_counter(); h.lsi_query(0, 1, e0, e1, cap, pairs); t_lsi_wall = time.perf_counter() - t0
    t0 = time.perf_counter(); h.pip_query(0, 1, None, p0, p1 - p0, closest, faces); t_pip_wall = time.perf_counter() - t0
    # each kernel with the chip to itself (synchronous queries), median of three
    alone = {"lsi": [], "pip": [], "walk": []}
    for _ in range(3):
        h.lsi_query(0, 1, e0, e1, cap, pairs); alone["lsi"].append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
        h.pip_query(0, 1, None, p0, p1 - p0, closest, faces); alone["pip"].append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
        if state["two_pass"]:
            alone["walk"].append(h.last_ms(_capi.RJ_T_PIP_WALK))
    lsi_alone_ms, pip_alone_ms = float(np.median(alone["lsi"])), float(np.median(alone["pip"]))
    walk_alone_ms = float(np.median(alone["walk"])) if alone["walk"] else None

    checks = None
    if args.check and rank == 0 and world == 1 and headline:
        checks = run_checks(h, torch, dev, n_x, cap, state.get("last_pairs", pairs), closest, faces, query, e0, e1)

    out = None
    if rank == 0:
        # ALGORITHMIC bytes: every input element once, every output once
        n_s_loc, n_p_loc = e1 - e0, p1 - p0
        b_lsi = 32 * n_s_loc + 32 * n_r + 8 * state["n"]
        b_pip = 16 * n_p_loc + 32 * n_r + 4 * n_p_loc + 4 * n_p_loc  # + face ids
        # PMC evidence (tools/profile_run.sh + tools/pmc_summary.py): quoted only for the headline
        # workload and only while it was measured on exactly these kernel sources
        traffic, sq, prof_note, fetch_cal = {}, {}, None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        is_headline = headline and world == 1 and args.scale == 1.0 and not args.emulate_shard and (base_name, query_name) == ("USCounty", "BlockGroup")
        shared_sq = {}
        plain = world == 1 and args.scale == 1.0 and not args.emulate_shard
        if os.path.exists(tp) and plain:
            doc = json.load(open(tp))
            sec = doc.get("sections", {}).get("%s_%s" % (base_name, query_name))
            if doc.get("kernel_source_hash") != _capi.kernel_source_hash():
                if is_headline or sec:
                    prof_note = "profiles/traffic.json is stale (measured on other kernel sources): traffic not quoted"
            elif is_headline:
                traffic, sq, prof_note = doc.get("traffic", {}), doc.get("sq", {}), "profiles/%s_* (counter passes: each kernel alone on its full grid)" % doc.get("tag")
                fetch_cal = doc.get("fetch_calibration")
                # (the same kernels on the grids of the shared schedule -- k_lsi2 on 512 blocks, the walk on 1 536 -- profiled one
                #  after the other: rocprofv3 serialises the kernels of a counter pass, tools/regime_probe.py)
                shared_sq = doc.get("sections", {}).get("shared", {}).get("sq", {})
            elif sec:  # (another pair with a section of its own: the ring-shaped ones)
                traffic, sq, prof_note = sec.get("traffic", {}), sec.get("sq", {}), "profiles/%s_pmc_sections.csv, section %s_%s (each kernel alone on its full grid)" % (doc.get("tag"), base_name, query_name)
                fetch_cal = doc.get("fetch_calibration")
        # (a base map of isolated rings has a column index: the first pass reads the point's strip instead of walking the tree)
        walk_name = "k_pip_strip" if state.get("columns") else ("k_pip_walk2" if state["walk_points"] == 2 else "k_pip_walk")
        pip_kernel = walk_name if state["two_pass"] else "k_pip"
        # (the kernel the handle's plan names: k_lsi2 / k_lsi, or -- a tree without a second order of its steep blocks: maps of closed
        #  rings -- k_lsi2x / k_lsix, the same bodies with nothing of that order left in them)
        lsi_kernel = ((state.get("plan") or {}).get("lsi") or {}).get("kernel") or ("k_lsi2" if state["lsi_segments"] == 2 else "k_lsi")
        # the PIP query's dominant kernel: its own HIP-event time in the timed steps (the three PIP kernels together: query_ms)
        pip_dom_ms = walk_k if walk_k else pip_k
        roof = {}
        for name, b, ms, kern in (("lsi", b_lsi, lsi_k, lsi_kernel), ("pip", b_pip, pip_dom_ms, pip_kernel)):
            ach = b / (ms * 1e-3) / 1e9
            # `frac` prices the ALGORITHMIC bytes (every input once, every output once) against HBM peak: an
            # effective-throughput figure.  `traffic` = bytes the kernel actually moved (PMC), `traffic_frac` =
            # that over the same time and peak: the real bandwidth fraction.
            r = {"bound": "hbm", "kernel": kern, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(kern),
                 "algorithmic_bytes": b, "kernel_ms": round(ms, 4), "pmc_source": prof_note}
            c = sq.get(kern, {})
            alone_ms = lsi_alone_ms if name == "lsi" else (walk_alone_ms or pip_alone_ms)
            if traffic.get(kern) and fetch_cal and kern == "k_pip_strip":
                r["fetch_calibration"] = fetch_cal   # (scattered reads: what the doubled counter rests on)
            if traffic.get(kern):
                # (the counter passes run each kernel alone on its full grid: price the bytes against THAT time)
                r["traffic_frac"] = round(traffic[kern] / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                # ... and against the time it takes IN the timed steps (beside the other side, on its share of the chip)
                r["frac_moved_bytes"] = round(traffic[kern] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                if traffic[kern] < b:
                    r["traffic_note"] = ("moves less than the algorithmic bytes: %s" %
                                         ("a pre-filter built at upload (4-byte cell codes + the base map's occupancy bitmap) skips "
                                          "the segments of groups with nothing near them" if name == "lsi" else
                                          "it reads 16-byte boxes and an 8-byte bucket-table entry per base segment, not the 32-byte segment"))
            if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
                # busy quad-cycles of the VALUs over the quad-cycles 1024 SIMDs have while the kernel runs (GRBM_GUI_ACTIVE
                # sums the 8 XCDs; a SIMD issues one VALU instruction per 4 cycles: tools/issue_probe.hip): what
                # actually limits a kernel whose HBM traffic is nowhere near the roofline
                avail = c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
                r["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] / avail, 3)
                if c.get("SQ_ACTIVE_INST_SCA"):  # (same normalisation: scalar-issue quad-cycles per SIMD's waves)
                    r["salu_busy_frac"] = round(c["SQ_ACTIVE_INST_SCA"] / avail, 3)
                if c.get("SQ_INSTS_VALU"):
                    units = n_p_loc if name == "pip" else n_s_loc
                    r["valu_per_query"] = round(c["SQ_INSTS_VALU"] / units, 2)
                    r["salu_per_query"] = round(c.get("SQ_INSTS_SALU", 0) / units, 2)
                # the measured limiter = the LARGEST of: the bytes moved against the HBM rate a kernel can actually reach
                # (6.29 TB/s streaming, MI355X_MICROARCH.md), the fraction of the VALU issue slots that are busy, and the
                # fraction of its wave-cycles the kernel spends waiting (dependent-load latency) -- not a threshold on one of them
                cands = {"valu-issue": r["valu_busy_frac"]}
                if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES"):
                    r["wait_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
                    cands["dependent-load latency"] = r["wait_frac"]
                if traffic.get(kern):
                    r["traffic_frac_of_achievable"] = round(traffic[kern] / (alone_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 5)
                    cands["hbm-traffic"] = r["traffic_frac_of_achievable"]
                r["limiter"] = max(cands, key=cands.get)
                r["limiter_frac"] = round(min(1.0, cands[r["limiter"]]), 3)
                if c.get("SQ_INSTS_VALU"):
                    # The roofline these kernels actually sit under: a SIMD issues one VALU wave-instruction per 4 cycles whatever
                    # the occupancy (tools/issue_probe.hip measured 4.2), so the chip's ceiling is SIMDs x clock / 4.
                    # `floor_*`: what a group MUST issue on this workload -- one load, test and ballot per expanded node, one
                    # table lookup + two corrections per leaf visit, four cross-lane reads and a test per scan step, loading,
                    # bounding and handing over the group -- with the visit counts of the instrumented kernel.
                    clk = getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2400000) * 1e3
                    simds = 4 * getattr(torch.cuda.get_device_properties(dev), "multi_processor_count", 256)
                    peak = simds * clk / 4.0
                    ach = c["SQ_INSTS_VALU"] / (alone_ms * 1e-3)
                    r["instruction_roofline"] = {"bound": "valu-issue", "achieved": round(ach / 1e9, 1), "peak": round(peak / 1e9, 1),
                                                 "unit": "G wave-instructions/s", "frac": round(ach / peak, 4),
                                                 "measured_on": "the kernel alone on its full grid (PMC pass)"}
                    if kern.startswith("k_pip_walk") and is_headline:
                        # (per 128-position group of k_pip_walk2, instrumented: profiles/r05_walk_stats.txt -- 10.4 pops (1.3 stale), 3.7 node
                        #  expansions, 5.4 leaf blocks = 7.5 (leaf, set) visits with 20.5 of 64 lanes wanting, 8.2 scan steps, 7.2 candidate
                        #  bodies, 1.1 sweeps; priced with the instruction counts of the kernel's ISA)
                        r["instruction_roofline"]["floor_valu_per_query"] = 3.8
                        r["instruction_roofline"]["floor_note"] = ("per 128-position group: 7.5 (leaf, set) visits x (22 lookup + 9 test) + 7.2 candidate bodies x 17 = 2.8 per point "
                                                                   "is the work itself at a third of the lanes; 10.4 pops x 9, 3.7 expansions x 25, the group's bound, "
                                                                   "sweeps, load / hand-over / scheduler are the other %.1f (profiles/r05_walk_stats.txt)" % (r["valu_per_query"] - 2.8))
            cs = shared_sq.get(kern, {})
            if cs.get("SQ_INSTS_VALU") and cs.get("GRBM_GUI_ACTIVE"):
                units = n_p_loc if name == "pip" else n_s_loc
                avail = cs["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
                r["on_its_share_of_the_chip"] = {
                    "note": "the kernel ALONE on the grid it has in the shared schedule (counter passes serialise kernels: the other side's instructions are not in these numbers)",
                    "valu_per_query": round(cs["SQ_INSTS_VALU"] / units, 2), "salu_per_query": round(cs.get("SQ_INSTS_SALU", 0) / units, 2),
                    "valu_busy_frac": round(cs.get("SQ_ACTIVE_INST_VALU", 0) / avail, 3),
                    "wait_frac_of_wave_cycles": round(cs.get("SQ_WAIT_ANY", 0) / cs["SQ_WAVE_CYCLES"], 3) if cs.get("SQ_WAVE_CYCLES") else None}
            roof[name] = r
        roof["pip"]["query_ms"] = round(pip_k, 4)  # all PIP kernels of a step, first launch to last end
        roof["pip"]["frac_query"] = round(b_pip / (pip_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if schedule in (1, 2):
            # the two sides ran BESIDE each other in the timed steps: a kernel's duration there is not what it
            # needs alone, and the durations add up to more than the step -- say so, and add the solo figures
            for name, b, alone_ms in (("lsi", b_lsi, lsi_alone_ms), ("pip", b_pip, walk_alone_ms or pip_alone_ms)):
                roof[name]["concurrent_with"] = pip_kernel if name == "lsi" else lsi_kernel
                roof[name]["kernel_ms_alone"] = round(alone_ms, 4)
                roof[name]["frac_alone"] = round(b / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            roof["pip"]["query_ms_alone"] = round(pip_alone_ms, 4)
        dom = "lsi" if lsi_k >= pip_k else "pip"
        b_step = b_lsi + b_pip + 56 * state["n"]  # + the 48-byte records and the pairs read back for them
        sharding = "single GPU"
        if world > 1:
            sharding = ("query map by chain range x%d, base+LBVH replicated, RCCL all-gather-v of LSI pairs (overlapped with the PIP kernels)%s"
                        % (world, " and all-gather of the PIP result queues (%d B per rank per step, behind the next step's kernels, "
                                  "double-buffered; all complete inside the timed region)" % (4 * max_pts) if gather_pip
                           else "; PIP results stay with their shard (--no-gather-pip)"))
        elif args.emulate_shard:
            sharding = "DIAGNOSTIC: rank 0's shard of a %d-way run on one GPU, value is NOT a job throughput" % args.emulate_shard
        out = {
            "metric": "LSI+PIP query throughput, %s |><| %s" % (base_name, query_name),
            "value": round(n_s / (ms_per_step * 1e-3) / 1e6, 3), "unit": "M query segments/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "warmup_extra_steps": state.get("extra_warmup", 0),
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int128+f64", "data": "synthetic",
            "config": {"workload": "%s(base, %d segs) |><| %s(query, %d segs, %d points), -query=lsi then -query=pip, "
                                   "-mode=lbvh (software LBVH)" % (base_name, n_r, query_name, n_s, n_p),
                       "sharding": sharding,
                       "xsect_factor": args.xsect_factor, "queue_capacity": cap, "scale": args.scale,
                       # what rj_set_option("pip_concurrent", 2) settled on for this workload (rank 0), and whether it had
                       # done so before the first timed step (no schedule trials inside the timed region)
                       "kernel_schedule": {1: "k_lsi and the PIP kernels share the chip (%d + %d blocks)" % share, 0: "k_lsi, then the PIP kernels",
                                           2: "k_lsi and the PIP kernels beside each other, each on its full grid",
                                           -1: "k_lsi, then the PIP kernels (re-ordered query sets are never paired)" if never_paired
                                               else "undecided (fewer than 5 paired steps)"}[schedule],
                       "schedule_settled_before_timing": bool(settled),
                       "pip_passes": ("%s + k_pip_exact (its first blocks locate the %s points whose candidate list overflowed)" % (walk_name, pip_rest)) if state["two_pass"]
                                     else "k_pip alone" + (" (the walk left %s lists to it: auto dropped the first pass)" % pip_rest if h.get_option("pip_walk") else "")},
            "lsi_ms": round(t_lsi_wall * 1e3, 4), "pip_ms": round(t_pip_wall * 1e3, 4),
            "lsi_msegs_per_s": round(n_s_loc * world / max(t_lsi_wall, 1e-9) / 1e6, 2) if world == 1 else None,
            "pip_mpoints_per_s": round(n_p / max(t_pip_wall, 1e-9) / 1e6, 2) if world == 1 else None,
            "lsi_points_ms": round(float(np.mean(pts_ms)), 4), "result_digest": result_digest,
            "intersections": n_x, "intersections_per_query_segment": round(n_x / max(1, n_s), 5), "build_index_ms": round(first_build_ms, 3), "build_index_wall_ms": round(first_build_wall_ms, 3),
            "build_index_runs_ms": round(build_runs_ms, 3), "rebuild_index_ms": round(build_ms, 3),
            "pip_caller_array": caller,
            "index_leaves": "polyline runs" if h.get_option("leaf_order_used0") == 1 else "Hilbert neighbours",
            "index_slots_per_segment": round(h.get_option("leaf_slots0") / max(1, n_r), 3),
            "index_extras": {"skyline": bool(h.get_option("skyline_used0")), "pip_columns": bool(h.get_option("pip_columns_used0")),
                             "closed_chains": h.get_option("closed_chains0")},
            "host_ms": {"generate": round(t_gen * 1e3, 1), "upload_and_segment_build": round(t_upload * 1e3, 1)},
            "roofline": roof[dom], "roofline_other": roof["pip" if dom == "lsi" else "lsi"],
            # the whole step against the same roofline: all algorithmic bytes of the step over the step's time
            "roofline_step": {"bound": "hbm", "achieved": round(b_step / (ms_per_step * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(b_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b_step},
        }
        if ms_pipelined is not None:
            out["ms_per_step_pipelined"] = round(ms_pipelined, 4)  # step j + 1 launched before step j's count / heads are read
        if ms_synced is not None:
            out["ms_per_step_synced"] = round(ms_synced, 4)        # every step ending in the host's read of the gathered heads
        out["pipelined"] = ms_synced is not None
        # the ranks that ran: the launcher's count, and -- on the native transport -- what RCCL counts in the handle's communicator
        out["plan"] = state.get("plan")  # rj_get_plan right after the timed steps: kernels, grids, schedule, and why
        out["ranks"] = {"world_size": world, "rccl_comm_ranks": h.get_option("comm_ranks") or None,
                        "transport": ("gloo (one-GPU rehearsal: every rank on cuda:0)" if args.rehearse_one_gpu else "rccl") if world > 1 else None}
        if world > 1 and not args.rehearse_one_gpu and out["ranks"]["rccl_comm_ranks"] != world:
            raise SystemExit("bench.py: RCCL counts %s ranks in the communicator, the launcher %d" % (out["ranks"]["rccl_comm_ranks"], world))
        if world > 1:
            out["multi_gpu_note"] = ("no hardware curve exists for N > 1 in this repository's records: this line is whatever node ran it; "
                                     "single-GPU emulations of a shard (--emulate-shard) are diagnostics, not scaling results. "
                                     "The steps of an N > 1 line are the pipelined ones (ms_per_step_synced beside them): the like-for-like figure "
                                     "of an N = 1 line is its ms_per_step_pipelined, not its ms_per_step")
        out["ms_per_step_median"] = round(step_median_ms, 4)
        out["ms_slowest_step"] = round(step_max_ms, 4)
        if ms_pairs_only is not None:
            out["ms_per_step_pairs_only"] = round(ms_pairs_only, 4)
            out["pip_gather_verified"] = gathered_ok
        if checks is not None:
            out["checks"] = checks
        if with_cpu:
            out["cpu_baseline"] = cpu_baseline(args, ctx, base_name, query_name)
    h.close()
    return out


# ---- what is printed ---------------------------------------------------------------------------------------------------------
# stdout carries ONE JSON line, the headline, small enough to survive any tail of the output (round 5's single line had grown
# to 20.9 KB and the driver's record lost its head): the contract fields, `config`, flat `roofline` / `roofline_other` /
# `roofline_step`, `cpu_baseline`, and one short entry per secondary pair.  Each secondary pair's own compact line goes to
# stderr BEFORE it (the reference prints its timings there too, src/util/timer.h:57-80), and the FULL record of everything --
# plan, checks, instruction rooflines, the kernels on their share of the chip -- to the file named in `detail`.
HEADLINE_MAX_BYTES = 6000
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel_ms", "traffic_frac",
             "traffic_frac_of_achievable", "frac_moved_bytes", "fetch_calibration", "limiter", "limiter_frac", "valu_busy_frac", "wait_frac",
             "valu_per_query", "salu_per_query", "concurrent_with", "kernel_ms_alone", "frac_alone", "query_ms", "pmc_source")
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "warmup_extra_steps", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
             "data", "config", "ms_per_step_pipelined", "ms_per_step_synced", "pipelined", "ms_per_step_median", "ms_slowest_step",
             "ms_per_step_pairs_only", "pip_gather_verified", "intersections", "intersections_per_query_segment", "lsi_points_ms",
             "build_index_ms", "rebuild_index_ms", "index_slots_per_segment", "result_digest", "ranks", "multi_gpu_note")
SECONDARY_KEYS = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_pipelined", "intersections",
                  "build_index_ms", "index_slots_per_segment")


def flat_roofline(r, drop=()):
    return {k: r[k] for k in ROOF_KEYS if k in r and k not in drop and not isinstance(r[k], dict)}


def compact(line, secondary=False):
    """the printed form of a full record: nothing nested deeper than `config` / a flat roofline / `cpu_baseline`"""
    cfg = line["config"]
    if secondary:
        o = {k: line[k] for k in SECONDARY_KEYS if line.get(k) is not None}
        o["line"] = "secondary"
        o["kernel_schedule"] = cfg.get("kernel_schedule")
        o["pip_passes"] = cfg.get("pip_passes")
    else:
        o = {k: line[k] for k in HEAD_KEYS if line.get(k) is not None or k == "vs_baseline"}
        o["config"] = {k: cfg[k] for k in ("workload", "sharding", "kernel_schedule", "schedule_settled_before_timing", "pip_passes") if k in cfg}
    for k in ("roofline", "roofline_other", "roofline_step"):
        if k in line:
            o[k] = flat_roofline(line[k], drop=("pmc_source",) if secondary else ())
    if line.get("pip_caller_array"):
        o["pip_caller_array_vs_map_owned"] = line["pip_caller_array"]["vs_map_owned"]
    if line.get("checks") is not None:
        o["checks_all_true"] = all(line["checks"].values())
    if line.get("plan"):
        o["plan_settled"] = bool(line["plan"].get("schedule", {}).get("settled"))
    if "cpu_baseline" in line:
        o["cpu_baseline"] = dict(line["cpu_baseline"])
    return o


def headline(out, sec, written):
    """the ONE stdout line: the headline's compact form + one short entry per secondary pair + where the full record is"""
    head = compact(out)
    if sec:
        head["secondary"] = {s["metric"].split(", ", 1)[1]: {
            "ms_per_step": s["ms_per_step"], "ms_per_step_pipelined": s.get("ms_per_step_pipelined"), "kernel": s["roofline"]["kernel"],
            "frac": s["roofline"]["frac"], "kernel_ms": s["roofline"]["kernel_ms"], "traffic": s["roofline"].get("traffic"),
            "limiter": s["roofline"].get("limiter"), "cpu_baseline": (s.get("cpu_baseline") or {}).get("value")} for s in sec}
    head["detail"] = written
    text = json.dumps(head)
    for k in ("multi_gpu_note", "result_digest", "secondary"):  # (never again a line whose head a tail cuts off: shed the longest first)
        if len(text) <= HEADLINE_MAX_BYTES:
            break
        head.pop(k, None)
        text = json.dumps(head)
    return text


def emit(out, sec, detail_path):
    """rank 0: the full record to `detail_path`, each secondary pair's compact line to stderr, the headline to stdout -- last"""
    written = None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
        with open(detail_path, "w") as f:
            json.dump({"headline": out, "secondary": sec}, f)
        written = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
    except OSError as e:   # (a read-only tree: the headline still goes out)
        print("bench.py: could not write %s: %s" % (detail_path, e), file=sys.stderr)
    for s in sec:
        print(json.dumps(compact(s, secondary=True)), file=sys.stderr)
    sys.stderr.flush()
    print(headline(out, sec, written))
    sys.stdout.flush()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- torch.distributed.run as
    a CHILD process, before this process has imported torch or touched the GPU (no exec from a process that holds a GPU
    context) -- relay its output (rank 0 prints the one JSON line) and leave with its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus is None:  # (under a launcher without --gpus: its rank count; an EXPLICIT --gpus must agree with it, below)
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a line that says n_gpus = WORLD_SIZE under a command that says --gpus N would be a lie either way: refuse
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; start it as `python bench.py --gpus N` "
                 "(it launches its own ranks) or with a matching --nproc-per-node" % (args.gpus, os.environ["WORLD_SIZE"]))
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
    env = (torch, dist, dev, world, rank, local_rank)

    with_cpu = world == 1 and not args.no_cpu_baseline
    out = run_workload(args, env, args.base, args.query, args.steps, args.warmup, True, with_cpu)
    # the harder pairs, in the same line: only on the default single-GPU run (each adds a few seconds)
    default_run = world == 1 and not args.emulate_shard and args.scale == 1.0 and (args.base, args.query) == ("USCounty", "BlockGroup")
    sec = []
    if default_run and not args.no_secondary:
        for b, q in SECONDARY:
            torch.cuda.empty_cache()
            sec.append(run_workload(args, env, b, q, max(5, min(args.steps, 10)), max(5, args.warmup), False, with_cpu))
    if rank == 0:
        emit(out, sec, args.detail)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_checks(h, torch, dev, n_x, cap, pairs, closest, faces, query, e0, e1):
    """Size-independent properties at full size (no oracle can run here in seconds):
    role symmetry, shard additivity, sortedness/uniqueness, the two forms of the records kernel against each other,
    permutation invariance of PIP, and the two-pass PIP against k_pip alone."""
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
    # the 48-byte records: the gcd-free kernel + the gcd kernel over what it declines against the gcd kernel over every
    # pair (two implementations of the narrowing store, bit for bit on every record of the step)
    if n_x:
        r1 = torch.empty((n_x, 6), dtype=torch.int64, device=dev)
        r2 = torch.empty((n_x, 6), dtype=torch.int64, device=dev)
        keep = h.get_option("lsi_points_split")
        h.set_option("lsi_points_split", 1); h.lsi_points(pairs, n_x, r1)
        h.set_option("lsi_points_split", 0); h.lsi_points(pairs, n_x, r2)
        h.set_option("lsi_points_split", keep)
        res["records_two_kernels_equal_one"] = bool(torch.equal(r1, r2))
    # PIP permutation invariance on a 1 M-point sample
    npts = min(1 << 20, query.n_points)
    pts = torch.from_numpy(query.pts[:npts]).to(dev)
    perm = torch.randperm(npts, device=dev)
    c1 = torch.empty(npts, dtype=torch.int32, device=dev)
    c2 = torch.empty(npts, dtype=torch.int32, device=dev)
    h.pip_query(0, 1, pts, 0, npts, c1, None)
    h.pip_query(0, 1, pts[perm].contiguous(), 0, npts, c2, None)
    res["pip_permutation_invariance"] = bool(torch.equal(c1[perm], c2) and torch.equal(c1, closest[:npts]))
    # the three PIP passes against k_pip alone, every vertex of the query map: eids and face ids
    if h.get_option("pip_walk") != 0:
        n_p = query.n_points
        c3 = torch.empty(n_p, dtype=torch.int32, device=dev)
        f3 = torch.empty(n_p, dtype=torch.int32, device=dev)
        h.set_option("pip_walk", 0)
        h.pip_query(0, 1, None, 0, n_p, c3, f3)
        h.set_option("pip_walk", 1)
        res["pip_two_pass_equals_single_kernel"] = bool(torch.equal(c3, closest[:n_p]) and torch.equal(f3, faces[:n_p]))
    return res


if __name__ == "__main__":
    main()
