"""Multi-GPU sharding of the query map and gathering of result queues.

The reference is single-GPU (no NCCL/MPI anywhere); this is new design.  Each query segment /
point is independent against a read-only base map + LBVH, so the query map is split into
contiguous CHAIN ranges balanced by edge count (a chain range is a contiguous eid range and a
contiguous point range, because eid = p_idx - ichain, src/map/map.h:200-203), the base map and
its LBVH are replicated, and the only exchange step is an all-gather(v) of the result queues:
one all-gather of per-rank counts + one padded all-gather of the used prefix.  On GPUs the
backend is "nccl" (= RCCL over xGMI); the same code runs on "gloo" for CPU tests.
"""
import torch
import torch.distributed as dist


def _needs_host_staging(t):
    """gloo has no GPU transport: stage device tensors through the host (CPU rehearsal of the
    N>1 path on a box with fewer GPUs than ranks).  With nccl (= RCCL) tensors stay on the GPU."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather_flat(out, inp):
    if _needs_host_staging(inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp)


def shard_of(query_map, world, rank):
    """-> dict(chains=(c0,c1), eids=(e0,e1), points=(p0,p1)) of this rank's shard."""
    c0, c1 = query_map.shard_chain_ranges(world)[rank]
    e0, e1 = query_map.chain_range_to_eids(c0, c1)
    return dict(chains=(c0, c1), eids=(e0, e1),
                points=(int(query_map.row_index[c0]), int(query_map.row_index[c1])))


def allgather_counts(n, device):
    world = dist.get_world_size()
    me = torch.tensor([int(n)], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    _all_gather_flat(out, me)
    return out


def allgather_pairs(pairs, n, scratch=None):
    """All-gather-v of (eid0, eid1) pair queues.  pairs: int32[cap,2] tensor whose first n rows
    are valid on this rank.  Returns (gathered int32[total,2] in rank order, counts int64[world]).
    `scratch`: optional preallocated flat int32 buffer of >= world*max_count*2 elements."""
    world = dist.get_world_size()
    counts = allgather_counts(n, pairs.device)
    cl = counts.tolist()
    gmax = max(cl) if cl else 0
    if gmax == 0:
        return pairs[:0], counts
    if pairs.shape[0] < gmax:  # only possible if some rank overflowed its queue
        raise ValueError("local queue smaller than the largest shard result")
    if scratch is not None and scratch.numel() >= world * gmax * 2:
        flat = scratch[:world * gmax * 2]
    else:
        flat = torch.empty(world * gmax * 2, dtype=pairs.dtype, device=pairs.device)
    _all_gather_flat(flat, pairs[:gmax].reshape(-1))  # flat in, flat out: nccl and gloo
    recv = flat.view(world, gmax, 2)
    out = torch.cat([recv[r, :cl[r]] for r in range(world)], dim=0)
    return out, counts


def allgather_point_results(ids, n, max_n):
    """All-gather of per-point results (closest eids / face ids) of contiguous point shards;
    shards are contiguous point ranges in rank order, so concatenation restores point order.
    ids: int32[>=n] on this rank, max_n: the largest shard size (same on every rank)."""
    world = dist.get_world_size()
    pad = torch.zeros(max_n, dtype=ids.dtype, device=ids.device)
    pad[:n] = ids[:n]
    flat = torch.empty(world * max_n, dtype=ids.dtype, device=ids.device)
    _all_gather_flat(flat, pad)
    recv = flat.view(world, max_n)
    counts = allgather_counts(n, ids.device).tolist()
    return torch.cat([recv[r, :counts[r]] for r in range(world)], dim=0)


class PointGather:
    """All-gather of the PIP result queues (closest eids) of contiguous point shards, off the critical path.

    A step's PIP kernel is the last thing the step runs, so its gather has nothing of the SAME step to hide
    behind; it runs on its own stream while the NEXT step computes (results lag one step; the buffers are
    double-buffered by the caller).  `begin(buf)` after the step's sync, `finish()` before the buffer is
    reused or the results are read.  Every rank sends `max_n` elements (its shard, padded: the buffers are
    allocated that large), so one `all_gather_into_tensor` does it and concatenating the valid prefixes in
    rank order restores point order.  ~4 bytes per query point per rank: 7/8 of 119 MB arrive at every GPU
    of an 8-GPU run of the headline pair."""

    def __init__(self, max_n, device, dtype=torch.int32):
        self.world = dist.get_world_size()
        self.max_n = int(max_n)
        self.device = device
        self.recv = [torch.empty(self.world * self.max_n, dtype=dtype, device=device) for _ in range(2)]
        self.comm_stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self._pending = None
        self._k = 0

    def begin(self, buf):
        """buf: this rank's result buffer of max_n elements, complete on the host's view (the step has synced)."""
        self.finish()
        out = self.recv[self._k]
        if self.comm_stream is None or _needs_host_staging(buf):
            _all_gather_flat(out, buf[:self.max_n])
            self._pending = None
        else:
            with torch.cuda.stream(self.comm_stream):
                dist.all_gather_into_tensor(out, buf[:self.max_n])
            self._pending = out
        self._last = out
        self._k ^= 1

    def finish(self):
        """-> the last gathered buffer as [world, max_n] (None before the first begin)"""
        if self._pending is not None:
            # a host wait, not a stream wait: the buffer's next writer is the PIP kernel, which may run on the
            # handle's SECOND stream, and that one is ordered behind nothing but the host
            self.comm_stream.synchronize()
            self._pending = None
        return getattr(self, "_last", None).view(self.world, self.max_n) if getattr(self, "_last", None) is not None else None


class PairExchange:
    """One-collective, overlapped all-gather-v of the LSI result queues.

    The sender's buffer is `[count (u64) | pairs ...]` (`send`, int32): the LSI kernel appends
    pairs behind the 8-byte head, `rj_lsi_count_to` drops the device-side count into the head on
    the same stream, and ONE all-gather of the first `2 + 2*slot` ints of every rank ships count and
    pairs together -- no host round trip between the LSI kernel and the exchange.  The collective
    runs on its own stream behind an event, so the PIP kernel that follows the LSI kernel on the
    compute stream overlaps it.  `slot` (pairs shipped per rank) adapts to twice the largest
    count seen; a step whose count exceeds the slot is re-gathered with a larger slot, so the
    result is always complete.

    usage per step:  h.lsi_query_async(..., capacity, ex.pairs);  ex.begin(h);
                     <enqueue more compute>;  views, counts = ex.finish()
    """

    def __init__(self, capacity, device, slot=4096):
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.capacity = int(capacity)
        self.device = device
        self.send = torch.zeros(2 + 2 * self.capacity, dtype=torch.int32, device=device)
        self.pairs = self.send[2:].view(self.capacity, 2)  # hand this to the LSI query
        self.slot = min(int(slot), self.capacity)
        self.recv = None
        self.comm_stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self._ready = None

    def _recv_for(self, slot):
        need = self.world * (2 + 2 * slot)
        if self.recv is None or self.recv.numel() < need:
            self.recv = torch.empty(need, dtype=torch.int32, device=self.device)
        return self.recv[:need]

    def _gather(self, slot, out=None):
        out = self._recv_for(slot) if out is None else out
        _all_gather_flat(out, self.send[:2 + 2 * slot])
        return out.view(self.world, 2 + 2 * slot)

    def begin(self, handle):
        """after the async LSI launch: stamp the count, then start the exchange behind an event.
        The event is recorded on torch's current stream, so the handle must be working on that
        stream (its default is a private one): `handle.set_stream(torch.cuda.current_stream()
        .cuda_stream)` before the LSI launch, or the collective could ship an unfinished queue --
        checked here instead of assumed."""
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device).cuda_stream
            if getattr(handle, "_stream_ptr", None) != cur:
                raise RuntimeError("PairExchange: call handle.set_stream(torch.cuda.current_stream().cuda_stream) "
                                   "before the LSI launch -- the exchange is ordered behind torch's current stream")
        handle.lsi_count_to(self.send)
        if self.comm_stream is None or _needs_host_staging(self.send):
            self._ready = None  # synchronous path (gloo): everything happens in finish()
            return
        out = self._recv_for(self.slot)  # (allocated on the compute stream, where it is consumed)
        ev = torch.cuda.Event()
        ev.record()  # on the compute stream: count + pairs are complete here
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            self._ready = self._gather(self.slot, out)

    def finish(self):
        """-> ([pairs view of rank 0, rank 1, ...], counts list); syncs once"""
        if self._ready is None:
            got = self._gather(self.slot)
        else:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            got = self._ready
        head = got[:, :2].cpu()  # the step's one host sync
        counts = [int(head[r, 0]) & 0xFFFFFFFF | (int(head[r, 1]) & 0xFFFFFFFF) << 32 for r in range(self.world)]
        if counts[self.rank] > self.capacity:
            raise OverflowError("intersection queue overflow: %d found, capacity %d" % (counts[self.rank], self.capacity))
        if max(counts) > self.capacity:
            raise OverflowError("intersection queue overflow on another rank")
        if max(counts) > self.slot:  # rare: grow the slot and gather this step again
            self.slot = min(self.capacity, 2 * max(counts))
            got = self._gather(self.slot)
            if got.is_cuda:
                torch.cuda.current_stream().synchronize()
        elif 4 * max(counts) < self.slot and self.slot > 4096:
            self.slot = max(4096, 2 * max(counts))  # shrink for the next step
        self._ready = None
        return [got[r, 2:2 + 2 * counts[r]].view(-1, 2) for r in range(self.world)], counts
