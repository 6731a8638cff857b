"""Multi-GPU sharding of the query map and gathering of result queues.

The reference is single-GPU (no NCCL/MPI anywhere); this is new design.  Each query segment /
point is independent against a read-only base map + LBVH, so the query map is split into
contiguous CHAIN ranges balanced by edge count (a chain range is a contiguous eid range and a
contiguous point range, because eid = p_idx - ichain, src/map/map.h:200-203), the base map and
its LBVH are replicated, and the only exchange step is an all-gather(v) of the result queues:
one collective per queue, heads (count, capacity) riding in front of the payload.

ONE exchange implementation: the library's (rj_exchange_* in include/rayjoin_amd.h -- RCCL through the handle's own
communicators, what query_exec -nranks and bench.py at N > 1 both run).  torch.distributed is the launcher's plumbing
here: rendezvous, the broadcast of the RCCL id, barriers.  Where RCCL cannot run -- the CPU tests (world 2/3 on gloo)
and the rehearsal of N ranks on a one-GPU box -- PairExchange / PointGather keep the same buffer layout and take the
same decisions from the same host function (rj_exchange_verdict) over a gloo all-gather: a test transport, named so.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _capi

HEAD = _capi.RJ_EXCHANGE_HEAD_WORDS  # 32-bit words in front of an exchange buffer's pairs: count (u64), capacity (u64)


def native_transport(device):
    """RCCL through the handle (the product path) -- unless the process group is gloo (CPU tests, one-GPU rehearsal)"""
    return device.type == "cuda" and dist.get_backend() == "nccl"


def ensure_comm(handle):
    """rj_comm_init on this rank's handle, once: rank 0 makes the RCCL id, torch.distributed carries it"""
    if getattr(handle, "_comm_ready", False):
        return
    uid = [_capi.Handle.comm_unique_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    handle.comm_init(dist.get_world_size(), dist.get_rank(), uid[0])
    handle._comm_ready = True


class _DevView:
    """a device pointer as something torch.as_tensor takes without a copy"""

    def __init__(self, ptr, shape, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


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

    A step's PIP kernels are the last thing the step runs, so its gather has nothing of the SAME step to hide
    behind; it runs while the NEXT step computes (results lag one step; the buffers are double-buffered by the
    caller) -- on the handle's SECOND communicator and a stream of its own (rj_exchange_u32_begin), so it never queues
    in front of the next step's pair exchange (two collectives of one communicator serialise: the advisor's round-3
    finding about the torch form).  `begin(buf)` after the step, `finish()` before the buffer is reused or the
    results are read.  Every rank sends `max_n` elements (its shard, padded: the buffers are allocated that large);
    concatenating the valid prefixes in rank order restores point order.  ~4 bytes per query point per rank: 7/8 of
    119 MB arrive at every GPU of an 8-GPU run of the headline pair."""

    def __init__(self, handle, max_n, device, dtype=torch.int32):
        self.h = handle
        self.world = dist.get_world_size()
        self.max_n = int(max_n)
        self.device = device
        self.native = native_transport(device)
        if self.native:
            ensure_comm(handle)
        self.recv = [torch.empty(self.world * self.max_n, dtype=dtype, device=device) for _ in range(2)]
        self._pending = False
        self._last = None
        self._k = 0

    def begin(self, buf):
        """buf: this rank's result buffer of max_n elements"""
        self.finish()
        out = self.recv[self._k]
        if self.native:
            self.h.exchange_u32_begin(buf, self.max_n, out)  # behind everything enqueued on the handle's streams
            self._pending = True
        else:
            _all_gather_flat(out, buf[:self.max_n])
        self._last = out
        self._k ^= 1

    def finish(self):
        """-> the last gathered buffer as [world, max_n] (None before the first begin)"""
        if self._pending:
            self.h.exchange_u32_finish()  # a host wait: the buffer's next writer is a kernel on another stream
            self._pending = False
        return self._last.view(self.world, self.max_n) if self._last is not None else None


class PairExchange:
    """One-collective, overlapped all-gather-v of the LSI result queues (rj_exchange_* of the C ABI).

    An exchange buffer is `[count (u64) | capacity (u64) | pairs ...]` (int32 words): the LSI kernel appends pairs
    behind the head, `begin` has the device-side count dropped into the head on the handle's stream and ONE all-gather
    of the first `HEAD + 2*slot` words of every rank ships count and pairs together -- no host round trip between the
    LSI kernel and the exchange -- on the handle's communication stream behind an event, so whatever follows the
    LSI kernel on the compute streams overlaps it.  `slot` (pairs shipped per rank) adapts to twice the largest count
    seen; a step whose count exceeds the slot is gathered again with a larger slot -- by every rank, all of which
    see the same counts -- so the result is always complete.  Two buffers: step k + 1 may be launched into the other
    one before step k's `finish`.

    usage per step:  h.lsi_query_async(..., capacity, ex.pairs[k]);  ex.begin(k);
                     <enqueue more compute, the next step>;  views, counts = ex.finish(k)
    """

    def __init__(self, handle, capacity, device, slot=4096, nbuf=2):
        self.h = handle
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.capacity = int(capacity)
        self.device = device
        self.native = native_transport(device)
        self.send = [torch.zeros(HEAD + 2 * self.capacity, dtype=torch.int32, device=device) for _ in range(nbuf)]
        self.pairs = [b[HEAD:].view(self.capacity, 2) for b in self.send]  # hand pairs[k] to the LSI query
        self.slot = min(int(slot), self.capacity)
        if self.native:
            ensure_comm(handle)
            handle.exchange_init(self.capacity, self.slot, self.send[0], self.send[1] if nbuf > 1 else None)
        else:
            cap_words = torch.tensor([self.capacity, 0], dtype=torch.int32, device=device)
            for b in self.send:
                b[2:4] = cap_words
            self.recv = [None] * nbuf

    # ---- the test transport (gloo): the same layout, the same verdict function, a synchronous all-gather ----
    def _gather_gloo(self, k, slot):
        need = self.world * (HEAD + 2 * slot)
        if self.recv[k] is None or self.recv[k].numel() < need:
            self.recv[k] = torch.empty(need, dtype=torch.int32, device=self.device)
        out = self.recv[k][:need]
        _all_gather_flat(out, self.send[k][:HEAD + 2 * slot])
        return out.view(self.world, HEAD + 2 * slot)

    def begin(self, k=0):
        """after the async LSI launch into pairs[k]"""
        if self.native:
            self.h.exchange_pairs_begin(k)
        else:
            self.h.lsi_count_to(self.send[k])  # (everything else happens in finish())

    def finish(self, k=0):
        """-> ([pairs view of rank 0, rank 1, ...], counts list); the step's one host sync on the LSI side.
        Raises OverflowError on EVERY rank when some rank's queue overflowed."""
        if self.native:
            try:
                counts, ptrs, _ = self.h.exchange_pairs_finish(k, self.world)
            except _capi.QueueOverflow as e:
                raise OverflowError(str(e))
            views = [torch.as_tensor(_DevView(p, (c, 2)), device=self.device) if c else self.pairs[k][:0] for p, c in zip(ptrs, counts)]
            return views, counts
        got = self._gather_gloo(k, self.slot)
        head = got[:, :HEAD].cpu().numpy().astype(np.uint32).astype(np.uint64)
        counts = (head[:, 0] | (head[:, 1] << np.uint64(32))).tolist()
        caps = (head[:, 2] | (head[:, 3] << np.uint64(32))).tolist()
        rc, mx, bad = _capi.exchange_verdict(counts, caps)
        if rc == _capi.RJ_E_OVERFLOW:
            raise OverflowError("intersection queue overflow on rank %d: %d found, capacity %d" % (bad, counts[bad], caps[bad]))
        if mx > self.slot:  # rare: grow the slot and gather this step again
            self.slot = min(self.capacity, 2 * mx)
            got = self._gather_gloo(k, self.slot)
        elif 4 * mx < self.slot and self.slot > 4096:
            self.slot = max(4096, 2 * mx)  # shrink for the next step
        return [got[r, HEAD:HEAD + 2 * counts[r]].view(-1, 2) for r in range(self.world)], counts
