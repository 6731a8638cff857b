"""The overlapped one-collective exchange of bench.py's N>1 step, on a single-rank RCCL group:
async LSI -> device-side count into the buffer head -> all-gather on a second stream while the
PIP kernel runs -> one host sync.  Results against the oracle."""
import os
import socket

import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def test_pair_exchange_rccl_single_rank(oracle):
    import torch
    import torch.distributed as dist
    from rayjoin_amd import dist as rjd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx = maps.Context([synth.lattice_map(9, 120, 31), synth.lattice_map(21, 50, 32)]).load()
        base, query = ctx.maps
        h = _capi.Handle(0)
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        h.upload_map(0, base.pts, base.row_index, base.left, base.right)
        h.upload_map(1, query.pts, query.row_index, query.left, query.right)
        h.build_lbvh(0)
        m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
        m1 = oracle.Map(query.pts, query.row_index, query.left, query.right)
        want = oracle.lsi_brute(m0, m1)
        want_pip = oracle.pip_brute(m0, 1, query.pts)
        cap = 4 * len(want)
        ex = rjd.PairExchange(h, cap, dev, slot=64)  # too small on purpose: first step must re-gather
        assert ex.native  # RCCL through the handle's communicator (rj_exchange_*), not torch's
        closest = torch.empty(query.n_points, dtype=torch.int32, device=dev)
        pg = rjd.PointGather(h, query.n_points, dev)
        for rep in range(4):
            k = rep % 2
            h.lsi_query_async(0, 1, 0, query.n_edges, cap, ex.pairs[k])
            ex.begin(k)
            h.pip_query(0, 1, None, 0, query.n_points, closest, None, sync=False)
            pg.begin(closest)  # behind the PIP kernels, on the second communicator's stream
            views, counts = ex.finish(k)
            assert counts == [len(want)]
            assert np.array_equal(pg.finish()[0].cpu().numpy().astype(np.uint32), want_pip)
            got = views[0].clone()
            h.sort_pairs(got, counts[0])
            assert np.array_equal(got.cpu().numpy().astype(np.uint32), want)
            torch.cuda.synchronize()
            assert np.array_equal(closest.cpu().numpy().astype(np.uint32), want_pip)
        # overflow is reported, not silently truncated
        small = rjd.PairExchange(h, 8, dev, slot=8, nbuf=1)
        h.lsi_query_async(0, 1, 0, query.n_edges, 8, small.pairs[0])
        small.begin(0)
        with pytest.raises(OverflowError):
            small.finish(0)
        # the synchronous exact form on the same communicator
        out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        n = h.lsi_query(0, 1, 0, query.n_edges, cap, out)
        flat = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        total, counts = h.allgather_pairs(out, n, flat, cap)
        assert total == n == len(want) and list(counts) == [n]
        h.sort_pairs(flat, n)
        assert np.array_equal(flat[:n].cpu().numpy().astype(np.uint32), want)
        with pytest.raises(_capi.RayJoinError):
            h.allgather_pairs(out, n, flat, n - 1)
        h.close()
    finally:
        dist.destroy_process_group()
