"""N>1 host logic on CPU: world_size-2 and -3 gloo runs of the sharding + all-gather-v code that
bench.py uses on RCCL.  The per-shard 'operator' here is the oracle (test infrastructure) so the
exchange logic is checked without a GPU: union of shard results == whole-map result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import rjoracle as O
    from rayjoin_amd import dist as rjd
    from rayjoin_amd import maps, synth
    O.lib().rjo_set_num_threads(2)
    ctx = maps.Context([synth.lattice_map(5, 70, 41), synth.lattice_map(11, 33, 42)]).load()
    base, query = ctx.maps
    sh = rjd.shard_of(query, world, rank)
    m0 = O.Map(base.pts, base.row_index, base.left, base.right)
    m1 = O.Map(query.pts, query.row_index, query.left, query.right)
    whole = O.lsi_brute(m0, m1)
    e0, e1 = sh["eids"]
    mine = whole[(whole[:, 1] >= e0) & (whole[:, 1] < e1)]
    cap = len(whole) + 8
    buf = torch.zeros((cap, 2), dtype=torch.int32)
    buf[:len(mine)] = torch.from_numpy(mine.astype(np.int32))
    gathered, counts = rjd.allgather_pairs(buf, len(mine))
    got = O.sort_pairs(gathered.numpy().astype(np.uint32))
    assert int(counts.sum()) == len(whole)
    assert np.array_equal(got, whole), "rank %d: gathered LSI pairs differ from the whole-map result" % rank
    # the one-collective exchange bench.py uses on RCCL (count rides at the head of the buffer);
    # slot 16 is far too small on purpose: the first finish() must grow it and re-gather
    class _FakeHandle:  # stands in for rj_lsi_count_to: the device-side count copy
        count = len(mine)

        def lsi_count_to(self, send):
            send[0] = self.count
            send[1] = 0
    fake = _FakeHandle()
    ex = rjd.PairExchange(fake, cap, torch.device("cpu"), slot=16)
    assert not ex.native  # (gloo: the test transport; on "nccl" the same class drives rj_exchange_* of the C ABI)
    for rep in range(4):
        k = rep % 2       # both exchange buffers
        ex.pairs[k][:len(mine)] = torch.from_numpy(mine.astype(np.int32))
        ex.begin(k)
        views, cl = ex.finish(k)
        assert cl == [int(c) for c in counts] and ex.slot >= max(cl)
        got2 = O.sort_pairs(torch.cat(views).numpy().astype(np.uint32))
        assert np.array_equal(got2, whole), "rank %d rep %d: PairExchange result differs" % (rank, rep)
    # one rank's queue overflowed its capacity: EVERY rank learns it from the gathered heads and raises -- nobody walks
    # into a further collective alone (round 3's native form returned early on the overflowing rank only)
    fake.count = cap + 5 if rank == world - 1 else len(mine)
    ex.begin(0)
    try:
        ex.finish(0)
        raise AssertionError("rank %d: overflow on rank %d went unnoticed" % (rank, world - 1))
    except OverflowError as e:
        assert "rank %d" % (world - 1) in str(e)
    fake.count = len(mine)
    ex.begin(0)
    assert ex.finish(0)[1] == [int(c) for c in counts]  # ... and the exchange works again afterwards
    # PIP: contiguous point shards concatenate back in point order
    p0, p1 = sh["points"]
    want = O.pip_brute(m0, 1, query.pts)
    ids = torch.from_numpy(want[p0:p1].astype(np.int32))
    max_n = max(b - a for a, b in (rjd.shard_of(query, world, r)["points"] for r in range(world)))
    allids = rjd.allgather_point_results(ids, p1 - p0, max_n)
    assert np.array_equal(allids.numpy().astype(np.uint32), want)
    # ... and the pipelined form bench.py times (begin after a step, finish a step later, double-buffered):
    # two "steps" with different contents, each gathered buffer complete when it is read
    pg = rjd.PointGather(None, max_n, torch.device("cpu"))
    for rep in range(3):
        buf = torch.full((max_n,), -1, dtype=torch.int32)
        buf[:p1 - p0] = ids + rep
        pg.begin(buf)
        got = pg.finish()
        cat = torch.cat([got[r, :b - a] for r, (a, b) in enumerate(rjd.shard_of(query, world, r)["points"] for r in range(world))])
        assert np.array_equal((cat - rep).numpy().astype(np.uint32), want), "rank %d rep %d: PointGather" % (rank, rep)
    # shards tile the chain range exactly and are balanced by edge count
    rngs = query.shard_chain_ranges(world)
    assert rngs[0][0] == 0 and rngs[-1][1] == query.n_chains
    assert all(rngs[i][1] == rngs[i + 1][0] for i in range(world - 1))
    sizes = [query.chain_range_to_eids(a, b) for a, b in rngs]
    ne = [b - a for a, b in sizes]
    assert sum(ne) == query.n_edges and max(ne) - min(ne) <= 2 * 33
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_and_allgatherv_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("ok%d" % r))


def test_allgatherv_plan_offsets_for_ragged_and_empty_shards():
    """The layout rj_allgather_pairs / rj_allgather_u32 compute between their two collectives (counts -> offsets,
    total, overflow) as a pure host function of the C ABI: RCCL itself cannot run with more than one rank on a
    one-GPU lease, the arithmetic that places every rank's slice can be checked for any N."""
    import numpy as np
    from rayjoin_amd import _capi
    rng = np.random.default_rng(5)
    cases = [[0], [7], [0, 0], [5, 0], [0, 9], [3, 4, 5], [0, 0, 0], [1, 0, 2], [2 ** 40, 1, 2 ** 41],
             [0, 0, 6, 0, 0, 0, 1, 0], [11, 0, 13, 17, 0, 19, 23, 29]]
    cases += [list(rng.integers(0, 1 << 20, n)) for n in (2, 3, 8) for _ in range(4)]
    for counts in cases:
        off, total, rc = _capi.allgatherv_plan(counts, sum(counts))
        assert rc == _capi.RJ_OK and total == sum(counts)
        assert list(off) == [sum(counts[:r]) for r in range(len(counts))]
        # slices tile [0, total) exactly, in rank order, empty ranks take no room
        assert all(int(off[r]) + counts[r] == (int(off[r + 1]) if r + 1 < len(counts) else total) for r in range(len(counts)))
        if total:
            off2, total2, rc2 = _capi.allgatherv_plan(counts, total - 1)
            assert rc2 == _capi.RJ_E_OVERFLOW and total2 == total and list(off2) == list(off)  # (the true total is still reported)
    assert _capi.allgatherv_plan([], 10)[2] == _capi.RJ_E_INVALID
    assert _capi.allgatherv_plan([2 ** 63, 2 ** 63], 2 ** 64 - 1)[2] == _capi.RJ_E_INVALID  # the sum does not fit 64 bits


def test_exchange_verdict_is_the_same_on_every_rank():
    """What a rank does after the heads have been gathered is rj_exchange_verdict of the gathered words -- the same words
    on every rank, so the same branch: overflow anywhere is overflow everywhere, the largest count sizes the re-gather."""
    from rayjoin_amd import _capi
    assert _capi.exchange_verdict([3, 0, 9], [10, 10, 10]) == (_capi.RJ_OK, 9, -1)
    assert _capi.exchange_verdict([3, 11, 9], [10, 10, 10]) == (_capi.RJ_E_OVERFLOW, 11, 1)
    assert _capi.exchange_verdict([3, 4, 9], [10, 3, 8]) == (_capi.RJ_E_OVERFLOW, 9, 1)   # ranks with different capacities
    assert _capi.exchange_verdict([0] * 8, [0] * 8) == (_capi.RJ_OK, 0, -1)
    assert _capi.exchange_verdict([2 ** 40], [2 ** 40 - 1]) == (_capi.RJ_E_OVERFLOW, 2 ** 40, 0)
    assert _capi.exchange_verdict([], [])[0] == _capi.RJ_E_INVALID
