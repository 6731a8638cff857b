"""The N>1 host path on the HIP operators (not on the oracle): bench.py launched exactly as the
driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`),
as fresh child processes, with every rank on the box's one GPU (`--rehearse-one-gpu`: collectives
over gloo).  Chain-range shards + exchange must reproduce the single-GPU results: intersection
count and an order-independent digest of pairs, intersection points, PIP eids and face ids."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(nproc, extra=(), plain=False):
    common = ["bench.py", "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--scale", "0.12", "--no-cpu-baseline"]
    if plain:  # the driver's single-GPU form with N > 1: bench.py starts its own ranks (a child torch.distributed.run)
        cmd = [sys.executable] + common + ["--rehearse-one-gpu"]
    elif nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + common + ["--rehearse-one-gpu"]
    else:
        cmd = [sys.executable] + common
    r = subprocess.run(cmd + list(extra), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_and_three_ranks_reproduce_the_single_gpu_results():
    one = _bench(1)
    assert one["n_gpus"] == 1 and one["intersections"] > 1000
    assert one["result_digest"]["pip_hits"] > 100000
    for nproc, extra in ((2, ()), (3, ("--gather-pip",))):
        many = _bench(nproc, extra)
        assert many["n_gpus"] == nproc and many["scaling"] == "strong"
        assert many["intersections"] == one["intersections"]
        assert many["result_digest"] == one["result_digest"], (nproc, many["result_digest"], one["result_digest"])
        assert "chain range x%d" % nproc in many["config"]["sharding"]


def test_plain_command_with_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without a launcher around it must not print a one-rank line that says two."""
    one = _bench(1)
    two = _bench(2, plain=True)
    assert two["n_gpus"] == 2 and two["ranks"]["world_size"] == 2
    assert "chain range x2" in two["config"]["sharding"]
    assert two["intersections"] == one["intersections"] and two["result_digest"] == one["result_digest"]
