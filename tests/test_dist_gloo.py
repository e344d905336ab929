"""The N>1 path on CPU: world_size-2 gloo processes shard a video list and gather results."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_videos, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import tspn_mi355x as tspn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = tspn.dist.shard_range(num_videos, rank, world)
    # stand-in for the per-video result rows a rank computes (video id encoded in the values)
    local = torch.stack([torch.full((3, 2), float(v)) for v in range(lo, hi)]) if hi > lo \
        else torch.zeros((0, 3, 2))
    out = tspn.dist.gather_results(local, num_videos)
    q.put((rank, lo, hi, out[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_videos", [8, 5])
def test_shard_and_gather_world2(num_videos):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_videos, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, lo, hi, vals in sorted(res):
        covered += list(range(lo, hi))
        assert vals == [float(v) for v in range(num_videos)]  # global order restored on every rank
    assert covered == list(range(num_videos))  # disjoint, complete, no pad-by-repeat


def test_shard_range_properties(tspn):
    for n in (0, 1, 7, 512):
        for w in (1, 2, 4, 8):
            r = [tspn.dist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == tspn.dist.shard_counts(n, w)
    with pytest.raises(ValueError):
        tspn.dist.shard_range(4, 2, 2)
    assert tspn.dist.gather_results(torch.ones(2, 3), 2).shape == (2, 3)  # no process group: identity
