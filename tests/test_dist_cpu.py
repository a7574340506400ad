"""The N>1 path on CPU: two gloo processes shard a batch, each computes its slice,
results are gathered; must equal the single-process result (weak-scaling data
parallelism with no data-path collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dkt_stereo_amd.shard import gather_disparity, shard_batch, shard_bounds


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_inference(x):
    # stands in for the per-pair network: any per-sample function
    return (x * 2.0 + x.flip(-1)).sum(dim=1, keepdim=True)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        batch = torch.randn(total, 3, 6, 10, generator=g)
        mine = shard_batch(batch)
        out = _fake_inference(mine)
        full = gather_disparity(out, total)            # all_gather flavour
        only0 = gather_disparity(out, total, dst=0)    # gather-to-rank-0 flavour
        ok = torch.equal(full, _fake_inference(batch))
        ok0 = (only0 is None) if rank != 0 else torch.equal(only0, full)
        q.put((rank, bool(ok and ok0), tuple(mine.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
def test_two_rank_shard_and_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(s[0] for _, _, s in res) == total
