"""-m gpu: the N>1 path over RCCL (backend "nccl") -- two processes, one device each, shard a batch, run the correlation
hot path on their slice, gather the results; equals the single-process result.  Needs two visible devices: skipped on the
1-GPU boxes `gpurun` hands out, runs on the driver's 8-GPU node.  The gloo twin (tests/test_dist_cpu.py) covers the same
code on CPU."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dist_cpu import _free_port

pytestmark = pytest.mark.gpu


def _per_pair(f1, f2):
    from dkt_stereo_amd.corr import CorrBlock1D
    b, _, h, w = f1.shape
    blk = CorrBlock1D(f1, f2, radius=4, num_levels=4)
    coords = torch.arange(w, device=f1.device, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w) - 3.25
    return blk(torch.cat([coords, torch.zeros_like(coords)], 1).contiguous())


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from dkt_stereo_amd.shard import gather_disparity, shard_batch
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        g = torch.Generator().manual_seed(0)
        f1 = torch.randn(total, 32, 12, 48, generator=g).to(dev)
        f2 = torch.randn(total, 32, 12, 48, generator=g).to(dev)
        with torch.no_grad():
            mine = _per_pair(shard_batch(f1), shard_batch(f2))
            full = gather_disparity(mine, total)
            only0 = gather_disparity(mine, total, dst=0)
            want = _per_pair(f1, f2)
        ok = torch.equal(full, want) and ((only0 is None) if rank != 0 else torch.equal(only0, full))
        q.put((rank, bool(ok), dist.get_backend(), int(mine.shape[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices (RCCL)")
@pytest.mark.parametrize("total", [8, 5])
def test_two_rank_shard_and_gather_over_rccl(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert {b for _, _, b, _ in res} == {"nccl"}
    assert sum(n for _, _, _, n in res) == total
