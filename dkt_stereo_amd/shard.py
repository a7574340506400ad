"""Multi-GPU data parallelism for stereo inference: one process per GPU,
contiguous batch shards, no data-path collective, one result gather.

The reference scales with ``nn.DataParallel`` (tools/ft_dkt.py:119,
tools/evaluate_stereo.py:361): one process, a thread per GPU, parameters
re-broadcast on every forward.  Stereo pairs are independent, so here each rank
owns a replica of the weights (loaded once) and a slice of the batch; the only
communication is the gather of the final (B/N,1,H,W) disparity maps
(``torch.distributed``: RCCL over xGMI with backend "nccl", gloo on CPU).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous [lo, hi) slice of `total` items for `rank`; the first
    total % world_size ranks take one extra (same rule as torch.chunk on
    evenly divisible sizes, well defined when it is not)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size %r/%r" % (rank, world_size))
    q, r = divmod(total, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(tensor, world_size=None, rank=None):
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(tensor.shape[0], world_size, rank)
    return tensor[lo:hi]


def gather_disparity(local, total, dst=None):
    """Gathers per-rank (b_i,1,H,W) results into the full (total,1,H,W) batch.
    dst=None: all_gather (every rank gets the result); dst=k: only rank k does
    (others return None).  Shards may be uneven (padded to the largest)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return local
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < maxb:
        pad = torch.cat([local, local.new_zeros((maxb - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    pad = pad.contiguous()
    # gloo moves device tensors for broadcast / all_reduce only: its gather of device results is staged through the host
    # (bench.py --dist-backend gloo: two ranks sharing one device in the GPU suite; RCCL takes the device tensors as they are)
    staged = pad.is_cuda and dist.get_backend() == "gloo"
    if staged:
        pad = pad.cpu()
    if dst is None:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        if rank != dst:
            return None
    full = torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    return full.to(local.device) if staged else full
