"""Multi-GPU plan for the hot path: videos are independent, so they shard.

One process per GPU (mirrors the reference's `mp.spawn`, base.py:65, and the rank
slicing of its DistributedSampler, lib/dataset/samplers/distributed.py:55-57 —
without the pad-by-repeat, since results must not be duplicated).  Weights are
replicated; the forward itself has NO collective.  The only exchange is one
final gather of the (small) per-video results, over RCCL on GPUs
(`torch.distributed` backend "nccl") or gloo in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(num_items, rank, world_size):
    """Contiguous block partition: rank r gets [lo, hi); sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(num_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(num_items, world_size):
    return [shard_range(num_items, r, world_size)[1] - shard_range(num_items, r, world_size)[0]
            for r in range(world_size)]


def gather_results(local, num_items, group=None, force=False):
    """All-gather per-video result rows back into global video order.

    `local` [n_local, ...]: this rank's results for its `shard_range` block.
    Returns [num_items, ...] on every rank.  Ragged shards use the pad-to-max
    scheme (one collective), cf. the unused reference helper lib/utils/comm.py:60-81.
    `force` issues the collective even for a single rank (used to exercise the RCCL path).
    """
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local
    counts = shard_counts(num_items, world)
    rank = dist.get_rank(group)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank}: got {local.shape[0]} rows, shard has {counts[rank]}")
    mx = max(counts)
    if local.shape[0] < mx:
        pad = local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))
        local = torch.cat([local, pad])
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)])
