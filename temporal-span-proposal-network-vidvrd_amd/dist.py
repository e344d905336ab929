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


# ---- the payload of the final gather: decoded results, not logits ------------------------------
# Per video the prediction loop keeps the top-200 (score, triplet, pair) rows (reference
# lib/modeling/predict.py:106-116) and, with PPN, the top-k pair indices: ~9-11 KB per video instead of
# 524 KB of logits.  The fields are packed into ONE byte row per video so that the
# exchange stays a single collective; `unpack_decoded` restores the typed views.
_DECODED_FIELDS = (("scores", torch.float32, 1), ("triplets", torch.int64, 3), ("pair_tids", torch.int64, 2))


def pack_decoded(scores, triplets, pair_tids, pair_proposals=None):
    """[B,M] fp32, [B,M,3] int64, [B,M,2] int64 (+ [B,k] int64) -> uint8 [B, row_bytes]."""
    b = scores.shape[0]

    def as_bytes(x):   # explicit row width: reshape(b, -1) is ambiguous for an empty shard (b == 0)
        x = x.contiguous()
        row = (x.numel() // b if b else int(torch.tensor(x.shape[1:]).prod())) * x.element_size()
        return x.view(torch.uint8).reshape(b, row)

    parts = [as_bytes(scores), as_bytes(triplets), as_bytes(pair_tids)]
    if pair_proposals is not None:
        parts.append(as_bytes(pair_proposals))
    return torch.cat(parts, dim=1)


def unpack_decoded(packed, m, k=0):
    """Inverse of pack_decoded for M = m rows per video and k pair proposals."""
    b = packed.shape[0]
    out, off = {}, 0
    for name, dtype, width in _DECODED_FIELDS:
        nbytes = m * width * torch.empty((), dtype=dtype).element_size()
        out[name] = packed[:, off:off + nbytes].contiguous().view(dtype).reshape((b, m) if width == 1 else (b, m, width))
        off += nbytes
    if k:
        out["pair_proposals"] = packed[:, off:off + 8 * k].contiguous().view(torch.int64).reshape(b, k)
        off += 8 * k
    if off != packed.shape[1]:
        raise ValueError(f"unpack_decoded: row has {packed.shape[1]} bytes, fields need {off}")
    return out


def gather_decoded(scores, triplets, pair_tids, num_items, pair_proposals=None, group=None, force=False):
    """The one collective of the path: all-gather of the decoded per-video results in global video order.
    Returns the dict of `unpack_decoded` over all `num_items` videos."""
    m = scores.shape[1]
    k = pair_proposals.shape[1] if pair_proposals is not None else 0
    packed = gather_results(pack_decoded(scores, triplets, pair_tids, pair_proposals), num_items, group=group,
                            force=force)
    return unpack_decoded(packed, m, k)
