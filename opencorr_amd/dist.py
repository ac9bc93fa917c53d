"""POI sharding across the GPUs of one node + all-gather of the result records.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo"
in the CPU tests).  Every POI is independent (src/oc_icgn.cpp:343-351), so the queue
is cut into contiguous blocks -- grid-ordered queues keep a rank's POIs spatially
compact, which is what the per-XCD L2 wants -- and the images are replicated.  The only
collective is ONE all-gather of fixed-size POI records after ICGN.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`: blocks of ceil(n/G) POIs, the last one short."""
    per = -(-n_total // world_size)
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def allgather_pois(local, n_total, group=None):
    """Gathers the per-rank POI blocks (n_local x F float32) into the full (n_total x F) queue.

    Blocks are padded to ceil(n_total / G) records so a single fixed-size
    ``all_gather_into_tensor`` suffices; the padding is dropped on return.
    """
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    floats = local.shape[1]
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0], floats), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * per, floats), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_total]
