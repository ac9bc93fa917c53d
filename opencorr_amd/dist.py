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


def allgather_pois(local, n_total, group=None, out=None, async_op=False):
    """Gathers the per-rank POI blocks (n_local x F float32) into the full (n_total x F) queue.

    Blocks are padded to ceil(n_total / G) records so a single fixed-size
    ``all_gather_into_tensor`` suffices; the padding is dropped on return.

    ``out``: optional preallocated (G * ceil(n_total / G), F) buffer to gather into (a steady-state
    pipeline reuses it).  ``async_op=True`` returns ``(queue, work)``: the collective then runs on the
    backend's own stream behind the work already enqueued on the current stream, and the caller's
    later kernels do not wait for it -- the next image pair is correlated while xGMI moves this
    one's records.  Call ``work.wait()`` before reading the queue or reusing the buffers.
    """
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    floats = local.shape[1]
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0], floats), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    if out is None:
        out = torch.empty((world * per, floats), dtype=local.dtype, device=local.device)
    elif tuple(out.shape) != (world * per, floats) or out.dtype != local.dtype or not out.is_contiguous():
        raise ValueError("out must be a contiguous (%d, %d) %s tensor" % (world * per, floats, local.dtype))
    work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    if async_op:
        return out[:n_total], work
    return out[:n_total]
