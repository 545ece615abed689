"""World sharding across the GPUs of one node (one process per GPU).

Worlds are independent, so the render itself needs no collective: rank r owns a
contiguous world range and its outputs are one contiguous slab of the global
[views,H,W,C] tensors.  ``gather_slabs`` is the optional
exchange step -- an all-gather over RCCL (backend "nccl") on xGMI, or gloo on
CPU tensors in the tests.
"""
import torch
import torch.distributed as dist

from .scenes import shard_range


def view_ranges(worlds, world_size):
    """[(view_lo, view_hi)] per rank for a list of (ni, io, nc, co) worlds."""
    out = []
    cams = [w[2] for w in worlds]
    for r in range(world_size):
        lo, hi = shard_range(len(worlds), r, world_size)
        out.append((sum(cams[:lo]), sum(cams[:hi])))
    return out


def gather_slabs(local, counts=None, group=None, out=None):
    """All-gather per-rank output slabs along dim 0 into the global tensor.

    ``local``  this rank's [views_r, ...] tensor (device tensor under nccl,
               CPU tensor under gloo);
    ``counts`` views per rank when they differ (ragged shards are padded to the
               largest slab so the exchange stays one fused collective);
    ``out``    a [world_size * max(views_r), ...] tensor of a previous call to
               receive into again (a per-step gather allocates nothing).
    World size 1 is a no-op that returns ``local`` itself."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    if ws == 1:
        return local
    if counts is None or len(set(counts)) == 1:
        shape = (ws * local.shape[0],) + tuple(local.shape[1:])
        if out is None or tuple(out.shape) != shape:
            out = local.new_empty(shape)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged shards: pad every slab to the largest, one fused collective, trim
    top = max(counts)
    padded = local.new_zeros((top,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    shape = (ws * top,) + tuple(local.shape[1:])
    if out is None or tuple(out.shape) != shape:
        out = local.new_empty(shape)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * top:r * top + c] for r, c in enumerate(counts)], dim=0)
