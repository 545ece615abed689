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


def gather_slabs(local, counts=None, group=None, out=None, scratch=None):
    """All-gather per-rank output slabs along dim 0 into the global tensor.

    ``local``   this rank's [views_r, ...] tensor (device tensor under nccl,
                CPU tensor under gloo);
    ``counts``  views per rank when they differ (ragged shards are padded to the
                largest slab so the exchange stays one fused collective);
    ``out``     the global [sum(views_r), ...] tensor of a previous call to
                receive into again;
    ``scratch`` a dict the caller keeps between calls: the padded send / receive
                buffers of the ragged path live in it, so that a per-step gather
                allocates nothing in either path.
    World size 1 is a no-op that returns ``local`` itself."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    if ws == 1:
        return local
    tail = tuple(local.shape[1:])
    if counts is None or len(set(counts)) == 1:
        shape = (ws * local.shape[0],) + tail
        if out is None or tuple(out.shape) != shape:
            out = local.new_empty(shape)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged shards: pad every slab to the largest, one fused collective, trim
    if len(counts) != ws:
        raise ValueError("counts has %d entries for %d ranks" % (len(counts), ws))
    rank = dist.get_rank(group)
    if local.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d views, counts says %d" % (rank, local.shape[0], counts[rank]))
    top = max(counts)
    if scratch is None:
        scratch = {}
    key = (top, ws, tail, local.dtype, local.device)
    if scratch.get("key") != key:
        scratch["key"] = key
        scratch["send"] = local.new_zeros((top,) + tail)
        scratch["recv"] = local.new_empty((ws * top,) + tail)
    send, recv = scratch["send"], scratch["recv"]
    send[:local.shape[0]] = local
    dist.all_gather_into_tensor(recv, send, group=group)
    shape = (sum(counts),) + tail
    if out is None or tuple(out.shape) != shape:
        out = local.new_empty(shape)
    torch.cat([recv[r * top:r * top + c] for r, c in enumerate(counts)], dim=0, out=out)
    return out
