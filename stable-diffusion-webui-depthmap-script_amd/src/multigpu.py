"""Multi-GPU layer: shard independent units (images / frames) across ranks, one process per GPU, and collate the
results on rank 0 with ONE gather (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU in the tests).

The reference is single-process and loops over images (src/core.py:133); images are independent units
(SURVEY.md 8e), so unit i goes to rank i // ceil(n / world) -- contiguous blocks keep the output order -- and the
only communication is the gather of the collated output.  No collective is needed on the data path itself.
"""
import math


def shard_bounds(n_units, world_size):
    """Contiguous [start, end) per rank; the first ranks get the larger shards; empty shards are allowed."""
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    per = int(math.ceil(n_units / world_size)) if n_units > 0 else 0
    out = []
    for r in range(world_size):
        s = min(r * per, n_units)
        e = min(s + per, n_units)
        out.append((s, e))
    return out


def my_shard(n_units, rank, world_size):
    return shard_bounds(n_units, world_size)[rank]


def global_rank(group, group_rank):
    """The GLOBAL rank of member `group_rank` of `group`: what torch.distributed's src= / dst= arguments mean, also when
    the collective runs on a sub-group (dist.get_rank(group) is group-local)."""
    import torch.distributed as dist
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def gather_units(local, n_units, group=None, dst=0):
    """Collate per-rank results on rank `dst` (a rank OF THE GROUP: sub-groups are translated for the collective).

    local: tensor [n_local, ...] with this rank's units (n_local may be 0, and differs by at most `per` across ranks).
    Returns the full [n_units, ...] tensor on `dst`, None elsewhere.  One gather: shards are padded to the common
    size `per` so a single fixed-size collective moves everything.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(n_units, world)
    per = max(e - s for s, e in bounds)
    s, e = bounds[rank]
    assert local.shape[0] == e - s, (local.shape, bounds[rank])
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    else:
        send = local.contiguous()
    gdst = global_rank(group, dst)
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=gdst, group=group)
        parts = [bufs[r][: bounds[r][1] - bounds[r][0]] for r in range(world)]
        return torch.cat(parts, dim=0)
    dist.gather(send, None, dst=gdst, group=group)
    return None


def render_sharded(images, depth, render_fn, group=None, dst=0):
    """Run `render_fn(images_shard, depth_shard) -> tensor [n_local, ...]` on this rank's shard of the batch and
    gather the results to `dst`.  `images`/`depth` are the FULL batch (every rank holds or can index it); only the
    rank's own slice is touched."""
    import torch.distributed as dist
    n = images.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    s, e = my_shard(n, rank, world)
    local = render_fn(images[s:e], depth[s:e])
    return gather_units(local, n, group=group, dst=dst)


def pack_collated(parts):
    """The collated output of a shard of units -- e.g. [stereo pairs uint8 [n,H,2W,3], depth uint16 [n,H,W], normal maps uint8
    [n,H,W,3]] -- as ONE byte buffer [n, bytes per unit], so that a single gather moves all of it (north_star: "a single RCCL
    gather over xGMI for the collated output").  Returns (packed, layout); `unpack_collated(buf, layout)` restores the tensors."""
    import torch
    n = parts[0].shape[0]
    flat, layout = [], []
    for t in parts:
        assert t.shape[0] == n and t.is_contiguous()
        b = t.view(torch.uint8) if t.dtype != torch.uint8 else t
        flat.append(b.reshape(n, -1))
        layout.append((tuple(t.shape[1:]), t.dtype, flat[-1].shape[1]))
    return torch.cat(flat, dim=1), layout


def unpack_collated(buf, layout):
    out, o = [], 0
    n = buf.shape[0]
    for shape, dtype, nbytes in layout:
        piece = buf[:, o:o + nbytes].contiguous()
        out.append(piece.view(dtype).reshape((n,) + shape))
        o += nbytes
    return out
