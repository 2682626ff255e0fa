"""Video mode of the funnel, frame parallel.

Reference: src/video_mode.py -- process_predicitons :103-128, gen_video :131-190 (first pass: raw predictions of all
frames; global normalisation; second pass: the funnel again with the normalised predictions as custom depth maps).

What is built: the COMPUTE of video mode on frame tensors --
  * ``process_predicitons`` (same name, same semantics, numpy in / numpy out) and
  * ``process_predictions_sharded``: the same normalisation when the frames are sharded over ranks (one process per
    GPU): 'none' needs the global min/max -> ONE all-reduce(MIN) of the pair (min, -max); 'experimental' needs
    +-2 frames of halo at the shard edges (an all-gather of every shard's first / last two frames) and the global
    0.5 / 99.5 percentiles, found
    EXACTLY by bisection on the float32 bit pattern with one all-reduce(SUM) of a counter per step (numpy's linear
    interpolation between the two neighbouring order statistics reproduced) -- no rank ever holds all frames;
  * ``gen_frames_sharded``: two passes over this rank's contiguous block of frames (network batched on the device,
    normalisation with the collectives above, depth -> uint16 -> stereo on the device) and a single gather of the
    collated output to rank 0 (src/multigpu.gather_units; RCCL over xGMI under the 'nccl' backend).
Container I/O (moviepy / imageio-ffmpeg / av: open_path_as_images :13-64, frames_to_video :67-100) is outside the hot
path and not built; callers hand in decoded frames.
"""
import numpy as np


def process_predicitons(predictions, smoothening='none'):
    """Drop-in for the reference's process_predicitons (:103-128; the spelling is the reference's): a list of [H,W]
    arrays in, a list of normalised arrays out ('none': float32, global min/max; 'experimental': float64, the 0.5 / 99.5
    percentiles of the 5-tap temporally smoothed clip; anything else: returned unchanged).  The work is the same tensor
    code that runs frame-parallel (process_predictions_sharded) on the whole clip as one local shard -- on the GPU when
    one is visible; nothing here is specific to a device."""
    if smoothening not in ('none', 'experimental') or len(predictions) == 0:
        return predictions
    import torch
    stack = torch.from_numpy(np.stack([np.asarray(p, dtype=np.float32) for p in predictions]))
    if torch.cuda.is_available():
        stack = stack.cuda()
    out = process_predictions_sharded(stack, smoothening, group=_LOCAL).cpu().numpy()
    return [out[i] for i in range(out.shape[0])]


_LOCAL = object()          # group sentinel: treat `local` as the whole clip even inside an initialised process group


def _is_multi(group):
    """True when `group` names more than one rank.  The _LOCAL sentinel is never collective: callers that work on data one
    rank holds completely (process_predicitons, the funnel's per-image 'Outliers' clipping, Boost's rank-0 post-processing)
    must not issue collectives other ranks do not take part in."""
    if group is _LOCAL:
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


# ---- sharded over ranks -------------------------------------------------------------------------------------------------------
def _f32_key(t):
    """Order-preserving map float32 -> int64 (sign-magnitude bit pattern -> two's-complement order)."""
    import torch
    i = t.contiguous().view(torch.int32).to(torch.int64)
    return torch.where(i < 0, -(i & 0x7fffffff) - 1, i)


def _key_to_f32(k):
    k = int(k)
    bits = k if k >= 0 else ((-(k + 1)) | 0x80000000)
    return float(np.array([bits], dtype=np.uint32).view(np.float32)[0])


def _kth_smallest(keys, k, group):
    """Exact k-th smallest (0-based) of the union of every rank's `keys` (int64 tensor): bisection on the value with one
    all-reduce of a count per step (33 steps for the float32 key range)."""
    import torch
    import torch.distributed as dist
    multi = _is_multi(group)
    lo, hi = -(1 << 31) - 1, (1 << 31)
    while hi - lo > 1:                                      # invariant: count(keys <= lo) <= k < count(keys <= hi)
        mid = (lo + hi) // 2
        c = (keys <= mid).sum().to(torch.int64).reshape(1)
        if multi:
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        if int(c.item()) > k:
            hi = mid
        else:
            lo = mid
    return hi


def _global_percentiles(local, qs, group):
    """np.percentile(all values of all ranks, qs) with numpy's default linear interpolation, float32 data.
    group=_LOCAL: the values of THIS rank only, no collective (even inside an initialised multi-rank process)."""
    import torch
    import torch.distributed as dist
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    if _is_multi(group):
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    n = int(n.item())
    keys = _f32_key(local.reshape(-1).float())
    out = []
    for q in qs:
        pos = (n - 1) * (q / 100.0)
        i0 = int(np.floor(pos))
        frac = pos - i0
        v0 = _key_to_f32(_kth_smallest(keys, i0, group))
        v1 = _key_to_f32(_kth_smallest(keys, min(i0 + 1, n - 1), group)) if frac > 0 else v0
        # numpy (>= 1.22, method='linear'; lib/_function_base_impl._lerp): the difference of the two order statistics is
        # taken in the dtype of the data (float32), the product with the float64 weight and the sum in float64, and from the
        # right end when the weight is >= 0.5
        d = np.float64(np.float32(v1) - np.float32(v0))
        v0, v1 = np.float64(v0), np.float64(v1)
        out.append(float(v1 - d * (1 - frac)) if frac >= 0.5 else float(v0 + d * frac))
    return out


def process_predictions_sharded(local, smoothening='none', group=None):
    """The normalisation of process_predicitons for frames sharded in CONTIGUOUS blocks over the ranks.
    local: float32 tensor [n_local, H, W] (this rank's frames, any device).  Returns the normalised tensor (float64 for
    'experimental', like numpy's promotion with the float64 percentiles; float32 for 'none')."""
    import torch
    import torch.distributed as dist
    multi = _is_multi(group)               # the _LOCAL sentinel is handed on as it is: the percentile helpers honour it too
    if smoothening == 'none':
        mm = torch.stack((local.min(), -local.max())) if local.numel() else torch.tensor([float('inf')] * 2, device=local.device)
        if multi:
            dist.all_reduce(mm, op=dist.ReduceOp.MIN, group=group)          # min and (negated) max in one collective
        mn, mx = mm[0], -mm[1]
        return (local - mn) / (mx - mn)
    if smoothening != 'experimental':
        return local
    rank = dist.get_rank(group) if multi else 0
    world = dist.get_world_size(group) if multi else 1
    n_local = local.shape[0]
    # halo: the two frames before and after this rank's block (the global first / last frame replicated: clip() of the
    # reference).  Every rank publishes its first and last (up to) two frames -- zero placeholders where it holds fewer,
    # so the collectives have one fixed shape even for empty shards -- and walks outwards over its neighbours until it
    # has two frames on each side (a neighbour may hold a single frame, or none).
    hw = tuple(local.shape[1:])
    if multi:
        counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=local.device), group=group)
        counts = [int(c.item()) for c in counts]
        k = min(n_local, 2)
        head = torch.zeros((2,) + hw, dtype=local.dtype, device=local.device)
        tail = torch.zeros((2,) + hw, dtype=local.dtype, device=local.device)
        if k:
            head[:k] = local[:k]                              # left aligned
            tail[2 - k:] = local[n_local - k:]                # right aligned
        heads = [torch.empty_like(head) for _ in range(world)]
        tails = [torch.empty_like(tail) for _ in range(world)]
        dist.all_gather(heads, head, group=group)
        dist.all_gather(tails, tail, group=group)
        before, after = [], []
        for r in range(rank - 1, -1, -1):
            c = min(counts[r], 2)
            before = [tails[r][2 - c + i] for i in range(c)] + before
            if len(before) >= 2:
                break
        for r in range(rank + 1, world):
            c = min(counts[r], 2)
            after = after + [heads[r][i] for i in range(c)]
            if len(after) >= 2:
                break
    else:
        before, after = [], []
    if n_local == 0:                                          # nothing to smooth here; still part of every collective below
        left = right = local.new_zeros((2,) + hw)
    else:
        before, after = before[-2:], after[:2]
        first = before[0] if before else local[0]             # exhausted the ranks on that side: the global first frame
        last = after[-1] if after else local[-1]
        left = torch.stack([first] * (2 - len(before)) + before)
        right = torch.stack(after + [last] * (2 - len(after)))
    ext = torch.cat((left, local, right), dim=0)
    processed = torch.zeros_like(local)
    for u, mul in enumerate([0.10, 0.20, 0.40, 0.20, 0.10]):                # same order of accumulation as the reference
        processed += mul * ext[u:u + n_local]
    a, b = _global_percentiles(processed, [0.5, 99.5], group)
    return (local.double() - a) / (b - a)


def gen_frames_sharded(frames_u8, predict_batch, inp, smoothening='none', group=None, batch=8, dst=0, invert=False):
    """Two-pass video pipeline on decoded frames, frame parallel.
    frames_u8: uint8 tensor [F, H, W, 3] (every rank may hold the full clip or just index its block); predict_batch:
    callable uint8 [b,H,W,3] -> float32 [b,H,W] raw prediction (e.g. DepthAnythingV2.infer_batch) on the rank's device.
    invert: True for the models whose raw output is near-is-dark (ids 0, 7-9, 10: depthmap_generation.py:402) -- the
    first pass of the reference yields 'depth_prediction' AFTER ``out *= -1`` (core.py:192-195), so the normalisation sees
    the negated predictions.  Returns on rank `dst` a dict {mode: uint8 [F, ...]} for the requested stereo modes plus
    'depth' (uint16 [F,H,W]); None elsewhere."""
    import torch
    import torch.distributed as dist
    from . import _native, multigpu
    from .stereoimage_generation import create_stereoimages_batch
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if multi else 0
    world = dist.get_world_size(group) if multi else 1
    n = frames_u8.shape[0]
    s, e = multigpu.my_shard(n, rank, world)
    dev = torch.device('cuda', torch.cuda.current_device())
    mine = frames_u8[s:e].to(dev)
    preds = [predict_batch(mine[i:i + batch]) for i in range(0, e - s, batch)]                    # pass 1 (:139-150)
    preds = torch.cat(preds) if preds else torch.empty((0,) + tuple(frames_u8.shape[1:3]), device=dev)
    if invert:
        preds = preds * -1                                                                        # core.py:192-195
    norm = process_predictions_sharded(preds, smoothening, group)                                 # :151
    # second pass: the normalised frames re-enter the funnel as custom depth maps, np.asarray(dp, dtype='float') = float64
    # (core.py:170-174), and are quantised in float64 (:211) -- also in 'none' mode, whose frames are float32
    if e > s:
        d16 = _native.convert_to_i16(norm.double().contiguous())
    else:
        d16 = torch.empty((0,) + tuple(frames_u8.shape[1:3]), dtype=torch.uint16, device=dev)
    modes = list(inp.get('stereo_modes', ['left-right']))
    outs = {'depth': d16}
    if inp.get('gen_stereo', True) and e > s:                                                     # pass 2 (:160)
        res = create_stereoimages_batch(mine, d16, inp.get('stereo_divergence', 2.5), inp.get('stereo_separation', 0.0), modes,
                                        inp.get('stereo_balance', 0.0), inp.get('stereo_offset_exponent', 1.0),
                                        inp.get('stereo_fill_algo', 'polylines_sharp'))
        outs.update(dict(zip(modes, res)))
    gathered = {}
    for k, v in outs.items():                                                                     # ONE gather per output
        if v.dtype == torch.uint16:                          # collectives have no uint16: same bits as int16
            g = multigpu.gather_units(v.view(torch.int16), n, group=group, dst=dst)
            gathered[k] = None if g is None else g.view(torch.uint16)
        else:
            gathered[k] = multigpu.gather_units(v, n, group=group, dst=dst)
    return gathered if rank == dst else None
