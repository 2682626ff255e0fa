"""Deterministic synthetic weights for model-parity tests: every parameter is a function of its NAME and shape only,
so the reference's module and ours (same checkpoint key names) get bit-identical weights without shipping a checkpoint.
Pure torch; used by make_golden_models.py (which imports the reference) and by the tests (which do not)."""
import math
import zlib

import torch


def fill_state_dict(sd):
    out = {}
    for name in sorted(sd.keys()):
        p = sd[name]
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
        r = torch.randn(p.shape, generator=g, dtype=torch.float32)
        leaf = name.split('.')[-1]
        if not p.dtype.is_floating_point:
            out[name] = p.clone()
            continue
        if leaf in ('gamma', 'gamma_1', 'gamma_2'):
            v = 1.0 + 0.1 * r
        elif leaf == 'weight' and p.dim() == 1:                 # LayerNorm / BatchNorm scale
            v = 1.0 + 0.1 * r
        elif leaf == 'weight':
            fan_in = p[0].numel() if p.dim() > 1 else p.numel()
            v = r / math.sqrt(max(fan_in, 1))
        elif leaf == 'bias':
            v = 0.1 * r
        elif leaf in ('running_mean',):
            v = 0.1 * r
        elif leaf in ('running_var',):
            v = 1.0 + 0.1 * r.abs()
        else:                                                   # cls_token, pos_embed, mask_token, bias tables ...
            v = 0.02 * r
        out[name] = v.to(p.dtype)
    return out


def synthetic_image(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def fill_state_dict_beit(sd):
    """fill_state_dict + BEiT-specific scales: gamma_1/gamma_2 near 1 would let a dozen random blocks blow the activations
    up, and a bias table of std 0.02 would not shape the softmax visibly."""
    out = fill_state_dict(sd)
    for k in list(out.keys()):
        if k.endswith("gamma_1") or k.endswith("gamma_2"):
            out[k] = out[k] * 0.3
        if k.endswith("relative_position_bias_table"):
            out[k] = out[k] * 25.0
    return out


def fill_state_dict_zoe(sd):
    """fill_state_dict_beit for ZoeDepth: the float buffer K_minus_1 of the log-binomial layer is a constant, not a weight."""
    out = fill_state_dict_beit(sd)
    for k in sd:
        if k.endswith("K_minus_1"):
            out[k] = sd[k].clone()
    return out


def boost_integral_image(seed=5, H=700, W=1000):
    """Deterministic gradient field + its integral image (cv2.integral layout) for the Boost patch-selection cases."""
    import numpy as np
    rng = np.random.default_rng(seed)
    grad = rng.random((H, W)) * (rng.random((H, W)) > 0.6)
    grad[200:420, 300:700] += 2.0                         # a dense-gradient region that the selection should grow around
    integ = np.zeros((H + 1, W + 1))
    integ[1:, 1:] = grad.cumsum(0).cumsum(1)
    return grad, integ
