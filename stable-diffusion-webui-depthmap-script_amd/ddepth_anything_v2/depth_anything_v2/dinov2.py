"""DINOv2 encoder of Depth-Anything-V2 on the shared MI355X transformer machinery (src/vit_mi355x.py).

Reference: ddepth_anything_v2/depth_anything_v2/dinov2.py (DinoVisionTransformer :37-321, model zoo :339-415),
dinov2_layers/{attention.py:49-62, block.py:82-107, layer_scale.py, mlp.py, patch_embed.py:69-83}.
Parameter names are the reference's, so its checkpoints load with ``load_state_dict(strict=True)``.
Inference only (no drop-path, no masks, no register tokens: the zoo never enables them, dinov2.py:404-414).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm


class _Attention(nn.Module):            # parameter container: attn.qkv / attn.proj (attention.py:44-47)
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _LayerScale(nn.Module):           # ls1.gamma / ls2.gamma (layer_scale.py:16-27)
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Block(vm.EncoderBlock):
    """x += ls1(attn(norm1(x))); x += ls2(mlp(norm2(x)))  (block.py:82-107, inference branch)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, init_values=1.0):
        super().__init__(dim, num_heads, mlp_ratio)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim)
        self.ls1 = _LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = vm.Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LayerScale(dim, init_values)

    def qkv_weights(self):
        c = self.dim
        w, b = self.attn.qkv.weight, self.attn.qkv.bias
        return w[:2 * c], b[:2 * c], w[2 * c:], b[2 * c:]

    def proj(self, o, b_v=None):
        return vm.linear(o, self.attn.proj.weight, vm.folded_proj_bias(self.attn.proj, b_v))

    def gammas(self):
        return self.ls1.gamma, self.ls2.gamma


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):               # B C H W -> B HW C   (patch_embed.py:69-83)
        if vm.patch_embed_hip_ok(self.proj, x):
            return vm.patch_embed_tokens(self.proj, x)
        return self.proj(x).flatten(2).transpose(1, 2)


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 init_values=1.0, interpolate_offset=0.1):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.patch_size = patch_size
        self.n_blocks = depth
        self.num_heads = num_heads
        self.interpolate_offset = interpolate_offset
        self.patch_embed = PatchEmbed(patch_size, 3, embed_dim)
        num_patches = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))    # present in the checkpoints, unused at inference
        self._pos_cache = {}
        self.init_weights()

    def init_weights(self):             # dinov2.py:176-181,324-329
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def interpolate_pos_encoding(self, npatch, w, h, dtype):
        """dinov2.py:183-210 (bicubic, +0.1 offset work-around); cached per input size: it only depends on (w, h)."""
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed.to(dtype)
        key = (w, h, dtype, self.pos_embed.device, self.pos_embed._version)
        hit = self._pos_cache.get(key)
        if hit is not None:
            return hit
        pos_embed = self.pos_embed.float()
        class_pos_embed = pos_embed[:, 0]
        patch_pos_embed = pos_embed[:, 1:]
        dim = pos_embed.shape[-1]
        w0 = w // self.patch_size + self.interpolate_offset
        h0 = h // self.patch_size + self.interpolate_offset
        sqrt_N = math.sqrt(N)
        sx, sy = float(w0) / sqrt_N, float(h0) / sqrt_N
        patch_pos_embed = F.interpolate(patch_pos_embed.reshape(1, int(sqrt_N), int(sqrt_N), dim).permute(0, 3, 1, 2),
                                        scale_factor=(sx, sy), mode="bicubic", antialias=False)
        assert int(w0) == patch_pos_embed.shape[-2] and int(h0) == patch_pos_embed.shape[-1]
        patch_pos_embed = patch_pos_embed.permute(0, 2, 3, 1).reshape(1, -1, dim)
        out = torch.cat((class_pos_embed.unsqueeze(0), patch_pos_embed), dim=1).to(dtype)
        if not torch.is_grad_enabled():
            vm.cache_store(self._pos_cache, key, out)
        return out

    def prepare_tokens(self, x):
        B, nc, w, h = x.shape           # the reference names them (w, h) in this order (dinov2.py:213)
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x), dim=1)
        return x + self.interpolate_pos_encoding(x.shape[1] - 1, w, h, x.dtype)

    def get_intermediate_layers(self, x, n, return_class_token=False, norm=True):
        """dinov2.py:264-307 (not-chunked branch): outputs of the blocks listed in `n`, final LayerNorm applied, class
        token split off.  The sequence is padded once and the pad rows are dropped when a tap is taken."""
        tokens = self.prepare_tokens(x)
        n_valid = tokens.shape[1]
        tokens = vm.pad_tokens(tokens, vm.pad_len(n_valid, tokens.shape[0]))
        take = sorted(set(range(len(self.blocks) - n, len(self.blocks))) if isinstance(n, int) else set(n))
        _, taps = vm.run_blocks(self.blocks, tokens, n_valid, None, set(take))
        outs = [taps[i] for i in take]
        if norm:
            outs = [self.norm(o) for o in outs]
        cls = [o[:, 0] for o in outs]
        outs = [o[:, 1:] for o in outs]
        return tuple(zip(outs, cls)) if return_class_token else tuple(outs)


_ZOO = {"vits": (384, 12, 6), "vitb": (768, 12, 12), "vitl": (1024, 24, 16)}


def DINOv2(model_name):
    """dinov2.py:396-415 (vitg uses a SwiGLU FFN and is not offered by the reference UI: model ids 12-14 only)."""
    if model_name not in _ZOO:
        raise NotImplementedError(f"DINOv2 '{model_name}' is not built (reference model ids 12-14 are vits/vitb/vitl)")
    dim, depth, heads = _ZOO[model_name]
    return DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=dim, depth=depth, num_heads=heads,
                                 init_values=1.0, interpolate_offset=0.1)
