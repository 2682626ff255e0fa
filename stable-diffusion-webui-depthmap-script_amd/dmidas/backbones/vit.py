"""ViT-L/16 and the ViT-B/16 + ResNet-50 hybrid backbones of MiDaS 3.0 DPT (dpt_large_384 / dpt_hybrid_384) on the shared
MI355X transformer machinery.

Reference: dmidas/backbones/vit.py (_resize_pos_embed :16-30, forward_flex :33-72, _make_vit_b_rn50_backbone :120-205,
_make_pretrained_* :78-110,208-221) and dmidas/backbones/utils.py (read-out / act_postprocess, :28-39,144-249).  The
transformer body and the ResNetV2 stem live in the un-vendored dependency timm~=0.9.2 (timm/models/vision_transformer.py,
vision_transformer_hybrid.py, resnetv2.py, timm/layers/{std_conv,norm_act,padding,pool2d_same}.py) and are restated
from their published source.  Parity: the reference's own dmidas code executed on a stand-in for the timm classes
(tests/golden/fake_timm.py) matches to 1e-4; timm's own part is unpinned (not installable here).

MI355X-first differences: padded token sequence + fused attention (src/vit_mi355x.py); the weight standardisation of
the stem's StdConv layers is computed once per weight version instead of in every forward; the resized position
embedding is cached per grid; hooks/global activations dict replaced by returned taps.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm
from .beit import ProjectReadout, _Skip


# ---- ResNetV2 stem (timm/models/resnetv2.py, preact=False, stem_type='same', StdConv2dSame eps=1e-8, GroupNorm 32) -------
def _pad_same(x, k, s, value=0.0):
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + (k - 1) + 1 - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + (k - 1) + 1 - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """Weight-standardised convolution with TF 'SAME' padding (timm/layers/std_conv.py).  The standardised weight only
    depends on the parameter: cached per weight version (timm recomputes it in every forward)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, eps=1e-8):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=0, bias=False)
        self.eps = eps
        self._std_cache = None

    def std_weight(self):
        w = self.weight
        key = (w._version, w.dtype, w.device, w.data_ptr())
        if self._std_cache is not None and self._std_cache[0] == key:
            return self._std_cache[1]
        wf = w.float().reshape(w.shape[0], -1)
        mean = wf.mean(dim=1, keepdim=True)
        var = wf.var(dim=1, keepdim=True, unbiased=False)
        ws = ((wf - mean) / torch.sqrt(var + self.eps)).reshape_as(w).to(w.dtype)
        if not torch.is_grad_enabled():
            if self._std_cache is not None:
                vm.cache_evicted()
            self._std_cache = (key, ws)
        return ws

    def forward(self, x):
        k, st = self.kernel_size[0], self.stride[0]
        ih, iw = x.shape[-2:]
        ph = max((math.ceil(ih / st) - 1) * st + (k - 1) + 1 - ih, 0)
        pw = max((math.ceil(iw / st) - 1) * st + (k - 1) + 1 - iw, 0)
        if ph % 2 == 0 and pw % 2 == 0:
            # TF 'SAME' padding that happens to be symmetric (every stride-1 3x3 of the stem): the convolution's own zero padding -- the
            # same values without the fill + copy of a padded tensor in front of every such layer
            return F.conv2d(x, self.std_weight(), None, self.stride, (ph // 2, pw // 2))
        x = _pad_same(x, k, st)
        return F.conv2d(x, self.std_weight(), None, self.stride, 0)


GROUPNORM_HIP = os.environ.get("DS_GROUPNORM", "1") != "0"        # A/B switch: 0 = torch's F.group_norm + separate ReLU / add


class GroupNormAct(nn.GroupNorm):
    def __init__(self, ch, apply_act=True):
        super().__init__(32, ch, eps=1e-5)
        self.apply_act = apply_act

    def forward(self, x, res=None, relu_after_res=False):
        """GroupNorm [+ ReLU]; with `res`: relu(group_norm(x) + res), the tail of a bottleneck (norm3, the add and the ReLU in one
        pass).  Half-precision CUDA activations take the in-tree two-launch kernel (csrc/ds_encoder_ops.hip: ds_group_norm_nchw --
        torch spends three launches, 15 us of them on 32 workgroups, plus a ReLU and an add per norm at batch 1)."""
        if GROUPNORM_HIP and vm.half_on_gpu(x) and not vm.STOCK[0] and not (torch.is_grad_enabled() and x.requires_grad):
            from src import _native
            if not x.is_contiguous():
                x = x.contiguous()
            if res is not None and not res.is_contiguous():
                res = res.contiguous()
            if _native.group_norm_supported(x, self.num_groups):
                return _native.group_norm(x, self.num_groups, self.weight, self.bias, self.eps,
                                          relu=self.apply_act or (res is not None and relu_after_res), res=res)
        x = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        if res is not None:
            x = x + res
            return F.relu(x) if relu_after_res else x
        return F.relu(x) if self.apply_act else x


class _Downsample(nn.Module):
    def __init__(self, in_ch, out_ch, stride):
        super().__init__()
        self.conv = StdConv2dSame(in_ch, out_ch, 1, stride)
        self.norm = GroupNormAct(out_ch, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x))


class _Bottleneck(nn.Module):
    def __init__(self, in_ch, out_ch, stride, downsample):
        super().__init__()
        mid = out_ch // 4
        if downsample:
            self.downsample = _Downsample(in_ch, out_ch, stride)
        else:
            self.downsample = None
        self.conv1 = StdConv2dSame(in_ch, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, out_ch, 1)
        self.norm3 = GroupNormAct(out_ch, apply_act=False)

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        x = self.norm1(self.conv1(x))
        x = self.norm2(self.conv2(x))
        return self.norm3(self.conv3(x), res=shortcut, relu_after_res=True)       # relu(norm3(conv3(x)) + shortcut)


class _Stage(nn.Module):
    def __init__(self, in_ch, out_ch, stride, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[_Bottleneck(in_ch if i == 0 else out_ch, out_ch, stride if i == 0 else 1, i == 0)
                                      for i in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class ResNetV2Stem(nn.Module):
    """patch_embed.backbone of vit_base_resnet50_384: ResNetV2(layers=(3, 4, 9)), output stride 16, 1024 channels."""

    def __init__(self):
        super().__init__()
        stem = nn.Module()
        stem.conv = StdConv2dSame(3, 64, 7, 2)
        stem.norm = GroupNormAct(64)
        self.stem = stem
        self.stages = nn.ModuleList([_Stage(64, 256, 1, 3), _Stage(256, 512, 2, 4), _Stage(512, 1024, 2, 9)])

    def forward_stages(self, x):
        x = self.stem.norm(self.stem.conv(x))
        x = F.max_pool2d(_pad_same(x, 3, 2, value=float('-inf')), 3, 2, 0)       # MaxPool2dSame
        outs = []
        for st in self.stages:
            x = st(x)
            outs.append(x)
        return outs


class _HybridEmbed(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.backbone = ResNetV2Stem()
        self.proj = nn.Conv2d(1024, embed_dim, kernel_size=1, stride=1)


class _PatchEmbed(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=16, stride=16)


# ---- ViT body ------------------------------------------------------------------------------------------------------------
class _Attention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Block(vm.EncoderBlock):
    """timm VisionTransformer Block without LayerScale: x += attn(norm1(x)); x += mlp(norm2(x))."""

    def __init__(self, dim, num_heads):
        super().__init__(dim, num_heads, 4.0)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = vm.Mlp(dim, dim * 4)

    def qkv_weights(self):
        c = self.dim
        w, b = self.attn.qkv.weight, self.attn.qkv.bias
        return w[:2 * c], b[:2 * c], w[2 * c:], b[2 * c:]

    def proj(self, o, b_v=None):
        return vm.linear(o, self.attn.proj.weight, vm.folded_proj_bias(self.attn.proj, b_v))

    def gammas(self):
        return None, None


class VisionTransformer(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, hybrid, img_size=384, num_classes=1000):
        super().__init__()
        self.patch_size = [16, 16]
        self.start_index = 1
        self.hybrid = hybrid
        self.patch_embed = _HybridEmbed(embed_dim) if hybrid else _PatchEmbed(embed_dim)
        n = (img_size // 16) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)          # present in the checkpoints; unused
        self._pos_cache = {}
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def resized_pos_embed(self, gs_h, gs_w, dtype):
        """vit.py:16-30 (bilinear, align_corners False), cached per grid."""
        key = (gs_h, gs_w, dtype, self.pos_embed.device, self.pos_embed._version)
        hit = self._pos_cache.get(key)
        if hit is not None:
            return hit
        posemb = self.pos_embed
        tok, grid = posemb[:, :1], posemb[0, 1:]
        gs_old = int(math.sqrt(len(grid)))
        grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid.float(), size=(gs_h, gs_w), mode="bilinear").to(posemb.dtype)
        grid = grid.permute(0, 2, 3, 1).reshape(1, gs_h * gs_w, -1)
        out = torch.cat([tok, grid], dim=1).to(dtype)
        if not torch.is_grad_enabled():
            vm.cache_store(self._pos_cache, key, out)
        return out

    def forward_taps(self, x, hooks, n_stage_taps):
        """forward_flex (vit.py:33-72) + the forward hooks (:135-141): returns the stem feature maps of the first
        `n_stage_taps` stages and the outputs of transformer blocks hooks[n_stage_taps:]."""
        b, c, h, w = x.shape
        grid = (h // 16, w // 16)
        stage_outs = []
        if self.hybrid:
            feats = self.patch_embed.backbone.forward_stages(x)
            stage_outs = feats[:n_stage_taps]
            t = self.patch_embed.proj(feats[-1]).flatten(2).transpose(1, 2)
        elif vm.patch_embed_hip_ok(self.patch_embed.proj, x):
            t = vm.patch_embed_tokens(self.patch_embed.proj, x)
        else:
            t = self.patch_embed.proj(x).flatten(2).transpose(1, 2)
        t = torch.cat((self.cls_token.expand(b, -1, -1).to(t.dtype), t), dim=1)
        t = t + self.resized_pos_embed(grid[0], grid[1], t.dtype)
        n_valid = t.shape[1]
        t = vm.pad_tokens(t, vm.pad_len(n_valid, t.shape[0]))
        block_hooks = list(hooks[n_stage_taps:])
        _, taps = vm.run_blocks(self.blocks, t, n_valid, grid, set(block_hooks), padded_taps=True)
        return stage_outs, [taps[i] for i in block_hooks], grid, n_valid


def _token_postprocess(vit_features, out_features, tail):
    return nn.Sequential(ProjectReadout(vit_features), _Skip(), _Skip(), nn.Conv2d(vit_features, out_features, 1), *tail)


class VitBackbone(nn.Module):
    """`pretrained` of the reference for vitl16_384 / vitb_rn50_384 (vit.py:78-110, 120-221), readout 'project'."""

    def __init__(self, model, features, vit_features, hooks, number_stages):
        super().__init__()
        self.model = model
        self.hooks = list(hooks)
        self.number_stages = number_stages               # stem stages tapped directly (hybrid: 2, plain ViT: 0)
        f = features
        posts = []
        for s in range(4):
            if s < number_stages:
                posts.append(nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity()))
                continue
            if number_stages == 0:                        # make_backbone_default layout (utils.py:165-243)
                tail = [[nn.ConvTranspose2d(f[0], f[0], 4, 4, 0)], [nn.ConvTranspose2d(f[1], f[1], 2, 2, 0)], [],
                        [nn.Conv2d(f[3], f[3], 3, 2, 1)]][s]
            else:                                         # _make_vit_b_rn50_backbone (vit.py:155-187)
                tail = [] if s == number_stages else [nn.Conv2d(f[3], f[3], 3, 2, 1)]
            posts.append(_token_postprocess(vit_features, f[s], tail))
        self.act_postprocess1, self.act_postprocess2, self.act_postprocess3, self.act_postprocess4 = posts

    def forward(self, x):
        stage_outs, taps, grid, n_valid = self.model.forward_taps(x, self.hooks, self.number_stages)
        posts = (self.act_postprocess1, self.act_postprocess2, self.act_postprocess3, self.act_postprocess4)
        outs = list(stage_outs)
        for tap, post in zip(taps, posts[self.number_stages:]):
            y = post[0].forward_padded(tap, n_valid)           # ProjectReadout on the padded block output
            y = y.reshape(y.shape[0], grid[0], grid[1], y.shape[2]).permute(0, 3, 1, 2)      # NHWC view, no copy
            for layer in list(post)[3:]:
                y = vm.conv_module(layer, y)
            outs.append(y)
        return outs


def make_vit(name, hooks):
    if name == "vitl16_384":
        model = VisionTransformer(1024, 24, 16, hybrid=False)
        features = [256, 512, 1024, 1024]
        return VitBackbone(model, features, 1024, hooks, 0), features
    if name == "vitb_rn50_384":
        model = VisionTransformer(768, 12, 12, hybrid=True)
        features = [256, 512, 768, 768]
        return VitBackbone(model, features, 768, hooks, 2), features
    raise KeyError(name)
