"""TEST INFRASTRUCTURE -- a stand-in for the un-vendored dependency timm~=0.9.2 (requirements.txt:8), just enough of
`timm.create_model("beit_*_patch16_*")` for the REFERENCE's own dmidas code to run in a container without timm:
parameter containers with timm's attribute names (timm/models/beit.py: Attention, Block, Beit; timm/layers: Mlp,
PatchEmbed).  The reference replaces every forward of these classes with its own functions
(dmidas/backbones/beit.py:137-156), so only the containers, Mlp.forward, LayerNorm and the patch conv come from here.
"""
import sys
import types

import torch
import torch.nn as nn


def gen_relative_position_index(window_size):
    num_relative_distance = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
    window_area = window_size[0] * window_size[1]
    coords = torch.stack(torch.meshgrid([torch.arange(window_size[0]), torch.arange(window_size[1])], indexing='ij'))
    coords_flatten = torch.flatten(coords, 1)
    relative_coords = coords_flatten[:, :, None] - coords_flatten[:, None, :]
    relative_coords = relative_coords.permute(1, 2, 0).contiguous()
    relative_coords[:, :, 0] += window_size[0] - 1
    relative_coords[:, :, 1] += window_size[1] - 1
    relative_coords[:, :, 0] *= 2 * window_size[1] - 1
    relative_position_index = torch.zeros(size=(window_area + 1,) * 2, dtype=relative_coords.dtype)
    relative_position_index[1:, 1:] = relative_coords.sum(-1)
    relative_position_index[0, 0:] = num_relative_distance - 3
    relative_position_index[0:, 0] = num_relative_distance - 2
    relative_position_index[0, 0] = num_relative_distance - 1
    return relative_position_index


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, window_size):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.register_buffer('k_bias', torch.zeros(dim), persistent=False)
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.window_size = window_size
        self.num_relative_distance = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        self.attn_drop = nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Identity()


class Block(nn.Module):
    def __init__(self, dim, num_heads, window_size, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, window_size)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * 4)
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))


class PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.flatten = True
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16)
        self.norm = nn.Identity()


class Beit(nn.Module):
    def __init__(self, img_size, dim, depth, heads, init_values):
        super().__init__()
        window = (img_size // 16, img_size // 16)
        self.patch_embed = PatchEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = None
        self.pos_drop = nn.Identity()
        self.rel_pos_bias = None
        self.grad_checkpointing = False
        self.blocks = nn.ModuleList([Block(dim, heads, window, init_values) for _ in range(depth)])
        self.norm = nn.Identity()
        self.fc_norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, 1000)


# ---- timm VisionTransformer / hybrid (vision_transformer.py, vision_transformer_hybrid.py, resnetv2.py) -------------------
import math
import torch.nn.functional as F


class VitAttention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class VitBlock(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = VitAttention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * 4)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


def _same_pad(x, k, s, value=0.0):
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    return F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)


class StdConv2dSame(nn.Conv2d):
    def __init__(self, i, o, k, stride=1):
        super().__init__(i, o, k, stride=stride, padding=0, bias=False)

    def forward(self, x):
        w = F.batch_norm(self.weight.reshape(1, self.out_channels, -1), None, None, training=True, momentum=0., eps=1e-8)
        return F.conv2d(_same_pad(x, self.kernel_size[0], self.stride[0]), w.reshape_as(self.weight), None, self.stride)


class GNAct(nn.GroupNorm):
    def __init__(self, ch, act=True):
        super().__init__(32, ch, eps=1e-5)
        self.act = act

    def forward(self, x):
        x = super().forward(x)
        return torch.relu(x) if self.act else x


class RnDown(nn.Module):
    def __init__(self, i, o, s):
        super().__init__()
        self.conv = StdConv2dSame(i, o, 1, s)
        self.norm = GNAct(o, False)

    def forward(self, x):
        return self.norm(self.conv(x))


class RnBottleneck(nn.Module):
    def __init__(self, i, o, s, first):
        super().__init__()
        m = o // 4
        if first:
            self.downsample = RnDown(i, o, s)
        self.first = first
        self.conv1, self.norm1 = StdConv2dSame(i, m, 1), GNAct(m)
        self.conv2, self.norm2 = StdConv2dSame(m, m, 3, s), GNAct(m)
        self.conv3, self.norm3 = StdConv2dSame(m, o, 1), GNAct(o, False)

    def forward(self, x):
        sc = self.downsample(x) if self.first else x
        y = self.norm3(self.conv3(self.norm2(self.conv2(self.norm1(self.conv1(x))))))
        return torch.relu(y + sc)


class RnStage(nn.Module):
    def __init__(self, i, o, s, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[RnBottleneck(i if d == 0 else o, o, s if d == 0 else 1, d == 0) for d in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class RnStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = StdConv2dSame(3, 64, 7, 2)
        self.norm = GNAct(64)

    def forward(self, x):
        x = self.norm(self.conv(x))
        return F.max_pool2d(_same_pad(x, 3, 2, float('-inf')), 3, 2)


class ResNetV2(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = RnStem()
        self.stages = nn.ModuleList([RnStage(64, 256, 1, 3), RnStage(256, 512, 2, 4), RnStage(512, 1024, 2, 9)])

    def forward(self, x):
        x = self.stem(x)
        for s in self.stages:
            x = s(x)
        return x


class HybridEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.backbone = ResNetV2()
        self.proj = nn.Conv2d(1024, dim, 1, 1)


class PlainEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, 16, 16)


class VisionTransformer(nn.Module):
    def __init__(self, dim, depth, heads, hybrid):
        super().__init__()
        self.patch_embed = HybridEmbed(dim) if hybrid else PlainEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 24 * 24 + 1, dim))
        self.no_embed_class = False
        self.dist_token = None
        self.pos_drop = nn.Identity()
        self.blocks = nn.Sequential(*[VitBlock(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, 1000)


_VIT = {"vit_large_patch16_384": (1024, 24, 16, False), "vit_base_resnet50_384": (768, 12, 12, True)}

_CFG = {"beit_large_patch16_512": (512, 1024, 24, 16, 1e-5), "beit_large_patch16_384": (384, 1024, 24, 16, 1e-5),
        "beit_base_patch16_384": (384, 768, 12, 12, 0.1)}


def create_model(name, pretrained=False, **kw):
    if name in _VIT:
        return VisionTransformer(*_VIT[name])
    return Beit(*_CFG[name])


def install():
    """Register the stand-in as `timm` (+ the submodules dmidas imports at module scope)."""
    from unittest import mock
    timm = types.ModuleType("timm")
    timm.create_model = create_model
    sys.modules["timm"] = timm
    for sub in ("timm.models", "timm.models.layers", "timm.models.registry", "timm.models.vision_transformer",
                "timm.models.helpers", "timm.models.layers.helpers", "timm.layers", "timm.models._registry",
                "timm.models._builder", "timm.models._manipulate", "timm.models.swin_transformer_v2",
                "timm.models.swin_transformer", "timm.models.levit", "timm.models.efficientnet"):
        sys.modules[sub] = mock.MagicMock()
    beit = types.ModuleType("timm.models.beit")
    beit.gen_relative_position_index = gen_relative_position_index
    sys.modules["timm.models.beit"] = beit
