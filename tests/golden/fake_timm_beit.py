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


_CFG = {"beit_large_patch16_512": (512, 1024, 24, 16, 1e-5), "beit_large_patch16_384": (384, 1024, 24, 16, 1e-5),
        "beit_base_patch16_384": (384, 768, 12, 12, 0.1)}


def create_model(name, pretrained=False, **kw):
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
