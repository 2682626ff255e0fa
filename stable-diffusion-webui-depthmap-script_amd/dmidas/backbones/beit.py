"""BEiT backbone of MiDaS 3.1 DPT on the shared MI355X transformer machinery.

Reference: dmidas/backbones/beit.py (patch_embed_forward :18-27, _get_rel_pos_bias :29-62, attention_forward :65-91,
block_forward :94-107, beit_forward_features :110-129, _make_pretrained_beit* :159-198) and dmidas/backbones/utils.py
(ProjectReadout :28-39, forward_adapted_unflatten :83-124, make_backbone_default :144-249).  The transformer body itself
lives in the un-vendored dependency timm~=0.9.2 (requirements.txt:8; timm/models/beit.py: Attention, Block, Beit,
gen_relative_position_index); it is restated here from its published source.  Parity: the reference's OWN dmidas code
(all the forwards it monkey-patches into timm's classes, the read-out, the DPT decoder) was executed on a stand-in for
timm's parameter containers (tests/golden/fake_timm.py) and our outputs match it to 1e-4 (tests/test_models_cpu.py);
what remains unpinned is timm's own part: the containers' shapes, Mlp/LayerNorm/patch-conv and
gen_relative_position_index (timm is not installable in the build container).

What is different from the reference on purpose:
  * the relative-position bias of a block depends only on (table, window): it is interpolated + gathered ONCE per
    resolution and cached as a padded [H, Np, Np] operand of ds_attention_fwd, instead of in every block of every
    forward (beit.py:29-62 runs F.interpolate + a 1M-element gather 24 times per image);
  * forward hooks and the module-global `activations` dict (utils.py:60-67,155-160) are replaced by returning the four
    taps from the block loop.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm


def gen_relative_position_index(window_size):
    """timm/models/beit.py gen_relative_position_index (0.9.x): pairwise relative positions inside the window plus three
    extra entries for cls->token, token->cls and cls->cls."""
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


class _Attention(nn.Module):
    """Parameter container with timm's names: qkv.weight (no bias), q_bias, v_bias, relative_position_bias_table,
    proj; k_bias is a constant zero (a non-persistent buffer in timm)."""

    def __init__(self, dim, num_heads, window_size):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.window_size = window_size
        self.num_relative_distance = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        self.proj = nn.Linear(dim, dim)


class Block(vm.EncoderBlock):
    """x += gamma_1 * attn(norm1(x)); x += gamma_2 * mlp(norm2(x))  (beit.py:94-107)."""

    def __init__(self, dim, num_heads, window_size, mlp_ratio=4.0, init_values=1e-5):
        super().__init__(dim, num_heads, mlp_ratio)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads, window_size)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = vm.Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        self._bias_cache = {}

    def qkv_weights(self):
        c = self.dim
        w = self.attn.qkv.weight
        qb = self.attn.q_bias
        key = (qb._version, qb.data_ptr(), qb.dtype)
        hit = getattr(self, "_bqk", None)
        if hit is None or hit[0] != key:
            if hit is not None:
                vm.cache_evicted()              # a live hipGraph may still read the old tensor
            b_qk = torch.cat((qb, torch.zeros_like(qb)))                                 # k_bias == 0 (beit.py:71)
            hit = (key, b_qk)
            if not torch.is_grad_enabled():
                self._bqk = hit
        return w[:2 * c], hit[1], w[2 * c:], self.attn.v_bias

    def proj(self, o, b_v=None):
        return vm.linear(o, self.attn.proj.weight, vm.folded_proj_bias(self.attn.proj, b_v))

    def gammas(self):
        return self.gamma_1, self.gamma_2

    def _resized_table(self, window_size):
        """beit.py:29-54: the bias table bilinearly resized to the run-time window, [(2Wh-1)(2Ww-1) + 3, H]."""
        a = self.attn
        old_h, old_w = 2 * a.window_size[0] - 1, 2 * a.window_size[1] - 1
        new_h, new_w = 2 * window_size[0] - 1, 2 * window_size[1] - 1
        table = a.relative_position_bias_table
        old_sub = table[:a.num_relative_distance - 3].reshape(1, old_w, old_h, -1).permute(0, 3, 1, 2)
        new_sub = F.interpolate(old_sub.float(), size=(int(new_h), int(new_w)), mode="bilinear").to(table.dtype)
        new_sub = new_sub.permute(0, 2, 3, 1).reshape(new_h * new_w, -1)
        return torch.cat([new_sub, table[a.num_relative_distance - 3:]])

    def rel_pos_bias(self, window_size):
        """[H, N, N] (query, key) for N = Wh*Ww + 1: beit.py:29-62 (bilinear resize of the table to the new window, then
        the index gather)."""
        new_table = self._resized_table(window_size)
        index = _relative_position_index(tuple(window_size), new_table.device)
        n = window_size[0] * window_size[1] + 1
        bias = new_table[index.view(-1)].view(n, n, -1)
        return bias.permute(2, 0, 1).contiguous()

    def attention_bias(self, n_pad, grid_hw, dtype, device):
        """Padded [H, Np(query), Np(key)]; cached per (window, dtype) until the table changes.  A dense float32 operand
        above DENSE_BIAS_BYTES_MAX is NOT kept (Boost's whole-image pass of the float32 path reaches 10^4 tokens: 6.5 GB
        per block, 155 GB for the 24 blocks): the block then hands out a _LazyBias that gathers per query tile from the
        resized table, like the reference, which builds the bias transiently in every block (beit.py:29-62)."""
        a = self.attn
        plain = dtype in (torch.float32, torch.float64) or device.type != 'cuda' or vm.STOCK[0]     # attention_reference's dense [H, Np, Np] operand
        if not plain:
            n_pad = (n_pad + 63) // 64 * 64            # the packed operand of the HIP kernel lives on whole 64-key tiles
        key = (tuple(grid_hw), n_pad, dtype, device, a.relative_position_bias_table._version, plain)
        hit = self._bias_cache.get(key)
        if hit is not None:
            return hit
        with torch.no_grad():
            if plain:
                heads = a.relative_position_bias_table.shape[1]
                if heads * n_pad * n_pad * 4 > DENSE_BIAS_BYTES_MAX:
                    if self._bias_cache:
                        self._bias_cache.clear()
                        vm.cache_evicted()
                    return _LazyBias(self._resized_table(tuple(grid_hw)).to(device=device, dtype=dtype), tuple(grid_hw), n_pad)
                bias = self.rel_pos_bias(tuple(grid_hw))                                 # H, N, N (query, key)
                n = bias.shape[-1]
                bt = torch.zeros((bias.shape[0], n_pad, n_pad), dtype=dtype, device=device)
                bt[:, :n, :n] = bias
            else:                                   # operand of the HIP kernel (register order, log2 units)
                from src import _native
                bt = _native.attention_bias_pack(self.rel_pos_bias(tuple(grid_hw)).to(device), n_pad, dtype)
        return vm.cache_store(self._bias_cache, key, bt, entries=2)     # the packed bias is large (H * Np^2): two sizes at most


DENSE_BIAS_BYTES_MAX = 256 << 20
_index_cache = {}


def _relative_position_index(window_size, device):
    """gen_relative_position_index, shared by all blocks (it depends on the window only); one entry is kept."""
    key = (tuple(window_size), str(device))
    hit = _index_cache.get(key)
    if hit is None:
        hit = vm.cache_store(_index_cache, key, gen_relative_position_index(window_size).to(device))
    return hit


class _LazyBias:
    """Relative-position bias too large to keep dense: rows(q0, q1) gathers [H, q1-q0, Np] from the resized table."""

    def __init__(self, table, window_size, n_pad):
        self.table, self.window_size, self.n_pad = table, window_size, n_pad
        self.n = window_size[0] * window_size[1] + 1

    def rows(self, q0, q1):
        index = _relative_position_index(self.window_size, self.table.device)
        out = self.table.new_zeros((self.table.shape[1], q1 - q0, self.n_pad))
        qe = min(q1, self.n)
        if qe > q0:
            out[:, :qe - q0, :self.n] = self.table[index[q0:qe].reshape(-1)].view(qe - q0, self.n, -1).permute(2, 0, 1)
        return out


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):               # beit.py:18-27: any input size
        if vm.patch_embed_hip_ok(self.proj, x):
            return vm.patch_embed_tokens(self.proj, x)
        return self.proj(x).flatten(2).transpose(1, 2)


class Beit(nn.Module):
    """timm Beit with use_abs_pos_emb=False, use_rel_pos_bias=True, global_pool='avg' (the beit_*_patch16_* entry points
    the reference instantiates, beit.py:160,177,188).  `fc_norm` and `head` exist only so that MiDaS checkpoints load
    strictly; the DPT taps are raw block outputs."""

    def __init__(self, img_size, embed_dim, depth, num_heads, init_values, num_classes=1000):
        super().__init__()
        self.patch_size = [16, 16]
        self.embed_dim = embed_dim
        window = (img_size // 16, img_size // 16)
        self.patch_embed = PatchEmbed(16, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, window, 4.0, init_values) for _ in range(depth)])
        self.fc_norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_taps(self, x, hooks, padded=False):
        """beit.py:110-129 + the forward hooks of utils.py:155-158: outputs of blocks `hooks`, unpadded [B, N, C] (padded: the
        padded [B, Np, C] block outputs, for ProjectReadout.forward_padded).  Returns (taps, grid, n_valid)."""
        grid = (x.shape[2] // 16, x.shape[3] // 16)
        t = self.patch_embed(x)
        t = torch.cat((self.cls_token.expand(t.shape[0], -1, -1).to(t.dtype), t), dim=1)
        n_valid = t.shape[1]
        t = vm.pad_tokens(t, vm.pad_len(n_valid, t.shape[0]))
        _, taps = vm.run_blocks(self.blocks, t, n_valid, grid, set(hooks), padded_taps=padded)
        return [taps[i] for i in hooks], grid, n_valid


class ProjectReadout(nn.Module):       # utils.py:28-39
    """cat(tokens[:, 1:], cls expanded) -> Linear(2C -> C) -> GELU, without the concatenation: the linear map splits into
    the token part (a GEMM with K = C on the tap as it is) and the cls part (one vector per image),
        W.[tok ; cls] + b = W_tok.tok + (W_cls.cls + b),
    which halves the read-out GEMM and removes the [B, N-1, 2C] copy; on the GPU in half precision the bias add, the exact
    (erf) GELU, the drop of the cls row and the token -> NHWC reshape are one HIP pass (ds_reassemble_readout)."""

    def __init__(self, in_features, start_index=1):
        super().__init__()
        self.start_index = start_index
        self.project = nn.Sequential(nn.Linear(2 * in_features, in_features), nn.GELU())

    def _split(self):
        lin = self.project[0]
        key = (lin.weight._version, lin.weight.data_ptr(), lin.weight.dtype)
        hit = getattr(self, "_w_split", None)
        if hit is None or hit[0] != key:
            if hit is not None:
                vm.cache_evicted()
            c = lin.in_features // 2
            hit = (key, lin.weight[:, :c].contiguous(), lin.weight[:, c:].contiguous())
            if not torch.is_grad_enabled():
                self._w_split = hit
        return hit[1], hit[2]

    def forward(self, x):
        if self.start_index != 1:
            readout = x[:, 0].unsqueeze(1).expand_as(x[:, self.start_index:])
            return self.project(torch.cat((x[:, self.start_index:], readout), -1))
        w_tok, w_cls = self._split()
        clsvec = F.linear(x[:, 0], w_cls, self.project[0].bias)               # [B, C]
        proj = F.linear(x, w_tok)                                             # [B, N, C] (row 0 is not used)
        if vm.half_on_gpu(x) and x.shape[-1] % 8 == 0:
            from src import _native
            return _native.reassemble_readout(proj, clsvec)
        return F.gelu(proj[:, 1:] + clsvec[:, None])

    def forward_padded(self, xp, n_valid):
        """The same on the PADDED block output [B, Np, C] the encoder leaves behind (contiguous: no copy of the tap, which a
        library GEMM on the sliced view x[:, :n_valid] makes): ONE in-tree GEMM whose epilogue adds the per-image cls vector,
        applies the exact GELU, drops the cls / pad rows and leaves the tokens NHWC (ds_linear_readout).  -> [B, N-1, C]."""
        from src import _native
        w_tok, w_cls = self._split()
        if (self.start_index == 1 and vm.READOUT_HIP and vm.LINEAR_HIP == "all" and vm.half_on_gpu(xp) and xp.is_contiguous()
                and _native.linear_readout_supported(xp, w_tok) and vm.hip_gemm_ok(xp.shape[0] * xp.shape[1], w_tok.shape[0])):
            # [B, C]: the cls half of the projection.  In-tree as well (rows padded to one 256-row tile): a library GEMM picks its
            # kernel -- and its summation order -- by the batch size, and the same image would come out differently at batch 8 and 32
            x0 = xp[:, 0].contiguous()
            if vm.invariant() and _native.linear_supported(x0, w_cls):
                clsvec = _native.linear(x0, w_cls, self.project[0].bias)
            else:
                clsvec = F.linear(x0, w_cls, self.project[0].bias)
            return _native.linear_readout(xp, n_valid, w_tok, clsvec)
        return self.forward(xp[:, :n_valid])


class _Skip(nn.Module):                 # placeholder for Transpose / Unflatten slots (no parameters) so indices match
    def forward(self, x):
        return x


def _postprocess(vit_features, out_features, tail):
    # indices 0..2 = readout, Transpose, Unflatten (utils.py:165-169); 3.. = the convolutions
    return nn.Sequential(ProjectReadout(vit_features), _Skip(), _Skip(), nn.Conv2d(vit_features, out_features, 1), *tail)


class BeitBackbone(nn.Module):
    """`pretrained` of the reference: .model (Beit) + .act_postprocess1-4 (utils.py:144-249, readout 'project')."""

    def __init__(self, model, features, vit_features, hooks):
        super().__init__()
        self.model = model
        self.hooks = list(hooks)
        f = features
        self.act_postprocess1 = _postprocess(vit_features, f[0], [nn.ConvTranspose2d(f[0], f[0], 4, 4, 0)])
        self.act_postprocess2 = _postprocess(vit_features, f[1], [nn.ConvTranspose2d(f[1], f[1], 2, 2, 0)])
        self.act_postprocess3 = _postprocess(vit_features, f[2], [])
        self.act_postprocess4 = _postprocess(vit_features, f[3], [nn.Conv2d(f[3], f[3], 3, 2, 1)])

    def forward(self, x):
        """forward_beit = forward_adapted_unflatten (utils.py:83-124): readout-project, tokens -> [B, C, h/16, w/16]
        (run-time grid, not the constructor's), then the per-tap convolutions."""
        taps, grid, n_valid = self.model.forward_taps(x, self.hooks, padded=True)
        outs = []
        for tap, post in zip(taps, (self.act_postprocess1, self.act_postprocess2, self.act_postprocess3, self.act_postprocess4)):
            y = post[0].forward_padded(tap, n_valid)           # ProjectReadout: [B, N-1, C]
            y = y.reshape(y.shape[0], grid[0], grid[1], y.shape[2]).permute(0, 3, 1, 2)     # NHWC view, no copy
            for layer in list(post)[3:]:
                y = vm.conv_module(layer, y)                    # library convolution + in-tree bias pass
            outs.append(y)
        return outs


def make_beit(name, hooks):
    cfg = {"beitl16_512": (512, 1024, 24, 16, [256, 512, 1024, 1024]),
           "beitl16_384": (384, 1024, 24, 16, [256, 512, 1024, 1024]),
           "beitb16_384": (384, 768, 12, 12, [96, 192, 384, 768])}[name]
    img, dim, depth, heads, features = cfg
    # timm: beit_large_* use init_values=1e-5, beit_base_* 0.1
    model = Beit(img, dim, depth, heads, init_values=1e-5 if dim == 1024 else 0.1)
    return BeitBackbone(model, features, dim, hooks), features
