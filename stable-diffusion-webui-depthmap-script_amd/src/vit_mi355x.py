"""Transformer-encoder machinery shared by the depth models (DINOv2 / DepthAnythingV2, BEiT-DPT, ViT-hybrid DPT),
laid out for an MI355X instead of being a transcription of the reference modules.

What is different from the reference's encoders (ddepth_anything_v2/depth_anything_v2/dinov2_layers/attention.py:49-62,
dmidas/backbones/beit.py:65-91):

* the token sequence is padded ONCE to a multiple of 64 after the patch embedding and stays padded through every
  block; pad tokens are ordinary rows of the GEMMs and are masked as attention KEYS, so they never reach a real token;
* the QKV projection is two library GEMMs writing the layouts the attention kernel consumes directly -- Q,K as
  ``[B, Np, 2, H, 64]`` (token major) and V TRANSPOSED as ``[B, H*64, Np]`` (key index contiguous: the MFMA operand of
  P.V) -- instead of one GEMM followed by reshape/permute copies;
* attention is ONE fused flash-style HIP kernel on the MFMA units (``ds_attention_fwd``, csrc/ds_attention.hip): the
  ``B x H x N x N`` score tensor of the reference (17 GB per block for BEiT-L at 1024^2) never exists; BEiT's relative
  position bias is an additive ``[H, Np, Np]`` operand that is built once per (layer, resolution) and cached, not
  re-interpolated and re-gathered in every block of every forward (beit.py:29-62);
* everything else (LayerNorm, GELU, the MLP GEMMs, convolutions of the DPT decoder) goes through the ROCm libraries
  behind torch (hipBLASLt / MIOpen): plain library GEMMs and convolutions are allowed to stay library calls.

float32 tensors take ``attention_reference`` (plain torch arithmetic, the definition the HIP kernel is tested against
and what the CPU parity tests against the reference's own modules run); float16 / bfloat16 tensors on a GPU take the HIP
kernel and raise if the library is missing -- there is no silent fallback.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

HEAD_DIM = 64
SEQ_ALIGN = 64


def pad_len(n):
    return (n + SEQ_ALIGN - 1) // SEQ_ALIGN * SEQ_ALIGN


def pad_tokens(x, n_pad):
    """[B, N, C] -> [B, n_pad, C] with zero rows appended."""
    b, n, c = x.shape
    if n == n_pad:
        return x
    out = x.new_zeros((b, n_pad, c))
    out[:, :n] = x
    return out


def attention_reference(qk, vt, n_valid, scale, bias=None):
    """Definition of the fused attention (float32 math).
    qk [B, Np, 2, H, D]; vt [B, H*D, Np]; keys >= n_valid are masked; bias optional [H, Np, Np] (added to q.k*scale).
    Returns [B, Np, H*D]; rows >= n_valid are unspecified (they are never read as keys)."""
    b, npad, _, h, d = qk.shape
    q = qk[:, :, 0].permute(0, 2, 1, 3).float()                # B H Np D
    k = qk[:, :, 1].permute(0, 2, 1, 3).float()
    v = vt.reshape(b, h, d, npad).permute(0, 1, 3, 2).float()   # B H Np D
    att = (q * scale) @ k.transpose(-2, -1)
    if bias is not None:
        att = att + bias.float().unsqueeze(0)
    if n_valid < npad:
        att[..., n_valid:] = float('-inf')
    att = att.softmax(dim=-1)
    out = att @ v
    return out.permute(0, 2, 1, 3).reshape(b, npad, h * d).to(qk.dtype)


def fused_attention(qk, vt, n_valid, scale, bias=None):
    if qk.dtype == torch.float32:
        return attention_reference(qk, vt, n_valid, scale, bias)
    from . import _native
    return _native.attention_fwd(qk, vt, n_valid, scale, bias)


class EncoderBlock(nn.Module):
    """Pre-norm transformer block with optional LayerScale.  Parameter NAMES follow the reference checkpoints of the
    family that instantiates it (see ``names``): DINOv2 ``norm1/attn.qkv/attn.proj/ls1.gamma/norm2/mlp.fc1/mlp.fc2/
    ls2.gamma`` (dinov2_layers/block.py:60-83); the BEiT subclass keeps timm's ``gamma_1/gamma_2`` and q/v biases."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, eps=1e-6):
        super().__init__()
        assert dim == num_heads * HEAD_DIM, "the attention kernel is built for head_dim 64"
        self.dim, self.num_heads = dim, num_heads
        self.scale = HEAD_DIM ** -0.5
        self.eps = eps

    # subclasses provide: ln1(x), qkv_weights() -> (w_qk, b_qk, w_v, b_v), proj(o), gamma1/gamma2 (or None), ln2, mlp
    def attention_bias(self, n_pad, grid_hw, dtype, device):
        return None

    def forward_padded(self, x, n_valid, grid_hw=None):
        b, npad, c = x.shape
        h = self.ln1(x)
        w_qk, b_qk, w_v, b_v = self.qkv_weights()
        qk = F.linear(h, w_qk, b_qk).view(b, npad, 2, self.num_heads, HEAD_DIM)
        vt = torch.matmul(w_v, h.transpose(1, 2))               # [B, C, Np]: V transposed, straight out of the GEMM
        if b_v is not None:
            vt = vt + b_v.view(1, c, 1)
        bias = self.attention_bias(npad, grid_hw, x.dtype, x.device)
        o = fused_attention(qk, vt, n_valid, self.scale, bias)
        o = self.proj(o)
        g1, g2 = self.gammas()
        x = x + (o * g1 if g1 is not None else o)
        m = self.mlp(self.ln2(x))
        x = x + (m * g2 if g2 is not None else m)
        return x


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (dinov2_layers/mlp.py:20-41; timm Mlp has the same parameter names)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


def count_encoder_flops(depth, n, dim, mlp_ratio=4.0):
    """Algorithmic FLOPs of `depth` blocks on n tokens: 2*(4 + 2*mlp_ratio)*n*dim^2 linear + 4*n^2*dim attention."""
    lin = 2.0 * (4.0 + 2.0 * mlp_ratio) * n * dim * dim
    att = 4.0 * n * n * dim
    return depth * (lin + att)
