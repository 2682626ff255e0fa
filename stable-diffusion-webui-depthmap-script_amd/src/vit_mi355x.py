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
import contextlib
import functools
import math
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

HEAD_DIM = 64
SEQ_ALIGN = 64

_CONSTANTS = {}

# Tensors derived from parameters or from the input size are cached on the modules (resized position embeddings, packed
# relative-position bias, folded / repacked weights).  A captured hipGraph (src/hip_graph.py) holds raw POINTERS to the
# cache entries its kernels read, so an entry that is dropped while such a graph is alive would be read after its memory
# was reused.  Every eviction therefore bumps this epoch; GraphedForward drops its graphs when it sees a new epoch and
# captures again on demand.  (Size-keyed caches keep a few entries so that alternating between shapes evicts nothing.)
CACHE_EPOCH = [0]
SIZE_CACHE_ENTRIES = 6


def cache_evicted():
    CACHE_EPOCH[0] += 1


def cache_store(cache, key, value, entries=None):
    """Insert into a size-keyed module cache; when it is full everything goes (and the epoch moves)."""
    if len(cache) >= (SIZE_CACHE_ENTRIES if entries is None else entries):
        cache.clear()
        cache_evicted()
    cache[key] = value
    return value


def device_constant(values, device, dtype=torch.float32):
    """Small constant tensor (ImageNet mean/std, filter taps ...) resident on ``device``, created ONCE per
    (values, device, dtype).  ``torch.tensor(list, device=gpu)`` is a pageable host-to-device copy on every call: a
    synchronising memcpy in the middle of the forward, and an operation a capturing stream refuses (hip_graph.py)."""
    key = (tuple(values), str(device), dtype)
    t = _CONSTANTS.get(key)
    if t is None:
        t = _CONSTANTS[key] = torch.tensor(list(values), dtype=dtype, device=device)
    return t


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


TIGHT_PAD = os.environ.get("DS_TIGHT_PAD", "1") != "0"      # A/B switch: 0 = pad every sequence to a multiple of 64 (rounds 1-3)


def pad_len(n, batch=None):
    """Token stride of a sequence of n tokens.  Without a batch size: n rounded up to 64 (one key tile of the attention
    kernel).  With one: the TIGHTEST pad (a multiple of 8, 16 or 32 instead of 64) whose batch * stride rows are whole
    256-row panels -- the unit of the token GEMMs and what ds_linear_vt needs for its columns: 32 x 1025 tokens pad to 1032
    (129 row panels; 1088 would be 136, and fc1's 2176 tiles a ninth, half-empty round on 256 CUs), 8 x 2443 to 2464.  The
    attention kernel takes any stride that is a multiple of 8; when no tight pad fits, 64."""
    if batch and TIGHT_PAD:
        for a in (8, 16, 32):
            s = (n + a - 1) // a * a
            if (batch * s) % 256 == 0:
                return s
    return (n + SEQ_ALIGN - 1) // SEQ_ALIGN * SEQ_ALIGN


def pad_tokens(x, n_pad):
    """[B, N, C] -> [B, n_pad, C] with zero rows appended."""
    b, n, c = x.shape
    if n == n_pad:
        return x
    out = x.new_zeros((b, n_pad, c))
    out[:, :n] = x
    return out


SCORE_BYTES_MAX = 1 << 30         # attention_reference never holds more than this many bytes of logits at once


def attention_reference(qk, vt, n_valid, scale, bias=None):
    """Definition of the fused attention (float32 math).
    qk [B, Np, 2, H, D]; vt [B, H*D, Np]; keys >= n_valid are masked; bias optional: a [H, Np, Np] tensor (added to
    q.k*scale) or an object with ``rows(q0, q1) -> [H, q1-q0, Np]`` (a bias too large to keep dense, built per query tile).
    Returns [B, Np, H*D]; rows >= n_valid are unspecified (they are never read as keys).
    Long sequences (Boost's whole-image pass reaches 10^4 tokens: B*H*Np^2 floats = 6.4 GB per block) are processed in
    query tiles of at most SCORE_BYTES_MAX bytes of logits; softmax rows are independent, so the result is the same."""
    b, npad, _, h, d = qk.shape
    # float64 in, float64 math (the "truth" evaluation of the parity budgets); the stock twin (stock_routing) computes in the
    # tensors' own dtype like the reference's modules do; float32 otherwise
    ct = qk.dtype if (qk.dtype == torch.float64 or STOCK[0]) else torch.float32
    q = qk[:, :, 0].permute(0, 2, 1, 3).to(ct)                 # B H Np D
    k = qk[:, :, 1].permute(0, 2, 1, 3).to(ct)
    v = vt.reshape(b, h, d, npad).permute(0, 1, 3, 2).to(ct)    # B H Np D
    kt = k.transpose(-2, -1)
    rows = max(1, min(npad, SCORE_BYTES_MAX // max(1, b * h * npad * 4)))
    outs = []
    for q0 in range(0, npad, rows):
        q1 = min(npad, q0 + rows)
        att = (q[:, :, q0:q1] * scale) @ kt
        if bias is not None:
            bt = bias.rows(q0, q1) if hasattr(bias, "rows") else bias[:, q0:q1]
            att = att + bt.to(ct).unsqueeze(0)
        if n_valid < npad:
            att[..., n_valid:] = float('-inf')
        outs.append(att.softmax(dim=-1) @ v)
    out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)
    return out.permute(0, 2, 1, 3).reshape(b, npad, h * d).to(qk.dtype)


def v_transposed(w_v, h):
    """V^T = W_v . h^T, [B, C, Np] out of h [B, Np, C].  float16 / bfloat16 on a GPU with every token GEMM in-tree
    (DS_LINEAR=all): ds_linear_vt, one GEMM over all tokens of the batch whose epilogue writes each batch element's
    [C, Np] block.  Otherwise ONE batched library GEMM whose second operand is read transposed in place
    (torch.matmul(w_v, h.transpose(1, 2)) would first materialise h^T)."""
    b = h.shape[0]
    if half_on_gpu(h) and LINEAR_HIP == "all":
        from . import _native
        if _native.linear_vt_supported(w_v, h) and hip_gemm_ok(w_v.shape[0], h.shape[0] * h.shape[1]):
            return _native.linear_vt(w_v, h)
    return torch.bmm(w_v.unsqueeze(0).expand(b, -1, -1), h.transpose(1, 2))


def folded_proj_bias(proj, b_v):
    """softmax rows sum to one, so attn @ (v + b_v) = attn @ v + b_v: the V bias commutes with the attention and folds
    into the output projection's bias, b_p + W_p . b_v -- one [C] vector per block instead of a pass over V^T.
    The folded vector depends only on parameters: cached on the module per parameter version (inference)."""
    if b_v is None:
        return proj.bias
    key = (proj.weight._version, proj.bias._version, b_v._version, b_v.data_ptr(), proj.weight.data_ptr(), proj.weight.dtype)
    hit = getattr(proj, "_folded_bias", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    out = proj.bias + F.linear(b_v.to(proj.weight.dtype), proj.weight)
    if not torch.is_grad_enabled():
        if hit is not None:
            cache_evicted()
        proj._folded_bias = (key, out)
    return out


LOG2E = 1.4426950408889634


# Which token GEMMs take the in-tree MFMA kernel (csrc/ds_linear.hip) on float16 / bfloat16 CUDA tensors:
#   "all" (default)  every Linear of the encoder blocks: Q/K projection, V^T (written transposed by the epilogue), the
#                    attention output projection and fc2 with LayerScale + the residual add in their epilogues, fc1 + GELU;
#   "proj"           fc1 + GELU and the fused output projection only;   "gelu"  fc1 + GELU only (round 2's default);
#   "0"              none (library GEMMs + aten GELU).
LINEAR_HIP = os.environ.get("DS_LINEAR", "all")


# The in-tree kernel's unit is a 256 x 256 tile on one CU: a launch with only a few tiles leaves most of the chip idle (a
# batch-1 ViT-B forward is 9 tiles), and the library's small-tile kernels are the better tool there.
LINEAR_HIP_MIN_TILES = int(os.environ.get("DS_LINEAR_MIN_TILES", 96))


def hip_gemm_ok(rows, out_features):
    # (rows >= 256 whatever the threshold says: the fused entry points -- ds_linear_residual, ds_linear_vt -- take whole tiles)
    return rows >= 256 and ((rows + 255) // 256) * (out_features // 256) >= LINEAR_HIP_MIN_TILES


# Every "half precision on the GPU -> in-tree kernel" decision of the networks goes through this predicate, so that ONE switch
# (stock_routing below) turns the same network into its stock-torch twin: aten LayerNorm / softmax / GELU / interpolate and library
# GEMMs / convolutions in half, i.e. the arithmetic the reference itself runs on a GPU in half precision.
STOCK = [False]


def half_on_gpu(x):
    return x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and not STOCK[0]


@contextlib.contextmanager
def stock_routing():
    """Inside the block NO in-tree kernel runs: the networks take the plain torch operation sequence in whatever dtype they are in
    (the float32 parity path's code, in half: attention as softmax(q.k^T * scale + bias) @ v with aten ops).  The yardstick of the
    float16 parity tests -- how far the reference's OWN half-precision arithmetic is from float32 on a given network -- never a
    product path."""
    saved = STOCK[0]
    STOCK[0] = True
    try:
        yield
    finally:
        STOCK[0] = saved


# ---- deterministic library convolutions (round 6) ---------------------------------------------------------------------------------
# The reference runs one image at a time (src/core.py:133): an image's depth cannot depend on its neighbours, on its position in a
# batch or on the launch.  The in-tree kernels have that property (one accumulation chain per output element, wherever its tile
# lands); MIOpen's solvers with split-K atomics do not -- measured on a fresh box (tools/batch_invariance_check.py,
# dpt_beit_large_512 at batch 32): the same image at units 0 / 13 / 31 differs by 5e-3 of the depth range and two launches of the
# same batch by 6e-3; with torch.backends.cudnn.deterministic (MIOpen: MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC) all of it is 0.
# BUT the deterministic attribute leaves MIOpen, for six of dpt_beit_large_512's convolutions, with nothing but its NAIVE solver
# (naive_conv_ab_nonpacked_fwd_nhwc_half_double_half: 42-199 ms per call, 390 ms per step instead of 40).  So the cure is the
# other one: INVARIANT below -- every convolution the in-tree GEMM can express goes in-tree, whatever the size of the launch, and
# the forward of the metric's network has no library convolution left.  deterministic_library() stays as an opt-in
# (DS_DETERMINISTIC=1) for the networks that still call the library: reference counted (the funnel drives forwards from several
# threads), the caller's setting restored when the last forward leaves.
DETERMINISTIC_LIBRARY = os.environ.get("DS_DETERMINISTIC", "0") == "1"
# INVARIANT (default on; DS_INVARIANT=0: the tile-count thresholds of rounds 4-5 decide again, A/B runs): in-tree kernels wherever
# the shapes allow -- an image's depth does not depend on its position in the batch, on its neighbours or on the launch
# (tests/test_gpu_models.py: the metric's batch, the same image at units 0 / 13 / 31, bit-identical).
INVARIANT = os.environ.get("DS_INVARIANT", "1") != "0"
_size_routed = threading.local()


def invariant():
    """INVARIANT, unless the calling thread is inside size_routed()."""
    return INVARIANT and not getattr(_size_routed, "depth", 0)


@contextlib.contextmanager
def size_routed():
    """Inside the block the tile-count thresholds decide again whether a convolution goes in-tree (what DS_INVARIANT=0 selects for the
    process).  For networks whose forward calls library kernels anyway -- dpt_hybrid_384's ResNetV2 stem is MIOpen -- sending every
    tiny launch through the in-tree GEMM buys no invariance and costs latency: BASELINE config 2 (one 512^2 image) 3.73 ms with the
    rule, 3.30 ms with the thresholds (a 3x3 convolution of 9 tiles is a chain of 36 K-tiles on nine CUs in k_linear256, MIOpen
    splits it)."""
    _size_routed.depth = getattr(_size_routed, "depth", 0) + 1
    try:
        yield
    finally:
        _size_routed.depth -= 1
_det_lock = threading.Lock()
_det_state = [0, False]                      # forwards inside, the caller's torch.backends.cudnn.deterministic


@contextlib.contextmanager
def deterministic_library():
    if not DETERMINISTIC_LIBRARY:
        yield
        return
    with _det_lock:
        if _det_state[0] == 0:
            _det_state[1] = torch.backends.cudnn.deterministic
            torch.backends.cudnn.deterministic = True
        _det_state[0] += 1
    try:
        yield
    finally:
        with _det_lock:
            _det_state[0] -= 1
            if _det_state[0] == 0:
                torch.backends.cudnn.deterministic = _det_state[1]


def deterministic_forward(fn):
    """Decorator of a network's top-level forward: library convolutions inside it are the deterministic ones (see above)."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with deterministic_library():
            return fn(*args, **kwargs)
    return wrapped


@contextlib.contextmanager
def library_routing():
    """Inside the block every token GEMM and every 3x3 convolution of the networks goes through the ROCm libraries behind
    torch (what DS_LINEAR=0 DS_CONV=0 select at import time); the fused attention, LayerNorm and element-wise kernels stay.
    The reference point of the route checks (tests/test_gpu_models.py, bench.py's route_check): the SAME network, the same
    input, the other GEMM / convolution implementation."""
    global LINEAR_HIP, CONV_HIP
    saved = LINEAR_HIP, CONV_HIP
    LINEAR_HIP, CONV_HIP = "0", False
    try:
        yield
    finally:
        LINEAR_HIP, CONV_HIP = saved


def linear(x, weight, bias=None, gelu=False):
    """[gelu](x @ weight.T + bias).  float16 / bfloat16 on a GPU: ds_linear when the switch above selects it (erf-GELU
    on the fp32 accumulator); everything else: the library GEMM and aten's exact GELU."""
    if half_on_gpu(x) and (LINEAR_HIP == "all" or (gelu and LINEAR_HIP in ("gelu", "proj"))):
        from . import _native
        if _native.linear_supported(x, weight) and (gelu or hip_gemm_ok(x.numel() // x.shape[-1], weight.shape[0])):
            return _native.linear(x, weight, bias, gelu)
    y = F.linear(x, weight, bias)
    return F.gelu(y) if gelu else y


def fused_attention(qk, vt, n_valid, scale, bias=None):
    """bias: None, or what the block's attention_bias() cached for this dtype: the packed operand of the HIP kernel
    (_native.attention_bias_pack) for float16 / bfloat16, a padded [H, Np(query), Np(key)] tensor for float32."""
    if qk.dtype in (torch.float32, torch.float64) or STOCK[0]:
        return attention_reference(qk, vt, n_valid, scale, bias)
    from . import _native
    return _native.attention_fwd(qk, vt, n_valid, scale, bias)


class EncoderBlock(nn.Module):
    """Pre-norm transformer block with LayerScale.  Parameter NAMES follow the reference checkpoints of the family that
    instantiates it: DINOv2 ``norm1/attn.qkv/attn.proj/ls1.gamma/norm2/mlp.fc1/mlp.fc2/ls2.gamma`` (dinov2_layers/
    block.py:60-83); the BEiT subclass keeps timm's ``gamma_1/gamma_2``, ``q_bias/v_bias`` and the bias table.
    Subclasses provide: norm1, norm2 (nn.LayerNorm), mlp, qkv_weights() -> (w_qk, b_qk, w_v, b_v),
    proj(o, b_v), gammas() -> (g1, g2), and optionally attention_bias()."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        assert dim == num_heads * HEAD_DIM, "the attention kernel is built for head_dim 64"
        self.dim, self.num_heads = dim, num_heads
        self.scale = HEAD_DIM ** -0.5

    def attention_bias(self, n_pad, grid_hw, dtype, device):
        return None

    def attend_o(self, h, n_valid, grid_hw):
        """LayerNorm-ed tokens -> (attention output before the projection, V bias to fold into the projection bias)."""
        b, npad, c = h.shape
        w_qk, b_qk, w_v, b_v = self.qkv_weights()
        qk = linear(h, w_qk, b_qk).view(b, npad, 2, self.num_heads, HEAD_DIM)
        vt = v_transposed(w_v, h)                               # [B, C, Np]: V transposed, straight out of the GEMM
        return fused_attention(qk, vt, n_valid, self.scale, self.attention_bias(npad, grid_hw, h.dtype, h.device)), b_v

    def attend(self, h, n_valid, grid_hw):
        """LayerNorm-ed tokens -> projected attention output (before LayerScale / residual)."""
        o, b_v = self.attend_o(h, n_valid, grid_hw)
        return self.proj(o, b_v)                                # V bias folded into the projection bias

    def forward_padded(self, x, n_valid, grid_hw=None):
        """Reference operation order with stock torch element-wise ops (float32 parity path)."""
        g1, g2 = self.gammas()
        a = self.attend(self.norm1(x), n_valid, grid_hw)
        x = x + (a if g1 is None else g1 * a)
        m = self.mlp(self.norm2(x))
        return x + (m if g2 is None else g2 * m)


# Round 5 experiment, built, measured, NOT the default: the LayerNorm in front of qkv and of fc1 folded INTO those GEMMs (ds_linear_ln /
# ds_linear_vt_ln: the GEMM reads the residual stream, its epilogue applies rstd and -mean * rstd * colsum), the LayerNorm pass -- read
# x, write the normalised copy -- replaced by a statistics pass that only reads x (the round-4 verdict's "do less" item, estimated at
# -0.6 ms per step).  Measured on one box, back to back (profiles/round5_ln_fold_ab.txt): 789.2 / 790.7 pairs/s folded against 798.7 /
# 798.3 with ds_residual_layernorm -- per step the Q/K GEMMs take +0.54 ms and fc1 + GELU +0.40 ms (17-22 us per launch: the
# epilogue is exposed time of a workgroup that owns its CU, and it now waits for two more operand fetches per tile and does three
# more multiply-adds per output), the statistics passes save 0.53 ms against the LayerNorm passes (11 us of 23 each): a net loss of
# 0.45 ms.  A bandwidth-bound streaming pass is the cheaper place for this arithmetic.  DS_LN_FOLD=1 takes the folded route.
LN_FOLD = os.environ.get("DS_LN_FOLD", "0") != "0"
LN_FOLD_CHANNELS = (384, 768, 1024, 1536)


def _ln_fold_params(blk, dtype):
    """Per block, once per parameter version: W' = W diag(ln_weight) (rounded to the network's dtype), colsum = W'.sum(1) in float32
    (of the ROUNDED weights: the identity LN(x) W^T = rstd (x W'^T - mean colsum) then holds exactly for what the MFMA computes),
    b' = b + W ln_bias, for the Q/K projection, V (whose constant goes into the projection bias like the V bias) and fc1."""
    w_qk, b_qk, w_v, b_v = blk.qkv_weights()
    n1, n2, fc1, pr = blk.norm1, blk.norm2, blk.mlp.fc1, blk.attn.proj
    tensors = (w_qk, b_qk, w_v, b_v, n1.weight, n1.bias, n2.weight, n2.bias, fc1.weight, fc1.bias, pr.weight, pr.bias)
    key = tuple((None if t is None else (t._version, t.data_ptr())) for t in tensors) + (dtype,)
    hit = getattr(blk, "_ln_fold", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    if hit is not None:
        cache_evicted()                     # a live hipGraph may still read the old tensors

    def fold(w, b, ln):
        wf = w.float()
        ws = (wf * ln.weight.float()[None, :]).to(dtype).contiguous()
        const = wf @ ln.bias.float() + (0.0 if b is None else b.float())
        return ws, ws.float().sum(1).contiguous(), const
    wqk, sqk, cqk = fold(w_qk, b_qk, n1)
    wv, sv, cv = fold(w_v, b_v, n1)
    w1, s1, c1 = fold(fc1.weight, fc1.bias, n2)
    # attn @ (v + c_v) = attn @ v + c_v (softmax rows sum to one): V's constant -- its bias and W_v . ln_bias -- joins the projection bias
    pb = ((0.0 if pr.bias is None else pr.bias.float()) + pr.weight.float() @ cv).to(dtype).contiguous()
    out = {"wqk": wqk, "sqk": sqk, "bqk": cqk.to(dtype).contiguous(), "wv": wv, "sv": sv, "proj_bias": pb,
           "w1": w1, "s1": s1, "b1": c1.to(dtype).contiguous()}
    if not torch.is_grad_enabled():
        blk._ln_fold = (key, out)
    return out


def _ln_fold_ok(blocks, x):
    if not (LN_FOLD and LINEAR_HIP == "all" and half_on_gpu(x) and x.is_contiguous() and x.shape[2] in LN_FOLD_CHANNELS):
        return False
    from . import _native
    rows = x.shape[0] * x.shape[1]
    if not hip_gemm_ok(rows, x.shape[2]):
        return False
    for blk in blocks:
        mlp, pr = getattr(blk, "mlp", None), getattr(getattr(blk, "attn", None), "proj", None)
        if not (isinstance(mlp, Mlp) and pr is not None and mlp.fc2.bias is not None and hasattr(blk, "qkv_weights")
                and isinstance(getattr(blk, "norm1", None), nn.LayerNorm) and isinstance(getattr(blk, "norm2", None), nn.LayerNorm)
                and blk.norm1.elementwise_affine and blk.norm2.elementwise_affine and blk.norm1.bias is not None and blk.norm2.bias is not None):
            return False
        w_qk, _, w_v, _ = blk.qkv_weights()
        if not (_native.linear_supported(x, w_qk) and _native.linear_supported(x, pr.weight) and _native.linear_supported(x, mlp.fc1.weight)
                and mlp.fc2.weight.shape[0] % 256 == 0 and mlp.fc2.weight.shape[1] % 128 == 0 and _native.linear_vt_supported(w_v, x)):
            return False
    return True


def _run_blocks_ln_fold(blocks, x, n_valid, grid_hw, take, padded_taps):
    """The encoder with every LayerNorm folded into the GEMM behind it: per block 2 statistics passes, 5 GEMMs (Q/K, V^T, projection +
    LayerScale + residual, fc1 + GELU, fc2 + LayerScale + residual) and the fused attention -- no normalised copy of the residual
    stream is ever written."""
    from . import _native
    taps = {}
    b, npad, c = x.shape
    for i, blk in enumerate(blocks):
        p = _ln_fold_params(blk, x.dtype)
        g1, g2 = blk.gammas()
        st = _native.row_stats(x, blk.norm1.eps)
        qk = _native.linear_ln(x, p["wqk"], p["sqk"], p["bqk"], st).view(b, npad, 2, blk.num_heads, HEAD_DIM)
        vt = _native.linear_vt_ln(p["wv"], p["sv"], x, st)
        o = fused_attention(qk, vt, n_valid, blk.scale, blk.attention_bias(npad, grid_hw, x.dtype, x.device))
        x = _native.linear_residual(o, blk.attn.proj.weight, p["proj_bias"], g1, x)
        st = _native.row_stats(x, blk.norm2.eps)
        a = _native.linear_ln(x, p["w1"], p["s1"], p["b1"], st, gelu=True)
        x = _native.linear_residual(a, blk.mlp.fc2.weight, blk.mlp.fc2.bias, g2, x)
        if i in take:
            taps[i] = x if padded_taps else x[:, :n_valid]
    return x, taps


def run_blocks(blocks, x, n_valid, grid_hw, take, padded_taps=False):
    """Run the encoder on the padded sequence x [B, Np, C]; returns (x, {index: tap}) with taps = unpadded block outputs
    (padded_taps: the padded [B, Np, C] block outputs themselves -- contiguous, what ds_linear_readout reads).
    float16/bfloat16 on a GPU: LayerScale + residual + the NEXT LayerNorm are one fused pass (ds_residual_layernorm), so a
    block is 5 GEMMs + 1 attention + 1 GELU + 2 fused element-wise kernels.  Otherwise: the plain torch sequence."""
    taps = {}
    fast = half_on_gpu(x)
    if not fast:
        for i, blk in enumerate(blocks):
            x = blk.forward_padded(x, n_valid, grid_hw)
            if i in take:
                taps[i] = x if padded_taps else x[:, :n_valid]
        return x, taps
    if _ln_fold_ok(blocks, x):
        return _run_blocks_ln_fold(blocks, x, n_valid, grid_hw, take, padded_taps)
    from . import _native
    n = len(blocks)
    _, h = _native.residual_layernorm(x, None, None, blocks[0].norm1.weight, blocks[0].norm1.bias, blocks[0].norm1.eps)
    big = hip_gemm_ok(x.shape[0] * x.shape[1], x.shape[2])      # (the fused epilogues pay only where the tile kernel does)
    for i, blk in enumerate(blocks):
        g1, g2 = blk.gammas()
        pr = getattr(getattr(blk, "attn", None), "proj", None)
        if (LINEAR_HIP in ("proj", "all") and pr is not None and big and _native.linear_supported(x, pr.weight)):
            # the projection GEMM adds LayerScale and the residual in its epilogue (ds_linear_residual): the LayerNorm pass
            # then reads x once instead of x and the branch
            o, b_v = blk.attend_o(h, n_valid, grid_hw)
            x = _native.linear_residual(o, pr.weight, folded_proj_bias(pr, b_v), g1, x)
            _, h2 = _native.residual_layernorm(x, None, None, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        else:
            p = blk.attend(h, n_valid, grid_hw)
            x, h2 = _native.residual_layernorm(x, p, g1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        mlp = blk.mlp
        fused_fc2 = (LINEAR_HIP == "all" and isinstance(mlp, Mlp) and big and _native.linear_supported(x, mlp.fc2.weight)
                     and _native.linear_supported(h2, mlp.fc1.weight) and mlp.fc2.bias is not None)
        if fused_fc2:                                        # x + gamma_2 * fc2(gelu(fc1(h2))): both tails in GEMM epilogues
            a = _native.linear(h2, mlp.fc1.weight, mlp.fc1.bias, True)
            x = _native.linear_residual(a, mlp.fc2.weight, mlp.fc2.bias, g2, x)
            if i + 1 < n:
                nxt = blocks[i + 1].norm1
                _, h = _native.residual_layernorm(x, None, None, nxt.weight, nxt.bias, nxt.eps)
        else:
            m = mlp(h2)
            if i + 1 < n:
                nxt = blocks[i + 1].norm1
                x, h = _native.residual_layernorm(x, m, g2, nxt.weight, nxt.bias, nxt.eps)
            else:
                x = x + (m if g2 is None else g2 * m)
        if i in take:
            taps[i] = x if padded_taps else x[:, :n_valid]
    return x, taps


def interpolate_bilinear(x, size=None, scale_factor=None, align_corners=True):
    """F.interpolate(mode="bilinear") of the DPT decoders: the HIP kernel for half-precision channels_last activations on
    the GPU (one pass at HBM speed), torch everywhere else (float32 parity path, CPU)."""
    if half_on_gpu(x) and x.shape[1] % 8 == 0:
        from . import _native
        return _native.upsample_bilinear(x, size=size, scale_factor=scale_factor, align_corners=align_corners)
    if size is not None:
        return F.interpolate(x, size=size, mode="bilinear", align_corners=align_corners)
    return F.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=align_corners)


# 3x3 decoder convolutions through the in-tree implicit GEMM (csrc/ds_linear.hip) when the launch has at least
# CONV_HIP_MIN_TILES tiles of 256 x 256 (DS_CONV_MIN_TILES); "0" keeps every convolution in the library.  A launch below 256
# tiles is ONE round of the persistent kernel whatever its size (~60 us at K = 2304), so small maps only pay where the library
# is worse than that: at 128 tiles (32 x 32^2 maps of the batch-32 benchmark) the two are level (61 vs 70 us), at 32 tiles
# (16^2 maps) the library is ahead -- but for Depth-Anything-V2's 37 x 66 maps at batch 8 (76 tiles) MIOpen's heuristic picks a
# 327 us kernel.  Round 4: 128 -> 48 (c5 213.9 -> 216.5 pairs/s on one box; 16 would give 219.2 there and cost the batch-32
# benchmark 0.6 %).
CONV_HIP = os.environ.get("DS_CONV", "1") != "0"
CONV_HIP_MIN_TILES = int(os.environ.get("DS_CONV_MIN_TILES", 48))
CONV_HEAD_HIP = os.environ.get("DS_CONV_HEAD", "1") != "0"      # A/B switch: the 256 x 128 tiles (out_channels % 256 == 128)
CONV_RELU_IN = os.environ.get("DS_CONV_RELU_IN", "1") != "0"    # A/B switch: conv1(relu(x)) of a residual unit with the ReLU inside the kernel


def conv3x3_hip_ok(conv, x):
    if not (CONV_HIP and half_on_gpu(x) and x.dim() == 4):
        return False
    from . import _native
    if conv.out_channels % 256 != 0 and not CONV_HEAD_HIP:
        return False
    tiles = (x.shape[0] * x.shape[2] * x.shape[3] + 255) // 256 * ((conv.out_channels + 255) // 256)
    return _native.conv3x3_supported(conv, x) and tiles >= (1 if invariant() else CONV_HIP_MIN_TILES)


def conv2d(conv, x):
    """conv(x): ds_conv3x3_nhwc where it applies (scratch.layerN_rn of the decoders), the library otherwise."""
    if conv3x3_hip_ok(conv, x):
        from . import _native
        return _native.conv3x3(conv, x)
    return conv(x)


CONV_BIAS_HIP = os.environ.get("DS_CONV_BIAS", "1") != "0"          # A/B switch
# Round 5: the last library GEMM-shaped work of a forward goes through the in-tree GEMM (csrc/ds_linear.hip) -- A/B switches:
CONV1X1_HIP = os.environ.get("DS_CONV1X1", "1") != "0"            # 1x1 convolutions (reassemble, fusion blocks' out_conv) = ds_linear on NHWC rows
CONVT_HIP = os.environ.get("DS_CONVT", "1") != "0"                # ConvTranspose2d with kernel == stride = ds_linear_shuffle
READOUT_HIP = os.environ.get("DS_READOUT", "1") != "0"            # the read-out projection on the padded taps = ds_linear_readout
CONV1X1_MIN_TILES = int(os.environ.get("DS_CONV1X1_MIN_TILES", 96))


def conv1x1_hip_ok(layer, x):
    """A 1x1, stride-1 nn.Conv2d on a half-precision channels_last activation whose GEMM -- [pixels, in] x [out, in]^T -- fills the
    chip: the NHWC memory IS the row-major [pixels, in] operand, no copy on either side."""
    if not (CONV1X1_HIP and LINEAR_HIP == "all" and half_on_gpu(x) and x.dim() == 4 and type(layer) is nn.Conv2d):
        return False
    if not (tuple(layer.kernel_size) == (1, 1) and tuple(layer.stride) == (1, 1) and tuple(layer.padding) == (0, 0) and layer.groups == 1
            and tuple(layer.dilation) == (1, 1)):
        return False
    pixels = x.shape[0] * x.shape[2] * x.shape[3]
    return (layer.out_channels % 256 == 0 and layer.in_channels % 128 == 0 and 128 <= layer.in_channels <= 16384 and pixels >= 256
            and ((pixels + 255) // 256) * (layer.out_channels // 256) >= (1 if invariant() else CONV1X1_MIN_TILES)
            and x.is_contiguous(memory_format=torch.channels_last))


def conv1x1(layer, x):
    from . import _native
    b, c, h, w = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(b * h * w, c)                   # a view of the channels_last memory
    y = _native.linear(rows, layer.weight.reshape(layer.out_channels, c), layer.bias)
    return y.view(b, h, w, layer.out_channels).permute(0, 3, 1, 2)


def conv_transpose_hip_ok(layer, x):
    if not (CONVT_HIP and LINEAR_HIP == "all" and half_on_gpu(x) and x.dim() == 4 and type(layer) is nn.ConvTranspose2d):
        return False
    from . import _native
    if not (_native.conv_transpose_shuffle_supported(layer, x) and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    pixels = x.shape[0] * x.shape[2] * x.shape[3]
    n = layer.stride[0] * layer.stride[0] * layer.out_channels
    return ((pixels + 255) // 256) * (n // 256) >= (1 if invariant() else CONV1X1_MIN_TILES)


# ---- round 6: the last library convolutions of the DPT networks as in-tree GEMMs (INVARIANT above) -------------------------------
def _gemm_weight(layer, x, kpad):
    """[out, kh, kw, in] image of a convolution's weight, flattened to [out, kh * kw * in] and zero padded to kpad columns, in the
    activation's dtype; cached on the module (keyed by the parameter's version)."""
    w = layer.weight
    key = (w.data_ptr(), w._version, x.dtype, kpad)
    cache = getattr(layer, "_ds_gemm_weight", None)
    if cache is None or cache[0] != key:
        wm = w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(x.dtype)
        if wm.shape[1] != kpad:
            wm = torch.cat([wm, wm.new_zeros((wm.shape[0], kpad - wm.shape[1]))], 1)
        cache = (key, wm.contiguous())
        layer._ds_gemm_weight = cache
    return cache[1]


def patch_embed_hip_ok(conv, x):
    """A patch embedding -- Conv2d with kernel == stride, no padding -- on a half-precision channels_last image: non-overlapping
    patches, i.e. a GEMM on [patches, kh * kw * in] (dmidas/backbones/beit.py:18-27, vit.py, dinov2_layers/patch_embed.py)."""
    if not (invariant() and LINEAR_HIP == "all" and half_on_gpu(x) and x.dim() == 4 and type(conv) is nn.Conv2d):
        return False
    k = conv.kernel_size
    return (k[0] == k[1] and tuple(conv.stride) == tuple(k) and tuple(conv.padding) == (0, 0) and conv.groups == 1
            and tuple(conv.dilation) == (1, 1) and conv.out_channels % 256 == 0 and x.shape[1] == conv.in_channels
            and x.shape[2] >= k[0] and x.shape[3] >= k[1] and x.is_contiguous(memory_format=torch.channels_last))


def patch_embed_tokens(conv, x):
    """conv(x).flatten(2).transpose(1, 2) -> [B, patches, out] through ds_linear: the patches are gathered once from the NHWC image
    ([B, gh, p, gw, p, C] -> [B, gh, gw, p, p, C]: one copy), K = p * p * C is zero padded to a multiple of 128 (DINOv2: 588 -> 640)."""
    from . import _native
    b, c, h, w = x.shape
    p = conv.kernel_size[0]
    gh, gw = h // p, w // p
    k = p * p * c
    kpad = (k + 127) // 128 * 128
    xv = x.permute(0, 2, 3, 1)[:, :gh * p, :gw * p].reshape(b, gh, p, gw, p, c).permute(0, 1, 3, 2, 4, 5)
    if kpad == k:
        rows = xv.reshape(b * gh * gw, k)
    else:
        rows = x.new_zeros((b * gh * gw, kpad))
        rows[:, :k] = xv.reshape(b * gh * gw, k)
    y = _native.linear(rows, _gemm_weight(conv, x, kpad), conv.bias)
    return y.view(b, gh * gw, conv.out_channels)


def strided3x3_hip_ok(layer, x):
    """The strided 3x3 of the reassemble stage (act_postprocess4 / resize_layers[3]: Conv2d(C, C, 3, stride 2, padding 1))."""
    if not (invariant() and LINEAR_HIP == "all" and half_on_gpu(x) and x.dim() == 4 and type(layer) is nn.Conv2d):
        return False
    return (tuple(layer.kernel_size) == (3, 3) and tuple(layer.stride) == (2, 2) and tuple(layer.padding) == (1, 1) and layer.groups == 1
            and tuple(layer.dilation) == (1, 1) and layer.padding_mode == "zeros" and layer.out_channels % 256 == 0
            and layer.in_channels % 128 == 0 and 9 * layer.in_channels <= 16384 and x.shape[1] == layer.in_channels)


def conv_strided3x3(layer, x):
    """layer(x) for a 3x3 / stride 2 / padding 1 convolution as gather + GEMM: the 3 x 3 windows of the zero padded NHWC map are
    gathered once ([B, ho, wo, 3, 3, C], a strided view copied to rows), then ds_linear with the bias in its epilogue."""
    from . import _native
    b, c, h, w = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous()          # [B, h + 2, w + 2, C]
    sb, sh, sw, sc = xp.stride()
    rows = xp.as_strided((b, ho, wo, 3, 3, c), (sb, 2 * sh, 2 * sw, sh, sw, sc)).reshape(b * ho * wo, 9 * c)
    y = _native.linear(rows, _gemm_weight(layer, x, 9 * c), layer.bias)
    return y.view(b, ho, wo, layer.out_channels).permute(0, 3, 1, 2)
PREPROCESS_HIP = os.environ.get("DS_PREPROCESS", "1") != "0"      # A/B switch: ds_preprocess_bicubic vs the torch chain


def conv_module(layer, x):
    """layer(x) for the library convolutions of the decoders (1x1, strided 3x3, ConvTranspose2d with a bias).  torch adds a
    convolution's bias with a separate strided broadcast kernel on ROCm; for half-precision channels_last activations the
    convolution runs WITHOUT its bias and ds_bias_act_nhwc adds it in place (a vectorised pass at HBM rate).  Everything else
    -- float32, CPU, no bias, circular padding, other modules -- is the plain call.
    Round 5: 1x1 convolutions and kernel == stride transposed convolutions that fill the chip are in-tree GEMMs with the bias in
    the epilogue (conv1x1, _native.conv_transpose_shuffle); what is left for the libraries is the strided 3x3 of act_postprocess4
    and maps too small for 256 x 256 tiles."""
    if conv1x1_hip_ok(layer, x):
        return conv1x1(layer, x)
    if strided3x3_hip_ok(layer, x):
        return conv_strided3x3(layer, x)
    if conv_transpose_hip_ok(layer, x):
        from . import _native
        return _native.conv_transpose_shuffle(layer, x)
    if (CONV_BIAS_HIP and half_on_gpu(x) and x.dim() == 4 and getattr(layer, "bias", None) is not None
            and layer.out_channels % 8 == 0 and getattr(layer, "padding_mode", "zeros") == "zeros"):
        if type(layer) is nn.Conv2d:
            y = layer._conv_forward(x, layer.weight, None)
        elif type(layer) is nn.ConvTranspose2d:
            y = F.conv_transpose2d(x, layer.weight, None, layer.stride, layer.padding, layer.output_padding, layer.groups, layer.dilation)
        else:
            return layer(x)
        if y.is_contiguous(memory_format=torch.channels_last):
            from . import _native
            return _native.bias_act(y, layer.bias, relu=False)
        return y + layer.bias.view(1, -1, 1, 1)
    return layer(x)


def residual_conv_unit(conv1, conv2, x, skip=None):
    """[skip +] ( conv2(relu(conv1(relu(x)))) + x ): ResidualConvUnit_custom (dmidas/blocks.py:352-377) / ResidualConvUnit
    (ddepth_anything_v2/.../util/blocks.py:56-85) and, with `skip`, the add of the fusion block around it (:427 / :135).
    Half precision on the GPU: the library convolutions run WITHOUT their bias and ds_bias_act_nhwc does bias + ReLU and
    bias + residual (+ skip) in one pass each.  Everything else: the plain torch sequence."""
    if (half_on_gpu(x) and x.shape[1] % 8 == 0 and conv1.bias is not None
            and conv2.bias is not None and conv1.out_channels % 8 == 0):
        from . import _native
        xc = x.contiguous(memory_format=torch.channels_last)
        if (conv3x3_hip_ok(conv1, xc) and conv3x3_hip_ok(conv2, xc) and conv1.out_channels == conv2.in_channels
                and conv2.out_channels % 256 == 0):
            # both convolutions as in-tree implicit GEMMs, the element-wise tails in their epilogues: 3 launches per unit
            # (round 6: the ReLU in front of conv1 is taken on the MFMA fragments inside the kernel: no clamp pass over x)
            sk = None if skip is None else skip.contiguous(memory_format=torch.channels_last)
            if CONV_RELU_IN and conv1.out_channels % 256 == 0:
                a = _native.conv3x3(conv1, xc, relu=True, relu_in=True)
            else:
                a = _native.conv3x3(conv1, F.relu(xc), relu=True)
            return _native.conv3x3(conv2, a, relu=False, res1=xc, res2=sk)
        c1 = conv1._conv_forward(F.relu(xc), conv1.weight, None)          # honours padding_mode (TILING_MODE: circular)
        if c1.is_contiguous(memory_format=torch.channels_last):
            a = _native.bias_act(c1, conv1.bias, relu=True)
            c2 = conv2._conv_forward(a, conv2.weight, None)
            if c2.is_contiguous(memory_format=torch.channels_last) and c2.shape == xc.shape:
                sk = None if skip is None else skip.contiguous(memory_format=torch.channels_last)
                return _native.bias_act(c2, conv2.bias, relu=False, res1=xc, res2=sk)
            out = c2 + conv2.bias.view(1, -1, 1, 1) + x
            return out if skip is None else skip + out
        out = conv2(F.relu(c1 + conv1.bias.view(1, -1, 1, 1))) + x
        return out if skip is None else skip + out
    out = conv1(F.relu(x))
    out = conv2(F.relu(out))
    out = out + x
    return out if skip is None else skip + out


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (dinov2_layers/mlp.py:20-41; timm Mlp has the same parameter names)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return linear(linear(x, self.fc1.weight, self.fc1.bias, gelu=True), self.fc2.weight, self.fc2.bias)


def count_encoder_flops(depth, n, dim, mlp_ratio=4.0):
    """Algorithmic FLOPs of `depth` blocks on n tokens: 2*(4 + 2*mlp_ratio)*n*dim^2 linear + 4*n^2*dim attention."""
    lin = 2.0 * (4.0 + 2.0 * mlp_ratio) * n * dim * dim
    att = 4.0 * n * n * dim
    return depth * (lin + att)
