"""ZoeDepth (model ids 7 zoedepth_n, 8 zoedepth_k, 9 zoedepth_nk) on the MI355X DPT engine.

Reference: dzoedepth/models/zoedepth/zoedepth_v1.py (single metric head), dzoedepth/models/zoedepth_nk/zoedepth_nk_v1.py
(two heads + router), dzoedepth/models/base_models/midas.py (MidasCore: the relative-depth DPT whose decoder features the
heads read), dzoedepth/models/layers/{attractor,dist_layers,localbins_layers,patch_transformer}.py and
dzoedepth/models/depth_model.py (padding + flip augmentation of infer).  How the funnel uses it:
src/depthmap_generation.py:196-209 (construction from the json configs), :443-452 (estimatezoedepth), :266-272 (zoedepth_n
stays float32; k / nk run in half).

The reference obtains the DPT core from torch.hub ("semjon00/MiDaS", DPT_BEiT_L_384) -- the same architecture as its own
vendored dmidas.DPTDepthModel(backbone="beitl16_384"), which is what runs here (dmidas/dpt_depth.py: fused attention,
fused residual+LayerNorm, ...).  Instead of forward hooks the core hands out its decoder features directly.
State-dict names are the reference's, so ZoeD_M12_{N,K,NK}.pt load by name.

Reference quirks that are reproduced on purpose:
  * the attractor functions are called with their DEFAULT strength (alpha=300, gamma=2): the layers store the configured
    attractor_alpha / attractor_gamma but never pass them on (attractor.py:124-131,195-202);
  * ZoeDepthNK constructs its attractor layers with (in_features, n_attractors[i]) positionally, i.e. as n_bins, so every
    level has the default 16 attractors (zoedepth_nk_v1.py:143-152);
  * infer mode carries no dataset config, so min_depth / max_depth of the single-head models are the constructor defaults
    1e-3 / 10 also for the 'kitti' (normed bins) version.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from dmidas.dpt_depth import DPTDepthModel, _constrain, midas_net_size
from src import vit_mi355x as vm

_MIDAS_BACKBONES = {"DPT_BEiT_L_384": "beitl16_384", "DPT_BEiT_L_512": "beitl16_512", "DPT_BEiT_B_384": "beitb16_384",
                    "DPT_Large": "vitl16_384", "DPT_Hybrid": "vitb_rn50_384"}
FEATURE_NAMES = ('out_conv', 'l4_rn', 'r4', 'r3', 'r2', 'r1')       # midas.py:173


class MidasCore(nn.Module):
    """midas.py:172-331.  `.core` is the DPT network (parameter prefix core.core.* like the reference)."""

    def __init__(self, midas, img_size=(384, 384), keep_aspect_ratio=True):
        super().__init__()
        self.core = midas
        self.output_channels = (256,) * 5                                # midas.py:368-376 (every DPT variant)
        self.net_h, self.net_w = int(img_size[0]), int(img_size[1])       # PrepForMidas: img_size = (H, W)
        self.keep_aspect_ratio = keep_aspect_ratio

    def set_net_size(self, width, height):
        """What estimatezoedepth does by poking the resizer's private fields (src/depthmap_generation.py:448-449)."""
        self.net_w, self.net_h = int(width), int(height)

    def prep(self, x):
        """PrepForMidas (midas.py:155-169): bilinear align_corners=True resize ('minimal', multiple of 32), then
        (x - 0.5) / 0.5."""
        h, w = x.shape[-2:]
        if self.keep_aspect_ratio:
            new_w, new_h = midas_net_size(w, h, self.net_w, self.net_h, "minimal", 32)
        else:                                                    # each axis scaled on its own (midas.py:121-122,148-150)
            new_w, new_h = _constrain(self.net_w / w * w, 32), _constrain(self.net_h / h * h, 32)
        x = F.interpolate(x, (int(new_h), int(new_w)), mode='bilinear', align_corners=True)
        return (x - 0.5) / 0.5

    def forward(self, x):
        """-> (rel_depth [B,h,w], [features in FEATURE_NAMES order])   (midas.py:268-287 with return_rel_depth=True)"""
        feats = {}
        rel = self.core(self.prep(x), feats)
        return rel, [feats[k] for k in FEATURE_NAMES]


def _mlp1x1(cin, hidden, cout, last):
    layers = [nn.Conv2d(cin, hidden, 1, 1, 0), nn.ReLU(inplace=True), nn.Conv2d(hidden, cout, 1, 1, 0)]
    if last is not None:
        layers.append(last)
    return nn.Sequential(*layers)


class SeedBinRegressor(nn.Module):
    """localbins_layers.py:29-69 (bounded) / :72-100 (unnormed=True: softplus centres)."""

    def __init__(self, in_features, n_bins=16, mlp_dim=256, min_depth=1e-3, max_depth=10, unnormed=False):
        super().__init__()
        self.unnormed, self.min_depth, self.max_depth = unnormed, min_depth, max_depth
        self._net = _mlp1x1(in_features, mlp_dim, n_bins, nn.Softplus() if unnormed else nn.ReLU(inplace=True))

    def forward(self, x):
        b = self._net(x)
        if self.unnormed:
            return b, b
        b = b + 1e-3
        widths_normed = b / b.sum(dim=1, keepdim=True)
        widths = (self.max_depth - self.min_depth) * widths_normed
        widths = F.pad(widths, (0, 0, 0, 0, 1, 0), mode='constant', value=self.min_depth)
        edges = torch.cumsum(widths, dim=1)
        return widths_normed, 0.5 * (edges[:, :-1] + edges[:, 1:])


class Projector(nn.Module):
    """localbins_layers.py:103-123."""

    def __init__(self, in_features, out_features, mlp_dim=128):
        super().__init__()
        self._net = _mlp1x1(in_features, mlp_dim, out_features, None)

    def forward(self, x):
        return self._net(x)


def _attract(dx, kind):
    """attractor.py:29-58 at the default strength the reference actually runs with (see the module docstring)."""
    alpha, gamma = 300.0, 2
    if kind == 'exp':
        return torch.exp(-alpha * (torch.abs(dx) ** gamma)) * dx
    return dx.div(1 + alpha * dx.pow(gamma))


class AttractorLayer(nn.Module):
    """attractor.py:61-138 (bounded centres) / :141-208 (unnormed=True)."""

    def __init__(self, in_features, n_bins, n_attractors=16, mlp_dim=128, min_depth=1e-3, max_depth=10, alpha=300, gamma=2,
                 kind='sum', attractor_type='exp', memory_efficient=False, unnormed=False):
        super().__init__()
        self.n_attractors, self.n_bins = n_attractors, n_bins
        self.min_depth, self.max_depth = min_depth, max_depth
        self.alpha, self.gamma, self.kind, self.attractor_type = alpha, gamma, kind, attractor_type     # stored, unused
        self.memory_efficient, self.unnormed = memory_efficient, unnormed
        if unnormed:
            self._net = _mlp1x1(in_features, mlp_dim, n_attractors, nn.Softplus())
        else:
            self._net = _mlp1x1(in_features, mlp_dim, n_attractors * 2, nn.ReLU(inplace=True))

    def forward(self, x, b_prev, prev_b_embedding=None, interpolate=True):
        if prev_b_embedding is not None:
            if interpolate:
                prev_b_embedding = F.interpolate(prev_b_embedding, x.shape[-2:], mode='bilinear', align_corners=True)
            x = x + prev_b_embedding
        a = self._net(x)
        n, _, h, w = a.shape
        if not self.unnormed:                                   # :103-107: the normalised pair is computed and dropped
            a = (a + 1e-3).view(n, self.n_attractors, 2, h, w)[:, :, 0]
        centers = F.interpolate(b_prev, (h, w), mode='bilinear', align_corners=True)
        if not self.memory_efficient:
            d = _attract(a.unsqueeze(2) - centers.unsqueeze(1), self.attractor_type)
            delta = d.mean(dim=1) if self.kind == 'mean' else d.sum(dim=1)
        else:
            delta = torch.zeros_like(centers)
            for i in range(self.n_attractors):
                delta += _attract(a[:, i].unsqueeze(1) - centers, self.attractor_type)
            if self.kind == 'mean':
                delta = delta / self.n_attractors
        new = centers + delta
        if self.unnormed:
            return new, new
        scaled = (self.max_depth - self.min_depth) * new + self.min_depth
        scaled, _ = torch.sort(scaled, dim=1)
        return new, torch.clip(scaled, self.min_depth, self.max_depth)


class _LogBinomial(nn.Module):
    """dist_layers.py:29-70: softmax over the log binomial pmf (Stirling) at temperature t."""

    def __init__(self, n_classes):
        super().__init__()
        self.K = n_classes
        self.register_buffer('k_idx', torch.arange(0, n_classes).view(1, -1, 1, 1))
        self.register_buffer('K_minus_1', torch.Tensor([n_classes - 1]).view(1, -1, 1, 1))

    def forward(self, x, t, eps=1e-4):
        if x.ndim == 3:
            x = x.unsqueeze(1)
        one_minus_x = torch.clamp(1 - x, eps, 1)
        x = torch.clamp(x, eps, 1)
        n, k = self.K_minus_1 + 1e-7, self.k_idx + 1e-7                   # log_binom (:29-33)
        lb = n * torch.log(n) - k * torch.log(k) - (n - k) * torch.log(n - k + 1e-7)
        y = lb + self.k_idx * torch.log(x) + (self.K - 1 - self.k_idx) * torch.log(one_minus_x)
        return torch.softmax(y / t, dim=1)


class ConditionalLogBinomial(nn.Module):
    """dist_layers.py:73-121."""

    def __init__(self, in_features, condition_dim, n_classes=256, bottleneck_factor=2, p_eps=1e-4, max_temp=50, min_temp=1e-7):
        super().__init__()
        self.p_eps, self.max_temp, self.min_temp = p_eps, max_temp, min_temp
        self.log_binomial_transform = _LogBinomial(n_classes)
        hidden = (in_features + condition_dim) // bottleneck_factor
        self.mlp = nn.Sequential(nn.Conv2d(in_features + condition_dim, hidden, 1, 1, 0), nn.GELU(),
                                 nn.Conv2d(hidden, 4, 1, 1, 0), nn.Softplus())

    def forward(self, x, cond):
        pt = self.mlp(torch.concat((x, cond), dim=1))
        p, t = pt[:, :2] + self.p_eps, pt[:, 2:] + self.p_eps
        p = p[:, 0] / (p[:, 0] + p[:, 1])
        t = (t[:, 0] / (t[:, 0] + t[:, 1])).unsqueeze(1)
        t = (self.max_temp - self.min_temp) * t + self.min_temp
        return self.log_binomial_transform(p, t)


class PatchTransformerEncoder(nn.Module):
    """patch_transformer.py:29-92: the router's tiny encoder (4 layers, d=128, 4 heads) on the 1/32 bottleneck."""

    def __init__(self, in_channels, patch_size=10, embedding_dim=128, num_heads=4, use_class_token=False):
        super().__init__()
        self.use_class_token = use_class_token
        layer = nn.TransformerEncoderLayer(embedding_dim, num_heads, dim_feedforward=1024)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=4)
        self.embedding_convPxP = nn.Conv2d(in_channels, embedding_dim, kernel_size=patch_size, stride=patch_size, padding=0)

    @staticmethod
    def positional_encoding_1d(length, batch, dim, device):
        pos = torch.arange(0, length, dtype=torch.float32, device=device).unsqueeze(1)
        idx = torch.arange(0, dim, 2, dtype=torch.float32, device=device).unsqueeze(0)
        div = torch.exp(idx * (-torch.log(torch.tensor(10000.0, device=device)) / dim))
        enc = pos * div
        enc = torch.cat([torch.sin(enc), torch.cos(enc)], dim=1)
        return enc.unsqueeze(1).repeat(1, batch, 1)

    def forward(self, x):
        e = self.embedding_convPxP(x).flatten(2)
        if self.use_class_token:
            e = F.pad(e, (1, 0))
        e = e.permute(2, 0, 1)
        s, n, d = e.shape
        e = e + self.positional_encoding_1d(s, n, d, e.device).to(dtype=e.dtype)
        return self.transformer_encoder(e)


class _DepthModel(nn.Module):
    """depth_model.py:35-152: the augmented inference interface."""

    def _infer(self, x):
        return self(x)['metric_depth']

    def _infer_with_pad_aug(self, x, pad_input=True, fh=3, fw=3, upsampling_mode='bicubic', padding_mode="reflect"):
        assert x.dim() == 4 and x.shape[1] == 3
        pad_h = pad_w = 0
        if pad_input:
            pad_h = int(np.sqrt(x.shape[2] / 2) * fh)
            pad_w = int(np.sqrt(x.shape[3] / 2) * fw)
            padding = [pad_w, pad_w] + ([pad_h, pad_h] if pad_h > 0 else [])
            x = F.pad(x, padding, mode=padding_mode)
        out = self._infer(x)
        if out.shape[-2:] != x.shape[-2:]:
            out = F.interpolate(out, size=(x.shape[2], x.shape[3]), mode=upsampling_mode, align_corners=False)
        if pad_h > 0:
            out = out[:, :, pad_h:-pad_h, :]
        if pad_w > 0:
            out = out[:, :, :, pad_w:-pad_w]
        return out

    def infer(self, x, pad_input=True, with_flip_aug=True):
        out = self._infer_with_pad_aug(x, pad_input=pad_input)
        if not with_flip_aug:
            return out
        flipped = self._infer_with_pad_aug(torch.flip(x, dims=[3]), pad_input=pad_input)
        return (out + torch.flip(flipped, dims=[3])) / 2

    @torch.no_grad()
    def infer_batch(self, images_u8, net_width, net_height):
        """estimatezoedepth (src/depthmap_generation.py:443-452) for a device-resident uint8 batch [B,H,W,3] (RGB as PIL
        gives it: this family is fed without the channel swap of the MiDaS path): ToTensor scaling, net size override,
        infer with padding + flip augmentation.  Returns [B,H,W] in the model's dtype converted to float32."""
        p = next(self.parameters())
        x = images_u8.to(p.device).permute(0, 3, 1, 2).to(torch.float32).div(255).to(p.dtype)   # ToTensor: u8 / 255
        self.core.set_net_size(net_width, net_height)
        return self.infer(x)[:, 0].float()


def _layer_types(bin_centers_type):
    if bin_centers_type not in ("normed", "softplus", "hybrid1", "hybrid2"):
        raise ValueError("bin_centers_type should be one of 'normed', 'softplus', 'hybrid1', 'hybrid2'")
    seed_unnormed = bin_centers_type in ("softplus", "hybrid2")
    attractor_unnormed = bin_centers_type in ("softplus", "hybrid1")
    return seed_unnormed, attractor_unnormed


class ZoeDepth(_DepthModel):
    """zoedepth_v1.py:38-206."""

    def __init__(self, core, n_bins=64, bin_centers_type="softplus", bin_embedding_dim=128, min_depth=1e-3, max_depth=10,
                 n_attractors=(16, 8, 4, 1), attractor_alpha=300, attractor_gamma=2, attractor_kind='sum', attractor_type='exp',
                 min_temp=5, max_temp=50, inverse_midas=False, **_unused):
        super().__init__()
        self.core = core
        self.min_depth, self.max_depth, self.bin_centers_type, self.inverse_midas = min_depth, max_depth, bin_centers_type, inverse_midas
        seed_unnormed, attractor_unnormed = _layer_types(bin_centers_type)
        btlnck = core.output_channels[0]
        self.conv2 = nn.Conv2d(btlnck, btlnck, kernel_size=1, stride=1, padding=0)
        self.seed_bin_regressor = SeedBinRegressor(btlnck, n_bins=n_bins, min_depth=min_depth, max_depth=max_depth, unnormed=seed_unnormed)
        self.seed_projector = Projector(btlnck, bin_embedding_dim)
        self.projectors = nn.ModuleList([Projector(c, bin_embedding_dim) for c in core.output_channels[1:]])
        self.attractors = nn.ModuleList([
            AttractorLayer(bin_embedding_dim, n_bins, n_attractors=n_attractors[i], min_depth=min_depth, max_depth=max_depth,
                           alpha=attractor_alpha, gamma=attractor_gamma, kind=attractor_kind, attractor_type=attractor_type,
                           unnormed=attractor_unnormed)
            for i in range(len(core.output_channels) - 1)])
        self.conditional_log_binomial = ConditionalLogBinomial(32 + 1, bin_embedding_dim, n_classes=n_bins, min_temp=min_temp,
                                                               max_temp=max_temp)

    @vm.deterministic_forward
    def forward(self, x, return_final_centers=False, return_probs=False):
        rel_depth, feats = self.core(x)
        outconv, btlnck, blocks = feats[0], feats[1], feats[2:]
        x = self.conv2(btlnck)
        _, seed_centers = self.seed_bin_regressor(x)
        if self.bin_centers_type in ('normed', 'hybrid2'):
            b_prev = (seed_centers - self.min_depth) / (self.max_depth - self.min_depth)
        else:
            b_prev = seed_centers
        prev_embedding = self.seed_projector(x)
        for projector, attractor, blk in zip(self.projectors, self.attractors, blocks):
            b_embedding = projector(blk)
            b_prev, b_centers = attractor(b_embedding, b_prev, prev_embedding, interpolate=True)
            prev_embedding = b_embedding
        if self.inverse_midas:
            rel_depth = 1.0 / (rel_depth + 1e-6)
            rel_depth = (rel_depth - rel_depth.min()) / (rel_depth.max() - rel_depth.min())
        rel_cond = F.interpolate(rel_depth.unsqueeze(1), size=outconv.shape[2:], mode='bilinear', align_corners=True)
        last = torch.cat([outconv, rel_cond], dim=1)
        b_embedding = F.interpolate(b_embedding, last.shape[-2:], mode='bilinear', align_corners=True)
        probs = self.conditional_log_binomial(last, b_embedding)
        b_centers = F.interpolate(b_centers, probs.shape[-2:], mode='bilinear', align_corners=True)
        output = dict(metric_depth=torch.sum(probs * b_centers, dim=1, keepdim=True))
        if return_final_centers or return_probs:
            output['bin_centers'] = b_centers
        if return_probs:
            output['probs'] = probs
        return output


class ZoeDepthNK(_DepthModel):
    """zoedepth_nk_v1.py:40-231: one bottleneck router (patch transformer + MLP) picks the 'nyu' or the 'kitti' head for
    the whole batch."""

    def __init__(self, core, bin_conf, bin_centers_type="softplus", bin_embedding_dim=128, n_attractors=(16, 8, 4, 1),
                 attractor_alpha=300, attractor_gamma=2, attractor_kind='sum', attractor_type='exp', min_temp=5, max_temp=50,
                 memory_efficient=False, inverse_midas=False, **_unused):
        super().__init__()
        self.core = core
        self.bin_conf = [dict(c) for c in bin_conf]
        self.bin_centers_type, self.inverse_midas = bin_centers_type, inverse_midas
        seed_unnormed, attractor_unnormed = _layer_types(bin_centers_type)
        btlnck = core.output_channels[0]
        self.conv2 = nn.Conv2d(btlnck, btlnck, kernel_size=1, stride=1, padding=0)
        self.patch_transformer = PatchTransformerEncoder(btlnck, 1, 128, use_class_token=True)
        self.mlp_classifier = nn.Sequential(nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 2))
        self.seed_bin_regressors = nn.ModuleDict({
            c['name']: SeedBinRegressor(btlnck, c['n_bins'], mlp_dim=bin_embedding_dim // 2, min_depth=c['min_depth'],
                                        max_depth=c['max_depth'], unnormed=seed_unnormed) for c in self.bin_conf})
        self.seed_projector = Projector(btlnck, bin_embedding_dim, mlp_dim=bin_embedding_dim // 2)
        self.projectors = nn.ModuleList([Projector(c, bin_embedding_dim, mlp_dim=bin_embedding_dim // 2)
                                         for c in core.output_channels[1:]])
        # positional (in_features, n_bins := n_attractors[i]); n_attractors keeps its default 16 (reference quirk, :143-152)
        self.attractors = nn.ModuleDict({
            c['name']: nn.ModuleList([
                AttractorLayer(bin_embedding_dim, n_attractors[i], mlp_dim=bin_embedding_dim, alpha=attractor_alpha,
                               gamma=attractor_gamma, kind=attractor_kind, attractor_type=attractor_type,
                               memory_efficient=memory_efficient, min_depth=c['min_depth'], max_depth=c['max_depth'],
                               unnormed=attractor_unnormed)
                for i in range(len(n_attractors))]) for c in self.bin_conf})
        self.conditional_log_binomial = nn.ModuleDict({
            c['name']: ConditionalLogBinomial(32, bin_embedding_dim, c['n_bins'], bottleneck_factor=4, min_temp=min_temp,
                                              max_temp=max_temp) for c in self.bin_conf})

    @vm.deterministic_forward
    def forward(self, x, return_final_centers=False, return_probs=False):
        rel_depth, feats = self.core(x)
        outconv, btlnck, blocks = feats[0], feats[1], feats[2:]
        x = self.conv2(btlnck)
        embedding = self.patch_transformer(x)[0]
        domain_logits = self.mlp_classifier(embedding)
        vote = torch.softmax(domain_logits.sum(dim=0, keepdim=True), dim=-1)
        name = ["nyu", "kitti"][int(torch.argmax(vote, dim=-1).squeeze().item())]
        conf = [c for c in self.bin_conf if c['name'] == name]
        if not conf:
            raise ValueError(f"bin_conf_name {name} not found in bin_confs")
        conf = conf[0]
        _, seed_centers = self.seed_bin_regressors[name](x)
        if self.bin_centers_type in ('normed', 'hybrid2'):
            b_prev = (seed_centers - conf['min_depth']) / (conf['max_depth'] - conf['min_depth'])
        else:
            b_prev = seed_centers
        prev_embedding = self.seed_projector(x)
        for projector, attractor, blk in zip(self.projectors, self.attractors[name], blocks):
            b_embedding = projector(blk)
            b_prev, b_centers = attractor(b_embedding, b_prev, prev_embedding, interpolate=True)
            prev_embedding = b_embedding
        b_centers = F.interpolate(b_centers, outconv.shape[-2:], mode='bilinear', align_corners=True)
        b_embedding = F.interpolate(b_embedding, outconv.shape[-2:], mode='bilinear', align_corners=True)
        probs = self.conditional_log_binomial[name](outconv, b_embedding)
        output = dict(domain_logits=domain_logits, metric_depth=torch.sum(probs * b_centers, dim=1, keepdim=True))
        if return_final_centers or return_probs:
            output['bin_centers'] = b_centers
        if return_probs:
            output['probs'] = probs
        return output


# the "model" + "infer" sections of the reference's json configs (dzoedepth/models/zoedepth/config_zoedepth.json,
# config_zoedepth_kitti.json, dzoedepth/models/zoedepth_nk/config_zoedepth_nk.json) as get_config(..., "infer") merges them
_COMMON = dict(bin_embedding_dim=128, n_attractors=[16, 8, 4, 1], attractor_alpha=1000, attractor_gamma=2, attractor_kind="mean",
               attractor_type="inv", midas_model_type="DPT_BEiT_L_384", min_temp=0.0212, max_temp=50.0, memory_efficient=True)
CONFIGS = {
    "zoedepth_n": dict(_COMMON, n_bins=64, bin_centers_type="softplus", inverse_midas=False, img_size=[384, 512], force_keep_ar=True,
                       checkpoint="ZoeD_M12_N.pt"),
    "zoedepth_k": dict(_COMMON, n_bins=64, bin_centers_type="normed", inverse_midas=False, img_size=[384, 768], force_keep_ar=True,
                       checkpoint="ZoeD_M12_K.pt"),
    "zoedepth_nk": dict(_COMMON, bin_centers_type="softplus", img_size=[384, 512], force_keep_ar=True, checkpoint="ZoeD_M12_NK.pt",
                        bin_conf=[dict(name="nyu", n_bins=64, min_depth=1e-3, max_depth=10.0),
                                  dict(name="kitti", n_bins=64, min_depth=1e-3, max_depth=80.0)]),
}


def build_zoedepth(kind, **overrides):
    """The model src/depthmap_generation.py:196-209 builds for ids 7 ('zoedepth_n'), 8 ('zoedepth_k'), 9 ('zoedepth_nk'),
    without weights.  Returns (model, checkpoint file name)."""
    cfg = dict(CONFIGS[kind], **overrides)
    midas = DPTDepthModel(path=None, backbone=_MIDAS_BACKBONES[cfg.pop("midas_model_type")], non_negative=True)
    core = MidasCore(midas, img_size=cfg.pop("img_size"), keep_aspect_ratio=cfg.pop("force_keep_ar"))
    ckpt = cfg.pop("checkpoint")
    model = ZoeDepthNK(core, **cfg) if "bin_conf" in cfg else ZoeDepth(core, **cfg)
    return model, ckpt
