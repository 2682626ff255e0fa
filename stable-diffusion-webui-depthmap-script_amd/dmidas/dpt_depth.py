"""MiDaS 3.x DPT depth model (BEiT, ViT-L and ViT-B/ResNet-50 hybrid backbones), MI355X-first.

Reference: dmidas/dpt_depth.py (DPT :31-139, DPTDepthModel :142-166), dmidas/blocks.py (_make_scratch, Interpolate,
ResidualConvUnit_custom :322-377, FeatureFusionBlock_custom :382-441), and its caller estimatemidas
(src/depthmap_generation.py:455-499) + Resize.get_size (dmidas/transforms.py:105-160).
Checkpoint key names are the reference's (pretrained.model.*, pretrained.act_postprocessN.*, scratch.*).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm

from .backbones.beit import make_beit
from .backbones.vit import make_vit

_HOOKS = {"beitl16_512": [5, 11, 17, 23], "beitl16_384": [5, 11, 17, 23], "beitb16_384": [2, 5, 8, 11],
          "vitl16_384": [5, 11, 17, 23], "vitb_rn50_384": [0, 1, 8, 11]}


class ResidualConvUnit_custom(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, x):               # blocks.py:352-377, activation ReLU(False), bn False
        return vm.residual_conv_unit(self.conv1, self.conv2, x)


class FeatureFusionBlock_custom(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.resConfUnit1 = ResidualConvUnit_custom(features)
        self.resConfUnit2 = ResidualConvUnit_custom(features)

    def forward(self, *xs, size=None):  # blocks.py:412-441, align_corners=True
        output = xs[0]
        if len(xs) == 2:                # skip add fused into the unit's last element-wise pass
            output = vm.residual_conv_unit(self.resConfUnit1.conv1, self.resConfUnit1.conv2, xs[1], skip=output)
        output = self.resConfUnit2(output)
        # 1x1 out_conv and bilinear interpolation commute (linear, weights sum to one): conv first, on 4x fewer pixels
        output = vm.conv_module(self.out_conv, output)
        if size is None:
            return vm.interpolate_bilinear(output, scale_factor=2, align_corners=True)
        return vm.interpolate_bilinear(output, size=tuple(size), align_corners=True)


class Interpolate(nn.Module):           # blocks.py:213-244 (no parameters; keeps the Sequential indices of the head)
    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners

    def forward(self, x):
        if self.mode == "bilinear":
            return vm.interpolate_bilinear(x, scale_factor=self.scale_factor, align_corners=bool(self.align_corners))
        return F.interpolate(x, scale_factor=self.scale_factor, mode=self.mode, align_corners=self.align_corners)


class DPT(nn.Module):
    def __init__(self, head, features=256, backbone="beitl16_512", readout="project", channels_last=False, use_bn=False, **kwargs):
        super().__init__()
        if backbone not in _HOOKS:
            raise NotImplementedError(f"DPT backbone '{backbone}' is not built (built: {sorted(_HOOKS)})")
        if readout != "project" or use_bn:
            raise NotImplementedError("only readout='project', use_bn=False (what src/depthmap_generation.py:129-176 uses)")
        self.channels_last = channels_last
        self.size_routed = backbone == "vitb_rn50_384"     # its ResNetV2 stem is library code: see vm.size_routed
        make = make_beit if backbone.startswith("beit") else make_vit
        self.pretrained, in_shape = make(backbone, _HOOKS[backbone])
        scratch = nn.Module()
        for i in range(4):
            setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(in_shape[i], features, kernel_size=3, stride=1, padding=1, bias=False))
        for i in range(1, 5):
            setattr(scratch, f"refinenet{i}", FeatureFusionBlock_custom(features))
        scratch.output_conv = head
        self.scratch = scratch

    def forward(self, x, features=None):               # dpt_depth.py:110-139
        """features: None, or a dict that receives what ZoeDepth's MidasCore taps with forward hooks
        (dzoedepth/models/base_models/midas.py:307-331): 'l4_rn', 'r4'..'r1' and 'out_conv' (the 32-channel activation
        after the head's second convolution + ReLU)."""
        if getattr(self, "size_routed", False):
            with vm.size_routed():
                return self._forward(x, features)
        return self._forward(x, features)

    def _forward(self, x, features=None):
        l1, l2, l3, l4 = self.pretrained(x)
        s = self.scratch
        l1, l2, l3, l4 = vm.conv2d(s.layer1_rn, l1), vm.conv2d(s.layer2_rn, l2), vm.conv2d(s.layer3_rn, l3), vm.conv2d(s.layer4_rn, l4)
        path_4 = s.refinenet4(l4, size=l3.shape[2:])
        path_3 = s.refinenet3(path_4, l3, size=l2.shape[2:])
        path_2 = s.refinenet2(path_3, l2, size=l1.shape[2:])
        path_1 = s.refinenet1(path_2, l1)
        head = s.output_conv
        if features is not None:
            features.update(l4_rn=l4, r4=path_4, r3=path_3, r2=path_2, r1=path_1)
            y = path_1
            for i, layer in enumerate(head):
                y = layer(y)
                if i == 3:
                    features['out_conv'] = y
            return y
        if (vm.half_on_gpu(path_1) and isinstance(head[1], Interpolate)
                and head[1].mode == "bilinear" and head[1].align_corners and tuple(head[2].weight.shape) == (32, 128, 3, 3)
                and head[2].padding_mode == 'zeros'):          # TILING_MODE makes the convolutions circular: library path
            # upsample x2 -> conv3x3 128->32 -> ReLU -> conv1x1 -> ReLU in one MFMA kernel (ds_dpt_head_tail)
            from src import _native
            # the head's first convolution (256 -> 128): the in-tree implicit GEMM on 256 x 128 tiles where the launch fills the chip
            # (round 4; MIOpen's heuristic choice for it varied from 1.75 to 2.1 ms per 32 x 256^2 between boxes), else the library
            y = vm.conv2d(head[0], path_1) if vm.conv3x3_hip_ok(head[0], path_1) else vm.conv_module(head[0], path_1)
            size = (int(y.shape[2] * head[1].scale_factor), int(y.shape[3] * head[1].scale_factor))
            return _native.dpt_head_tail(y, size, head[2], head[4], relu_out=isinstance(head[5], nn.ReLU))
        return head(path_1)


def _constrain(x, multiple, min_val=0, max_val=None):
    y = int(np.round(x / multiple) * multiple)
    if max_val is not None and y > max_val:
        y = int(np.floor(x / multiple) * multiple)
    if y < min_val:
        y = int(np.ceil(x / multiple) * multiple)
    return y


def midas_net_size(width, height, net_w, net_h, resize_method="minimal", multiple=32):
    """Resize.get_size with keep_aspect_ratio=True (dmidas/transforms.py:105-160)."""
    scale_h, scale_w = net_h / height, net_w / width
    if resize_method == "lower_bound":
        scale_h = scale_w = max(scale_w, scale_h)
    elif resize_method == "upper_bound":
        scale_h = scale_w = min(scale_w, scale_h)
    elif resize_method == "minimal":
        if abs(1 - scale_w) < abs(1 - scale_h):
            scale_h = scale_w
        else:
            scale_w = scale_h
    else:
        raise ValueError(f"resize_method {resize_method} not implemented")
    if resize_method == "lower_bound":
        return _constrain(scale_w * width, multiple, min_val=net_w), _constrain(scale_h * height, multiple, min_val=net_h)
    if resize_method == "upper_bound":
        return _constrain(scale_w * width, multiple, max_val=net_w), _constrain(scale_h * height, multiple, max_val=net_h)
    return _constrain(scale_w * width, multiple), _constrain(scale_h * height, multiple)


class DPTDepthModel(DPT):
    def __init__(self, path=None, non_negative=True, **kwargs):
        features = kwargs.get("features", 256)
        head_features_1 = kwargs.pop("head_features_1", features)
        head_features_2 = kwargs.pop("head_features_2", 32)
        head = nn.Sequential(
            nn.Conv2d(head_features_1, head_features_1 // 2, kernel_size=3, stride=1, padding=1),
            Interpolate(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(head_features_1 // 2, head_features_2, kernel_size=3, stride=1, padding=1),
            nn.ReLU(True),
            nn.Conv2d(head_features_2, 1, kernel_size=1, stride=1, padding=0),
            nn.ReLU(True) if non_negative else nn.Identity(),
            nn.Identity())
        super().__init__(head, **kwargs)
        if path is not None:
            self.load(path)

    def load(self, path):               # dmidas/base_model.py:5-16
        parameters = torch.load(path, map_location=torch.device('cpu'))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        self.load_state_dict(parameters, strict=False)     # timm's non-persistent buffers may or may not be in the file

    @vm.deterministic_forward
    def forward(self, x, features=None):
        return super().forward(x, features).squeeze(dim=1)

    @staticmethod
    def preprocess(images_u8, net_size=512, net_h=None, resize_mode="minimal", mean=0.5, std=0.5, dtype=None):
        """The transform chain of estimatemidas (src/depthmap_generation.py:457-476: Resize(keep_aspect_ratio, multiple of 32,
        INTER_CUBIC) -> NormalizeImage -> PrepareForNet) after get_raw_prediction's channel swap and /255 (:381), as tensor ops:
        uint8 [B,H,W,3] (RGB) -> float32 [B,3,nh,nw].  Pinned against the reference's own transform classes run with a numpy
        stand-in for cv2.resize (tests/golden/make_golden_transforms.py); cv2's arithmetic itself stays unpinned."""
        b, h, w, _ = images_u8.shape
        nw, nh = midas_net_size(w, h, int(net_size), int(net_size if net_h is None else net_h), resize_mode)
        if images_u8.is_cuda and vm.PREPROCESS_HIP and not vm.STOCK[0]:
            # one pass over the image bytes (ds_preprocess_bicubic): flip, / 255, bicubic, normalise, cast, channels_last
            from src import _native
            return _native.preprocess_bicubic(images_u8, (nh, nw), mean, std, flip=True, dtype=dtype)
        x = images_u8.flip(-1).permute(0, 3, 1, 2).float() / 255.0
        x = F.interpolate(x, size=(nh, nw), mode="bicubic", align_corners=False)
        x = (x - mean) / std
        return x if dtype is None else x.to(dtype)

    # ---- device-resident pre/post of estimatemidas (src/depthmap_generation.py:455-499) ------------------
    @torch.no_grad()
    def infer_batch(self, images_u8, net_size=512, resize_mode="minimal", mean=0.5, std=0.5, net_h=None):
        """uint8 [B,H,W,3] (RGB as the funnel hands it over) -> float32 [B,H,W] raw prediction, on the device.
        net_size is the requested network WIDTH, net_h the HEIGHT (default: the same) -- Resize gets both (:460-462), and with
        NET_SIZE_MATCH on a non-square image (core.py:177-181) they differ, which changes the 'minimal' scale choice.
        get_raw_prediction swaps R and B once (:381) and estimatemidas does not swap back, so the network sees BGR
        order -- reproduced.  cv2.INTER_CUBIC resize -> torch bicubic (same kernel a=-0.75, half-pixel centres; cv2 is
        not available here: unpinned); prediction upsampled with torch bicubic align_corners=False exactly as :484-489."""
        b, h, w, _ = images_u8.shape
        dtype = self.scratch.layer1_rn.weight.dtype
        x = self.preprocess(images_u8, net_size, net_h, resize_mode, mean, std, dtype=dtype)
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)      # (already so: the fused kernel writes channels_last)
        pred = self.forward(x)
        pred = F.interpolate(pred.unsqueeze(1), size=(h, w), mode="bicubic", align_corners=False).squeeze(1)
        return pred.float()
