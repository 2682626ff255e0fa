"""Depth-Anything-V2: DINOv2 encoder + DPT head, MI355X-first.

Reference: ddepth_anything_v2/depth_anything_v2/dpt.py (DPTHead :37-150, DepthAnythingV2 :153-221),
util/blocks.py (ResidualConvUnit :32-85, FeatureFusionBlock :88-148, _make_scratch :4-29),
util/transform.py (Resize.get_size :58-107), and its caller src/depthmap_generation.py:548-559.
Checkpoint key names are the reference's.  The decoder's convolutions run through MIOpen (channels_last on the GPU).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm

from .dinov2 import DINOv2


class ResidualConvUnit(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, x):               # blocks.py:56-85 (bn=False, activation ReLU(False))
        return vm.residual_conv_unit(self.conv1, self.conv2, x)


class FeatureFusionBlock(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)

    def forward(self, *xs, size=None):  # blocks.py:121-148 (align_corners=True)
        output = xs[0]
        if len(xs) == 2:                # skip add fused into the unit's last element-wise pass
            output = vm.residual_conv_unit(self.resConfUnit1.conv1, self.resConfUnit1.conv2, xs[1], skip=output)
        output = self.resConfUnit2(output)
        # the reference interpolates, then applies the 1x1 out_conv; both are linear and the bilinear weights sum to one,
        # so they commute (bias included): the conv runs on 4x fewer pixels and the upsample writes the final tensor
        output = vm.conv_module(self.out_conv, output)
        if size is None:
            return vm.interpolate_bilinear(output, scale_factor=2, align_corners=True)
        return vm.interpolate_bilinear(output, size=tuple(size), align_corners=True)


class DPTHead(nn.Module):
    def __init__(self, in_channels, features=256, use_bn=False, out_channels=(256, 512, 1024, 1024), use_clstoken=False):
        super().__init__()
        if use_bn or use_clstoken:
            raise NotImplementedError("use_bn / use_clstoken are never enabled by the reference's model table "
                                      "(src/depthmap_generation.py:243-247)")
        oc = list(out_channels)
        self.projects = nn.ModuleList([nn.Conv2d(in_channels, c, kernel_size=1) for c in oc])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(oc[0], oc[0], kernel_size=4, stride=4, padding=0),
            nn.ConvTranspose2d(oc[1], oc[1], kernel_size=2, stride=2, padding=0),
            nn.Identity(),
            nn.Conv2d(oc[3], oc[3], kernel_size=3, stride=2, padding=1)])
        scratch = nn.Module()
        for i in range(4):
            setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(oc[i], features, kernel_size=3, stride=1, padding=1, bias=False))
        for i in range(1, 5):
            setattr(scratch, f"refinenet{i}", FeatureFusionBlock(features))
        scratch.output_conv1 = nn.Conv2d(features, features // 2, kernel_size=3, stride=1, padding=1)
        scratch.output_conv2 = nn.Sequential(
            nn.Conv2d(features // 2, 32, kernel_size=3, stride=1, padding=1), nn.ReLU(True),
            nn.Conv2d(32, 1, kernel_size=1, stride=1, padding=0), nn.ReLU(True), nn.Identity())
        self.scratch = scratch

    def forward(self, out_features, patch_h, patch_w):          # dpt.py:117-150
        out = []
        for i, x in enumerate(out_features):
            x = x[0]
            # [B, ph*pw, C] IS the NHWC image: a channels_last view, not a permuted copy (reference: permute + reshape)
            x = x.reshape(x.shape[0], patch_h, patch_w, x.shape[-1]).permute(0, 3, 1, 2)
            if vm.half_on_gpu(x) and not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)     # (the tokens without their cls row: one copy, which a library convolution makes too)
            x = vm.conv_module(self.resize_layers[i], vm.conv_module(self.projects[i], x))
            out.append(x)
        l1, l2, l3, l4 = out
        s = self.scratch
        l1, l2, l3, l4 = vm.conv2d(s.layer1_rn, l1), vm.conv2d(s.layer2_rn, l2), vm.conv2d(s.layer3_rn, l3), vm.conv2d(s.layer4_rn, l4)
        path_4 = s.refinenet4(l4, size=l3.shape[2:])
        path_3 = s.refinenet3(path_4, l3, size=l2.shape[2:])
        path_2 = s.refinenet2(path_3, l2, size=l1.shape[2:])
        path_1 = s.refinenet1(path_2, l1)
        out = vm.conv2d(s.output_conv1, path_1)                # 256 -> 128: the in-tree implicit GEMM (256 x 128 tiles) where it fills the chip
        size = (int(patch_h * 14), int(patch_w * 14))
        if vm.half_on_gpu(out) and tuple(s.output_conv2[0].weight.shape) == (32, 128, 3, 3):
            # upsample -> conv3x3 128->32 -> ReLU -> conv1x1 -> ReLU in one MFMA kernel (ds_dpt_head_tail)
            from src import _native
            return _native.dpt_head_tail(out, size, s.output_conv2[0], s.output_conv2[2], relu_out=True)
        out = vm.interpolate_bilinear(out, size=size, align_corners=True)
        return s.output_conv2(out)


_LAYER_IDX = {'vits': [2, 5, 8, 11], 'vitb': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}


def _constrain(x, multiple, min_val):
    y = int(np.round(x / multiple) * multiple)
    if y < min_val:
        y = int(np.ceil(x / multiple) * multiple)
    return y


def lower_bound_size(width, height, target, multiple=14):
    """Resize.get_size with keep_aspect_ratio, 'lower_bound', ensure_multiple_of=14 (transform.py:58-107)."""
    scale_h, scale_w = target / height, target / width
    if scale_w > scale_h:
        scale_h = scale_w
    else:
        scale_w = scale_h
    return _constrain(scale_w * width, multiple, target), _constrain(scale_h * height, multiple, target)


class DepthAnythingV2(nn.Module):
    def __init__(self, encoder='vitl', features=256, out_channels=(256, 512, 1024, 1024), use_bn=False, use_clstoken=False):
        super().__init__()
        self.intermediate_layer_idx = dict(_LAYER_IDX)
        self.encoder = encoder
        self.pretrained = DINOv2(model_name=encoder)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn, out_channels=out_channels, use_clstoken=use_clstoken)

    @vm.deterministic_forward
    def forward(self, x):               # dpt.py:176-184
        patch_h, patch_w = x.shape[-2] // 14, x.shape[-1] // 14
        features = self.pretrained.get_intermediate_layers(x, self.intermediate_layer_idx[self.encoder], return_class_token=True)
        depth = self.depth_head(features, patch_h, patch_w)
        return F.relu(depth).squeeze(1)

    # ---- device-resident pre/post: replaces image2tensor (dpt.py:196-221) + the bilinear upsample of
    # estimatedepthanything_v2 (src/depthmap_generation.py:548-559) without leaving the GPU --------------------------
    @torch.no_grad()
    def image2tensor_device(self, images_u8, input_size=518):
        """images_u8: uint8 [B, H, W, 3] on the model's device, channel order AS THE FUNNEL HANDS IT OVER (RGB).
        The reference swaps channels three times on the way in (get_raw_prediction :381, estimatedepthanything_v2
        :550, image2tensor :209), i.e. the network sees BGR order with RGB-ordered mean/std -- reproduced here.
        Bicubic resize = cv2.INTER_CUBIC's kernel (a = -0.75, half-pixel centres, replicated border, no antialias);
        cv2 itself is not available in this environment: parity of the resize is unpinned."""
        b, h, w, _ = images_u8.shape
        nw, nh = lower_bound_size(w, h, input_size)
        if images_u8.is_cuda and vm.PREPROCESS_HIP and not vm.STOCK[0]:
            # one pass over the image bytes (ds_preprocess_bicubic), already in the network's dtype
            from src import _native
            dtype = self.pretrained.blocks[0].norm1.weight.dtype
            return _native.preprocess_bicubic(images_u8, (nh, nw), vm.IMAGENET_MEAN, vm.IMAGENET_STD, flip=True, dtype=dtype), (h, w)
        x = images_u8.flip(-1).permute(0, 3, 1, 2).float() / 255.0
        x = F.interpolate(x, size=(nh, nw), mode="bicubic", align_corners=False)
        mean = vm.device_constant(vm.IMAGENET_MEAN, x.device).view(1, 3, 1, 1)
        std = vm.device_constant(vm.IMAGENET_STD, x.device).view(1, 3, 1, 1)
        return (x - mean) / std, (h, w)

    @torch.no_grad()
    def infer_batch(self, images_u8, input_size=518):
        """uint8 [B,H,W,3] -> float32 [B,H,W] raw prediction (larger = nearer), on the device."""
        x, (h, w) = self.image2tensor_device(images_u8, input_size)
        dtype = self.pretrained.blocks[0].norm1.weight.dtype      # the reference casts to the model's dtype (:555)
        depth = self.forward(x.to(dtype)).float()
        return F.interpolate(depth[:, None], (h, w), mode="bilinear", align_corners=True)[:, 0]
