"""LeReS / AdelaiDepth relative-depth network (reference model id 0 "res101", the base estimator of Boost), MI355X-first.

Reference: lib/multi_depth_model_woauxi.py (RelDepthModel :6-20, DepthModel :23-32), lib/network_auxi.py (Decoder :15-62,
FTB :98-143, FFM :191-213, AO :238-257), lib/Resnext_torch.py (Bottleneck :70-118, ResNet :121-226, resnext101_32x8d
:230-239) and its caller estimateleres / scale_torch (src/depthmap_generation.py:406-440).  Checkpoint key names are
the reference's (depth_model.encoder_modules.encoder.*, depth_model.decoder_modules.*).

Inference-only differences: every BatchNorm is folded into the convolution in front of it once per weight version (the
reference runs conv and BN as two passes over every activation: 104 BN layers in the encoder alone), ReLU is fused by
the library where it can; tensors stay channels_last.  Arithmetic stays float32 like the reference (Boost never uses
half: src/depthmap_generation.py:271).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm


def _folded(conv, bn, cache_holder):
    w, b = conv.weight, conv.bias
    key = (w._version, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, w.dtype,
           w.device, w.data_ptr())
    c = getattr(cache_holder, "_fold_cache", None)
    if c is not None and c[0] == key:
        return c[1], c[2]
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    wf = w * scale.view(-1, 1, 1, 1)
    bf = bn.bias - bn.running_mean * scale
    if b is not None:
        bf = bf + b * scale
    if w.is_cuda:
        wf = wf.contiguous(memory_format=torch.channels_last)
    if not torch.is_grad_enabled():
        if c is not None:
            vm.cache_evicted()
        cache_holder._fold_cache = (key, wf, bf)
    return wf, bf


# Round 5: the grouped 3x3 convolutions of the bottlenecks (40 % of the encoder's convolution time in MIOpen for 7 % of its flops)
# and the `relu(out + identity)` tails run in-tree on float32 channels_last CUDA tensors (csrc/ds_gconv.hip).  A/B switches:
GCONV_HIP = os.environ.get("DS_GCONV", "1") != "0"
ADD_RELU_HIP = os.environ.get("DS_ADD_RELU", "1") != "0"
# Round 6: what follows a LIBRARY convolution -- the (folded) bias, the shortcut / residual add, the ReLU -- is one in-tree pass
# (ds_bias_act_f32) instead of torch's bias pass + add pass + clamp pass; same operations in the same order, same bits.
BIAS_ACT_HIP = os.environ.get("DS_BIAS_ACT_F32", "1") != "0"


def _gconv_image(conv, wf):
    """The folded grouped weight regrouped for ds_gconv3x3_nhwc_f32, cached beside the folded weight it was made from."""
    from src import _native
    c = getattr(conv, "_gconv_cache", None)
    key = (wf.data_ptr(), wf._version)
    if c is not None and c[0] == key:
        return c[1]
    img = _native.gconv_weight_image(wf, conv.groups)
    if not torch.is_grad_enabled():
        if c is not None:
            vm.cache_evicted()
        conv._gconv_cache = (key, img)
    return img


def _wants_grad(*tensors):
    """The in-tree kernels are raw launches outside autograd: a caller that backpropagates through the network keeps torch's ops."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def add_relu(a, b):
    """relu(a + b): one in-tree pass for float32 CUDA tensors of one dense layout, torch's two otherwise."""
    if (ADD_RELU_HIP and not _wants_grad(a, b) and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.stride() == b.stride()
            and a.numel() % 4 == 0 and (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last)) and not vm.STOCK[0]):
        from src import _native
        return _native.add_relu(a, b)
    return F.relu(a + b)


def _library_conv_tail(y, b, relu, res):
    """[relu]((y + b) [+ res]) behind a bias-free library convolution: ds_bias_act_f32 in place where it applies, torch otherwise."""
    if b is not None and BIAS_ACT_HIP and y.is_cuda and not vm.STOCK[0] and not _wants_grad(y, b, res):
        from src import _native
        if _native.bias_act_f32_ok(y, b, res):
            return _native.bias_act_f32(y, b, relu, res)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    if res is not None:
        return add_relu(y, res) if relu else y + res
    return F.relu(y) if relu else y


def conv_act(x, conv, relu=False, res=None):
    """[relu](conv(x) [+ res]) for a plain nn.Conv2d with a bias (the decoder's convolutions, lib/network_auxi.py:116-121)."""
    if conv.bias is None or not (BIAS_ACT_HIP and x.is_cuda and x.dtype == torch.float32 and not vm.STOCK[0] and not _wants_grad(x, conv.weight)):
        y = conv(x)
        if res is not None:
            return add_relu(y, res) if relu else y + res
        return F.relu(y) if relu else y
    return _library_conv_tail(conv._conv_forward(x, conv.weight, None), conv.bias, relu, res)       # honours padding_mode


def conv_bn(x, conv, bn, relu=False, res=None):
    """[relu](BatchNorm(conv(x)) [+ res]) in inference mode as one convolution with folded weights."""
    w, b = _folded(conv, bn, conv)
    cpg = conv.in_channels // max(conv.groups, 1)
    if (GCONV_HIP and conv.groups > 1 and conv.padding_mode == 'zeros' and x.is_cuda and x.dtype == torch.float32 and not vm.STOCK[0]
            and not _wants_grad(x, w) and tuple(conv.stride) == (1, 1) and cpg in (8, 16, 32)):     # (checked BEFORE the layout copy)
        from src import _native
        xc = x.contiguous(memory_format=torch.channels_last)
        if res is None and _native.gconv3x3_supported(xc, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            return _native.gconv3x3(xc, _gconv_image(conv, w), b, relu, conv.in_channels // conv.groups)
    fused = BIAS_ACT_HIP and x.is_cuda and x.dtype == torch.float32 and not vm.STOCK[0] and not _wants_grad(x, w)
    if conv.padding_mode != 'zeros':       # TILING_MODE (src/depthmap_generation.py:250-260): what nn.Conv2d._conv_forward does
        x = F.pad(x, conv._reversed_padding_repeated_twice, mode=conv.padding_mode)
        y = F.conv2d(x, w, None if fused else b, conv.stride, 0, conv.dilation, conv.groups)
    else:
        y = F.conv2d(x, w, None if fused else b, conv.stride, conv.padding, conv.dilation, conv.groups)
    if fused:
        return _library_conv_tail(y, b, relu, res)
    if res is not None:
        return add_relu(y, res) if relu else y + res
    return F.relu(y) if relu else y


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False, groups=32, base_width=8):
        super().__init__()
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):               # Resnext_torch.py:96-118
        identity = x if self.downsample is None else conv_bn(x, self.downsample[0], self.downsample[1])
        out = conv_bn(x, self.conv1, self.bn1, relu=True)
        out = conv_bn(out, self.conv2, self.bn2, relu=True)
        return conv_bn(out, self.conv3, self.bn3, relu=True, res=identity)     # relu(bn3(conv3(out)) + identity): one pass behind the GEMM


class ResNeXt101_32x8d(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))):
            layers = [Bottleneck(inplanes, planes, stride, downsample=True)]
            inplanes = planes * 4
            layers += [Bottleneck(inplanes, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layers))
        for m in self.modules():        # Resnext_torch.py:158-163
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):               # Resnext_torch.py:198-223: features at 1/4, 1/8, 1/16, 1/32
        x = conv_bn(x, self.conv1, self.bn1, relu=True)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        feats = []
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(x)
            feats.append(x)
        return feats


class _Encoder(nn.Module):              # network_auxi.DepthNet: holds `.encoder`
    def __init__(self):
        super().__init__()
        self.encoder = ResNeXt101_32x8d()

    def forward(self, x):
        return self.encoder(x)


class FTB(nn.Module):
    def __init__(self, inchannels, midchannels=512):
        super().__init__()
        self.conv1 = nn.Conv2d(inchannels, midchannels, 3, padding=1, bias=True)
        self.conv_branch = nn.Sequential(nn.ReLU(inplace=True), nn.Conv2d(midchannels, midchannels, 3, padding=1, bias=True),
                                         nn.BatchNorm2d(midchannels), nn.ReLU(inplace=True),
                                         nn.Conv2d(midchannels, midchannels, 3, padding=1, bias=True))
        _init_decoder(self)

    def forward(self, x):               # network_auxi.py:116-121
        # conv_branch starts with ReLU(inplace=True): it rectifies x itself before `x + conv_branch(x)` is formed
        x = conv_act(x, self.conv1, relu=True)
        b = conv_bn(x, self.conv_branch[1], self.conv_branch[2], relu=True)
        return conv_act(b, self.conv_branch[4], relu=True, res=x)             # relu(x + conv(b)): the add commutes bit for bit


class FFM(nn.Module):
    def __init__(self, inchannels, midchannels, outchannels, upfactor=2):
        super().__init__()
        self.ftb1 = FTB(inchannels, midchannels)
        self.ftb2 = FTB(midchannels, outchannels)
        self.upfactor = upfactor

    def forward(self, low_x, high_x):   # network_auxi.py:207-213
        x = self.ftb2(self.ftb1(low_x) + high_x)
        return F.interpolate(x, scale_factor=self.upfactor, mode='bilinear', align_corners=True)


class AO(nn.Module):
    def __init__(self, inchannels, outchannels, upfactor=2):
        super().__init__()
        self.upfactor = upfactor
        self.adapt_conv = nn.Sequential(nn.Conv2d(inchannels, inchannels // 2, 3, padding=1, bias=True),
                                        nn.BatchNorm2d(inchannels // 2), nn.ReLU(inplace=True),
                                        nn.Conv2d(inchannels // 2, outchannels, 3, padding=1, bias=True),
                                        nn.Upsample(scale_factor=upfactor, mode='bilinear', align_corners=True))
        _init_decoder(self)

    def forward(self, x):               # network_auxi.py:255-257
        x = conv_bn(x, self.adapt_conv[0], self.adapt_conv[1], relu=True)
        x = self.adapt_conv[3](x)
        return F.interpolate(x, scale_factor=self.upfactor, mode='bilinear', align_corners=True)


def _init_decoder(mod):                 # network_auxi.py:36-52 and the per-module init_params
    for m in mod.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.01)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        inch, mid = [256, 512, 1024, 2048], [256, 256, 256, 512]
        self.conv = FTB(inch[3], mid[3])
        self.conv1 = nn.Conv2d(mid[3], mid[2], 3, padding=1, bias=True)
        self.ffm2 = FFM(inch[2], mid[2], mid[2])
        self.ffm1 = FFM(inch[1], mid[1], mid[1])
        self.ffm0 = FFM(inch[0], mid[0], mid[0])
        self.outconv = AO(mid[0], 1, upfactor=2)
        _init_decoder(self)

    def forward(self, features):        # network_auxi.py:53-62
        x_32x = self.conv(features[3])
        x_16 = F.interpolate(self.conv1(x_32x), scale_factor=2, mode='bilinear', align_corners=True)
        x_8 = self.ffm2(features[2], x_16)
        x_4 = self.ffm1(features[1], x_8)
        x_2 = self.ffm0(features[0], x_4)
        return self.outconv(x_2)


class DepthModel(nn.Module):
    def __init__(self, encoder='resnext101_stride32x8d'):
        super().__init__()
        if encoder != 'resnext101_stride32x8d':
            raise NotImplementedError("only the resnext101 encoder the reference loads (src/depthmap_generation.py:111)")
        self.encoder_modules = _Encoder()
        self.decoder_modules = Decoder()

    @vm.deterministic_forward
    def forward(self, x):
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        return self.decoder_modules(self.encoder_modules(x))


class RelDepthModel(nn.Module):
    def __init__(self, backbone='resnext101'):
        super().__init__()
        if backbone != 'resnext101':
            raise NotImplementedError("only backbone='resnext101' (src/depthmap_generation.py:111)")
        self.depth_model = DepthModel('resnext101_stride32x8d')

    def inference(self, rgb):
        with torch.no_grad():
            return self.depth_model(rgb)

    # ---- device-resident pre/post of estimateleres (src/depthmap_generation.py:406-440) ---------------------------------
    @torch.no_grad()
    def infer_batch(self, images_u8, net_w, net_h):
        """uint8 [B,H,W,3] as the funnel hands it over (RGB).  get_raw_prediction swaps R/B (:381) and estimateleres swaps
        back (:408), so the network sees RGB; the resize ignores the aspect ratio (cv2.resize default = bilinear,
        half-pixel centres -> torch bilinear align_corners=False, antialias off); cubic resize back (cv2.INTER_CUBIC ->
        torch bicubic, a=-0.75).  cv2 is not available here: the two resizes are parity-unpinned."""
        b, h, w, _ = images_u8.shape
        x = images_u8.permute(0, 3, 1, 2).float() / 255.0
        x = F.interpolate(x, size=(int(net_h), int(net_w)), mode="bilinear", align_corners=False)
        mean = vm.device_constant(vm.IMAGENET_MEAN, x.device).view(1, 3, 1, 1)
        std = vm.device_constant(vm.IMAGENET_STD, x.device).view(1, 3, 1, 1)
        x = ((x - mean) / std).to(self.depth_model.decoder_modules.conv1.weight.dtype)
        pred = self.depth_model(x).float()
        return F.interpolate(pred, size=(h, w), mode="bicubic", align_corners=False)[:, 0]
