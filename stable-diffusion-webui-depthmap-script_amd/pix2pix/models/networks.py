"""The merge network of Boost: pix2pix U-Net generator 'unet_1024' (10 down-samplings, 2 -> 1 channels, no normalisation).

Reference: pix2pix/models/networks.py (define_G :119-166, UnetGenerator :444-473, UnetSkipConnectionBlock :476-550),
instantiated by Pix2Pix4DepthModel (pix2pix/models/pix2pix4depth_model.py:58-59: norm 'none', no dropout).  The module
tree reproduces the reference's nested nn.Sequential indices so that 'latest_net_G.pth' loads by key name
(model.model.0.weight, model.model.1.model.1.weight, ...).

Forward is a flat loop over the ten levels instead of ten nested module calls.  One quirk of the reference is kept on
purpose: its down-path LeakyReLU is in-place, so the tensor that reaches the skip concatenation is the ACTIVATED input
of the block (networks.py:509,545-550: `torch.cat([x, self.model(x)], 1)` after `self.model` has rectified x in place).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from src import vit_mi355x as vm


RELU_CAT_HIP = os.environ.get("DS_RELU_CAT", "1") != "0"         # A/B switch


def _relu_cat(s, t):
    """relu(torch.cat([s, t], 1))"""
    if RELU_CAT_HIP and s.is_cuda and not vm.STOCK[0] and not (torch.is_grad_enabled() and (s.requires_grad or t.requires_grad)):
        from src import _native
        if _native.relu_cat_f32_ok(s, t):
            return _native.relu_cat_f32(s, t)
    return F.relu(torch.cat([s, t], 1))


class _Identity(nn.Module):
    def forward(self, x):
        return x


class UnetSkipConnectionBlock(nn.Module):
    """Parameter container with the reference's Sequential layout (norm_layer = Identity, use_bias = False)."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False):
        super().__init__()
        self.outermost, self.innermost = outermost, innermost
        if input_nc is None:
            input_nc = outer_nc
        downconv = nn.Conv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1, bias=False)
        if outermost:
            upconv = nn.ConvTranspose2d(inner_nc * 2, outer_nc, kernel_size=4, stride=2, padding=1)
            model = [downconv, submodule, nn.ReLU(True), upconv, nn.Tanh()]
        elif innermost:
            upconv = nn.ConvTranspose2d(inner_nc, outer_nc, kernel_size=4, stride=2, padding=1, bias=False)
            model = [nn.LeakyReLU(0.2, True), downconv, nn.ReLU(True), upconv, _Identity()]
        else:
            upconv = nn.ConvTranspose2d(inner_nc * 2, outer_nc, kernel_size=4, stride=2, padding=1, bias=False)
            model = [nn.LeakyReLU(0.2, True), downconv, _Identity(), submodule, nn.ReLU(True), upconv, _Identity()]
        self.model = nn.Sequential(*model)

    @property
    def downconv(self):
        return self.model[0] if self.outermost else self.model[1]

    @property
    def upconv(self):
        return self.model[3] if (self.outermost or self.innermost) else self.model[5]

    @property
    def submodule(self):
        return None if self.innermost else (self.model[1] if self.outermost else self.model[3])


class UnetGenerator(nn.Module):
    def __init__(self, input_nc=2, output_nc=1, num_downs=10, ngf=64):
        super().__init__()
        blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, innermost=True)
        for _ in range(num_downs - 5):
            blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=blk)
        blk = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=blk)
        blk = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=blk)
        blk = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=blk)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True)
        for m in self.modules():        # init_weights 'normal', gain 0.02 (networks.py:67-96)
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, 0.0, 0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.0)

    def levels(self):
        out, b = [], self.model
        while b is not None:
            out.append(b)
            b = b.submodule
        return out

    @vm.deterministic_forward
    def forward(self, x):
        lv = self.levels()
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        skips = []
        t = lv[0].downconv(x)                                   # outermost: no activation in front of the first conv
        for b in lv[1:]:
            t = F.leaky_relu(t, 0.2)                            # in-place in the reference: this IS the skip tensor
            skips.append(t)
            t = b.downconv(t)
        # up path: `cat([skip, upconv(relu(t))])` is read by the parent's ReLU and by nothing else (networks.py:519,545-550), so the
        # concatenation is written rectified: one pass (ds_relu_cat_f32) instead of a copy pass and a clamp pass
        r = F.relu(t)
        for b, s in zip(reversed(lv[1:]), reversed(skips)):
            r = _relu_cat(s, b.upconv(r))
        return torch.tanh(lv[0].upconv(r))
