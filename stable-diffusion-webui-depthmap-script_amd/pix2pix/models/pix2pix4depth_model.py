"""Inference face of the reference's Pix2Pix4DepthModel (pix2pix/models/pix2pix4depth_model.py): set_input (:96-106),
forward (:114-116) and the `(fake_B + 1) / 2` every caller applies (src/depthmap_generation.py:906-907,1040-1041), batched
and device resident.  Weights: './models/pix2pix/latest_net_G.pth' (src/depthmap_generation.py:287-299)."""
import os

import torch
import torch.nn as nn

from . import networks


class Pix2Pix4DepthModel(nn.Module):
    def __init__(self, opt=None):
        super().__init__()
        self.netG = networks.UnetGenerator(2, 1, 10, 64)
        self.save_dir = './models/pix2pix'

    def load_networks(self, epoch='latest'):
        path = os.path.join(self.save_dir, f'{epoch}_net_G.pth')
        sd = torch.load(path, map_location='cpu')
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        self.netG.load_state_dict(sd, strict=True)

    @staticmethod
    def _norm(t):
        """per-image min/max to [0,1], then *2-1 (pix2pix4depth_model.py:100-104, 109-112)."""
        mn = t.amin(dim=(-2, -1), keepdim=True)
        mx = t.amax(dim=(-2, -1), keepdim=True)
        return ((t - mn) / (mx - mn)) * 2 - 1

    @torch.no_grad()
    def merge(self, outer, inner):
        """outer, inner: float32 [P, 1024, 1024] (low-res / high-res estimate or base / patch estimate).
        Returns (fake_B + 1) / 2 as float32 [P, 1024, 1024]."""
        a = torch.stack((self._norm(outer), self._norm(inner)), dim=1).to(next(self.netG.parameters()).dtype)
        return (self.netG(a)[:, 0].float() + 1) / 2
