"""Generates tests/golden/model_cases.npz by running the REFERENCE's own torch modules (imported from /root/reference
with stubs for the third-party modules that are absent here and unused by forward(): cv2, torchvision) on synthetic
name-seeded weights (model_weights.py).  Run in the build container only; the tests read the committed .npz.

    python tests/golden/make_golden_models.py
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import model_weights as mw  # noqa: E402

REF = "/root/reference"


def reference_dav2(encoder, features, out_channels):
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    sys.path.insert(0, REF)
    from ddepth_anything_v2.depth_anything_v2.dpt import DepthAnythingV2
    sys.path.pop(0)
    return DepthAnythingV2(encoder=encoder, features=features, out_channels=out_channels)


def reference_dpt(backbone):
    """The reference's own dmidas.dpt_depth.DPTDepthModel, running on a stand-in for timm's Beit containers
    (fake_timm.py): every forward on the path -- beit_forward_features, block_forward, attention_forward,
    _get_rel_pos_bias, forward_adapted_unflatten, ProjectReadout, FeatureFusionBlock_custom, the head -- is the
    reference's code."""
    import fake_timm
    fake_timm.install()
    for name in ("cv2",):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    sys.path.insert(0, REF)
    from dmidas.dpt_depth import DPTDepthModel
    sys.path.pop(0)
    return DPTDepthModel(path=None, backbone=backbone, non_negative=True)


def make_boost_selection_cases():
    """Boost's patch selection (pure Python on an integral image) run with the REFERENCE's own functions."""
    import json
    for name in ("cv2", "skimage", "skimage.measure", "torchvision", "torchvision.transforms", "diffusers", "transformers",
                 "dmarigold", "dmarigold.marigold", "dzoedepth", "dzoedepth.models", "dzoedepth.models.builder",
                 "dzoedepth.utils", "dzoedepth.utils.config", "modules", "modules.shared", "modules.devices"):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    import fake_timm
    fake_timm.install()
    sys.path.insert(0, REF)
    import src.depthmap_generation as ref_dg
    sys.path.pop(0)
    H, W = 700, 1000
    grad, integ = mw.boost_integral_image(5, H, W)
    out = {"integral_checksum": np.array([float(integ[-1, -1]), float(integ[350, 500])])}
    for name, (blsize, factor) in {"case_a": (112, 1.0), "case_b": (96, 0.45)}.items():
        stride = int(round(blsize * 0.75))
        img_stub = np.zeros((H, W, 3))
        grid = ref_dg.applyGridpatch(blsize, stride, img_stub, [0, 0, 0, 0])
        gf = float(grad.sum() / grad.size)
        sel = ref_dg.adaptiveselection(integ, {k: {"rect": list(v["rect"]), "size": v["size"]} for k, v in grid.items()}, gf, factor)
        meta = {"blsize": blsize, "stride": stride, "shape": [H, W, 3], "gf": gf, "factor": factor,
                "grid_rects": [[int(x) for x in grid[str(i)]["rect"]] for i in range(len(grid))],
                "selected_rects": [[int(x) for x in sel[str(i)]["rect"]] for i in range(len(sel))]}
        out[name + "_meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        print(name, len(grid), "grid patches ->", len(sel), "selected")
    np.savez_compressed(os.path.join(HERE, "boost_selection_cases.npz"), **out)


def main():
    out = {}
    torch.manual_seed(0)
    # Depth-Anything-V2 small (src/depthmap_generation.py:243): 10 x 13 patches -> 131 tokens (non-square pos-embed, padding)
    m = reference_dav2('vits', 64, [48, 96, 192, 384]).eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    x = mw.synthetic_image((2, 3, 140, 182), seed=11)
    with torch.no_grad():
        y = m(x)
        taps = m.pretrained.get_intermediate_layers(x, [2, 5, 8, 11], return_class_token=True)
    out["dav2_vits_140x182_out"] = y.numpy()
    out["dav2_vits_140x182_tap3_tokens"] = taps[3][0].numpy()
    out["dav2_vits_140x182_tap0_cls"] = taps[0][1].numpy()
    # square input at the native 37x37 grid would skip the interpolation branch; 70x70 exercises the w == h, npatch != N case
    x2 = mw.synthetic_image((1, 3, 70, 70), seed=12)
    with torch.no_grad():
        out["dav2_vits_70x70_out"] = m(x2).numpy()
    # MiDaS 3.1 DPT BEiT-B/16 (reference backbone "beitb16_384"): 10 x 14 patches -> window (10, 14) != the table's
    # native (24, 24): exercises the bilinear table resize + index gather; 141 tokens -> padded to 192
    m = reference_dpt("beitb16_384").eval()
    sd = mw.fill_state_dict_beit(m.state_dict())
    m.load_state_dict(sd, strict=True)
    x3 = mw.synthetic_image((2, 3, 160, 224), seed=13)
    with torch.no_grad():
        out["dpt_beitb_160x224_out"] = m(x3).numpy()
        l1, l2, l3, l4 = m.forward_transformer(m.pretrained, x3)
    out["dpt_beitb_160x224_layer4"] = l4.numpy()
    out["dpt_beitb_160x224_layer1_sample"] = l1[:, ::16, ::4, ::4].numpy()
    # MiDaS 3.0 dpt_hybrid_384 (backbone "vitb_rn50_384": ViT-B/16 on a ResNetV2-50 stem; BASELINE config 2's network):
    # 160 x 224 input -> 10 x 14 tokens, position embedding resized from 24 x 24
    m = reference_dpt("vitb_rn50_384").eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    x4 = mw.synthetic_image((2, 3, 160, 224), seed=14)
    with torch.no_grad():
        out["dpt_hybrid_160x224_out"] = m(x4).numpy()
        l1, l2, l3, l4 = m.forward_transformer(m.pretrained, x4)
    out["dpt_hybrid_160x224_layer2"] = l2[:, ::8].numpy()
    out["dpt_hybrid_160x224_layer4"] = l4.numpy()
    # LeReS res101 (model id 0, Boost's base estimator): the reference's lib/ is torch-only and imports as is
    sys.path.insert(0, REF)
    from lib.multi_depth_model_woauxi import RelDepthModel
    sys.path.pop(0)
    m = RelDepthModel(backbone='resnext101').eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    x5 = mw.synthetic_image((2, 3, 96, 160), seed=15)
    with torch.no_grad():
        out["leres_96x160_out"] = m.depth_model(x5).numpy()
        feats = m.depth_model.encoder_modules(x5)
    out["leres_96x160_feat3"] = feats[3].numpy()
    # Boost merge network: pix2pix U-Net 'unet_1024', norm none (reference pix2pix/models/networks.py imports as is)
    sys.path.insert(0, REF)
    from pix2pix.models import networks as ref_networks
    sys.path.pop(0)
    g = ref_networks.UnetGenerator(2, 1, 10, 64, norm_layer=ref_networks.get_norm_layer('none'), use_dropout=False).eval()
    sd = mw.fill_state_dict(g.state_dict())
    g.load_state_dict(sd, strict=True)
    x6 = mw.synthetic_image((1, 2, 1024, 1024), seed=16).clamp(-1, 1)
    with torch.no_grad():
        y6 = g(x6.clone())
    out["unet1024_out_sample"] = y6[0, 0, ::8, ::8].numpy()
    out["unet1024_out_mean_abs"] = np.array([float(y6.abs().mean()), float(y6.mean()), float(y6.std())])
    np.savez_compressed(os.path.join(HERE, "model_cases.npz"), **out)
    make_boost_selection_cases()
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).mean()))


if __name__ == "__main__":
    main()
