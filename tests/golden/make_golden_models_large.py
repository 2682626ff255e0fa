"""Golden outputs at the BENCHMARKED network shapes (tests/golden/model_cases_large.npz), made by the REFERENCE's own
torch modules on name-seeded synthetic weights (model_weights.py) -- same recipe as make_golden_models.py, full size:

  * dpt_beit_large_512 (model id 1, BASELINE config 3): the reference's dmidas code on the fake_timm containers,
    batch 1, 512x512 input -> 32x32 + 1 = 1025 tokens (the launch shape of k_attention_fwd in bench.py, with the bias);
  * Depth-Anything-V2 ViT-L (model id 14, BASELINE config 5): the vendored reference modules as they are,
    batch 1, 518x924 input (a 1080p frame at input_size 518) -> 37x66 + 1 = 2443 tokens.

Run in the build container only (float32 on the CPU, a few minutes):  python tests/golden/make_golden_models_large.py
Stored: every second row / column of the output plus summary statistics (the tests compare the same sample).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import model_weights as mw  # noqa: E402
import make_golden_models as mgm  # noqa: E402


def stats(y):
    return np.array([float(y.mean()), float(y.std()), float(y.min()), float(y.max())])


def main():
    out = {}
    m = mgm.reference_dpt("beitl16_512").eval()
    m.load_state_dict(mw.fill_state_dict_beit(m.state_dict()), strict=True)
    x = mw.synthetic_image((1, 3, 512, 512), seed=31)
    with torch.no_grad():
        y = m(x).numpy()
        l1, l2, l3, l4 = m.forward_transformer(m.pretrained, x)
    out["dpt_beitl512_512x512_layer4_s"] = l4[:, ::4].numpy().copy()          # reassembled tap of block 23: [1, 256, 16, 16]
    out["dpt_beitl512_512x512_out_s2"] = y[:, ::2, ::2].copy()
    out["dpt_beitl512_512x512_stats"] = stats(y)
    print("dpt_beit_large_512", y.shape, stats(y))
    del m
    m = mgm.reference_dav2('vitl', 256, [256, 512, 1024, 1024]).eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    x = mw.synthetic_image((1, 3, 518, 924), seed=32)
    with torch.no_grad():
        y = m(x).numpy()
        taps = m.pretrained.get_intermediate_layers(x, [4, 11, 17, 23], return_class_token=True)
    # the head's final ReLUs leave a sparse map under random weights: the encoder taps (what 24 blocks of the fused attention
    # produce at N = 2443) are pinned as well -- every 8th token, every 4th channel of taps 11 and 23
    out["dav2_vitl_518x924_tap1_s"] = taps[1][0][:, ::8, ::4].numpy().copy()
    out["dav2_vitl_518x924_tap3_s"] = taps[3][0][:, ::8, ::4].numpy().copy()
    out["dav2_vitl_518x924_out_s2"] = y[:, ::2, ::2].copy()
    out["dav2_vitl_518x924_stats"] = stats(y)
    print("dav2_vitl", y.shape, stats(y))
    np.savez_compressed(os.path.join(HERE, "model_cases_large.npz"), **out)


if __name__ == "__main__":
    main()
