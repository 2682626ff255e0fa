"""Golden output of the metric's network in its SECOND form (tests/golden/model_cases_net1024.npz): dpt_beit_large_512 with
NET_SIZE_MATCH on a 1024 x 1024 frame -- net 1024, 64 x 64 + 1 = 4097 tokens (reference: src/core.py:177-181 sets net_width /
net_height to the image size; the BEiT blocks then interpolate their relative-position tables to the 64 x 64 window,
dmidas/backbones/beit.py:38-63) -- made by the REFERENCE's own dmidas modules on the name-seeded synthetic weights of
model_weights.py (fake_timm containers, as make_golden_models_large.py), batch 1, float32 on the CPU.

Build container only (a few minutes):  python tests/golden/make_golden_models_net1024.py
Stored: every fourth row / column of the depth, the reassembled tap of block 23 (every fourth channel) and summary statistics.
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
from make_golden_models_large import stats  # noqa: E402


def main():
    m = mgm.reference_dpt("beitl16_512").eval()
    m.load_state_dict(mw.fill_state_dict_beit(m.state_dict()), strict=True)
    x = mw.synthetic_image((1, 3, 1024, 1024), seed=33)
    with torch.no_grad():
        y = m(x).numpy()
        l1, l2, l3, l4 = m.forward_transformer(m.pretrained, x)
    out = {"dpt_beitl512_1024x1024_layer4_s": l4[:, ::4].numpy().copy(),          # [1, 256, 32, 32]
           "dpt_beitl512_1024x1024_out_s4": y[:, ::4, ::4].copy(),
           "dpt_beitl512_1024x1024_stats": stats(y)}
    print("dpt_beit_large_512 @ net 1024", y.shape, stats(y))
    np.savez_compressed(os.path.join(HERE, "model_cases_net1024.npz"), **out)


if __name__ == "__main__":
    main()
