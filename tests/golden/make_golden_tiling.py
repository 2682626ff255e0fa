"""Golden outputs for TILING_MODE (reference src/depthmap_generation.py:250-260: every module whose type is exactly
nn.Conv2d / nn.Conv1d gets padding_mode='circular' after construction) from the reference's own modules, built like
make_golden_models.py builds them.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tiling.py        -> tiling_cases.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden_models as mgm  # noqa: E402
import model_weights as mw  # noqa: E402


def hijack(model):
    """The reference's own lines (:251-260), applied from outside because they live inside ModelHolder.load_models."""
    def flatten(el):
        flattened = [flatten(children) for children in el.children()]
        res = [el]
        for c in flattened:
            res += c
        return res
    n = 0
    for layer in [layer for layer in flatten(model) if type(layer) == torch.nn.Conv2d or type(layer) == torch.nn.Conv1d]:
        layer.padding_mode = 'circular'
        n += 1
    return n


def main():
    out = {}
    m = mgm.reference_dpt("beitb16_384").eval()
    m.load_state_dict(mw.fill_state_dict_beit(m.state_dict()), strict=True)
    out["dpt_beitb_n_convs"] = np.array([hijack(m)])
    with torch.no_grad():
        out["dpt_beitb_160x224_out"] = m(mw.synthetic_image((2, 3, 160, 224), seed=13)).numpy()
    m = mgm.reference_dpt("vitb_rn50_384").eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    out["dpt_hybrid_n_convs"] = np.array([hijack(m)])
    with torch.no_grad():
        out["dpt_hybrid_160x224_out"] = m(mw.synthetic_image((2, 3, 160, 224), seed=14)).numpy()
    m = mgm.reference_dav2('vits', 64, [48, 96, 192, 384]).eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    out["dav2_vits_n_convs"] = np.array([hijack(m)])
    with torch.no_grad():
        out["dav2_vits_140x182_out"] = m(mw.synthetic_image((2, 3, 140, 182), seed=11)).numpy()
    sys.path.insert(0, mgm.REF)
    from lib.multi_depth_model_woauxi import RelDepthModel
    sys.path.pop(0)
    m = RelDepthModel(backbone='resnext101').eval()
    m.load_state_dict(mw.fill_state_dict(m.state_dict()), strict=True)
    out["leres_n_convs"] = np.array([hijack(m)])
    with torch.no_grad():
        out["leres_96x160_out"] = m.depth_model(mw.synthetic_image((2, 3, 96, 160), seed=15)).numpy()
    np.savez_compressed(os.path.join(HERE, "tiling_cases.npz"), **out)
    print({k: (v.shape, v.ravel()[:1]) for k, v in out.items()})


if __name__ == "__main__":
    main()
