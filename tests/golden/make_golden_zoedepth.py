"""Golden outputs of the REFERENCE's own ZoeDepth code (dzoedepth/models/zoedepth/zoedepth_v1.py, zoedepth_nk_v1.py,
base_models/midas.py, layers/*, depth_model.py) -> tests/golden/zoedepth_cases.npz.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_zoedepth.py

Everything on the path is the reference's code, built through its own get_config + build_model like
src/depthmap_generation.py:196-209 does.  Outside shims:
  * torch.hub.load -- the reference fetches the DPT core from GitHub (midas.py:343; no network here) -- returns the
    reference's own vendored dmidas.DPTDepthModel of the same architecture, running on the stand-in for the un-vendored
    timm containers (fake_timm.py, as for model_cases.npz);
  * torchvision (absent): transforms.Normalize / ToTensor are the two tiny classes below; cv2: MagicMock (unused).
The numeric cases use the BEiT-B core (midas_model_type override, same 256-channel decoder features) on small inputs so
that the fixture stays small; the default BEiT-L builds contribute their state-dict names and shapes.
Weights: model_weights.fill_state_dict_zoe (a function of each tensor's name).
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
BACKBONES = {"DPT_BEiT_L_384": "beitl16_384", "DPT_BEiT_B_384": "beitb16_384"}


def install_shims():
    import fake_timm
    fake_timm.install()
    sys.modules.setdefault("cv2", mock.MagicMock())
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
            s = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
            return (x - m) / s

    class ToTensor:
        def __call__(self, pil):
            a = np.asarray(pil, dtype=np.uint8)
            return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255)

    tr.Normalize, tr.ToTensor = Normalize, ToTensor
    tv.transforms = tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
    sys.path.insert(0, REF)
    from dmidas.dpt_depth import DPTDepthModel

    def hub_load(repo, name, **kw):
        return DPTDepthModel(path=None, backbone=BACKBONES[name], non_negative=True)
    torch.hub.load = hub_load


def build(model_name, **kw):
    from dzoedepth.models.builder import build_model
    from dzoedepth.utils.config import get_config
    conf = get_config(model_name, "infer", **kw)
    conf["pretrained_resource"] = None                       # no checkpoint: synthetic weights below
    m = build_model(conf).eval()
    m.load_state_dict(mw.fill_state_dict_zoe(m.state_dict()), strict=True)
    return m


def run_case(m, x, w, h):
    m.core.prep.resizer._Resize__width = w                   # estimatezoedepth, src/depthmap_generation.py:448-449
    m.core.prep.resizer._Resize__height = h
    with torch.no_grad():
        full = m.infer(x)
        raw = m(x)
    return full.numpy(), raw['metric_depth'].numpy(), raw


def main():
    install_shims()
    out = {}
    names = {}
    for tag, model_name, kw in (("n", "zoedepth", {}), ("k", "zoedepth", {"config_version": "kitti"}), ("nk", "zoedepth_nk", {})):
        m = build(model_name, midas_model_type="DPT_BEiT_B_384", **kw)
        x = torch.rand((1, 3, 88, 120), generator=torch.Generator().manual_seed(21))
        full, raw, d = run_case(m, x, 160, 128)
        out[f"{tag}_88x120_infer"] = full
        out[f"{tag}_88x120_forward"] = raw
        if "domain_logits" in d:
            out[f"{tag}_88x120_domain_logits"] = d["domain_logits"].numpy()
        x2 = torch.rand((2, 3, 70, 150), generator=torch.Generator().manual_seed(22))
        out[f"{tag}_70x150_infer"] = run_case(m, x2, 224, 96)[0]
        print(tag, full.shape, float(full.mean()), float(full.std()))
        del m
        # names + shapes of the default (BEiT-L) build, heads only (the core is covered by the DPT goldens)
        big = build(model_name, **kw) if tag != "k" else None
        if big is not None:
            sd = big.state_dict()
            names[tag] = sorted(f"{k}:{tuple(v.shape)}" for k, v in sd.items() if not k.startswith("core.core.pretrained"))
            out[f"{tag}_n_tensors"] = np.array([len(sd)])
            del big
    for tag, lst in names.items():
        out[f"{tag}_head_names"] = np.array(lst)
    np.savez_compressed(os.path.join(HERE, "zoedepth_cases.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
