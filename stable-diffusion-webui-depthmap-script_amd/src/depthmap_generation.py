"""Model side of the funnel: ``ModelHolder`` (reference: src/depthmap_generation.py:40-403).

The holder keeps the reference's interface (``ensure_models``, ``get_raw_prediction``, ``offload``/``reload``/
``unload_models``, ``update_settings``, ``get_default_net_size``) so ``core_generation_funnel`` is wired exactly like
the reference.  Built model families:
    id  0           LeReS res101 (ResNeXt101-32x8d)      (lib.multi_depth_model_woauxi.RelDepthModel; reference :101-114)
    ids 1, 2        MiDaS 3.1 DPT BEiT-L/16 512 / 384   (dmidas.dpt_depth.DPTDepthModel; reference :116-146)
    ids 3, 4        MiDaS 3.0 dpt_large_384 (ViT-L/16) / dpt_hybrid_384 (ViT-B/16 + ResNetV2-50 stem)   (reference :147-170)
    ids 7, 8, 9     ZoeDepth N / K / NK on the DPT BEiT-L/16 384 core   (dzoedepth.zoedepth; reference :196-209, :443-452)
    ids 12, 13, 14  Depth-Anything-V2 small/base/large   (ddepth_anything_v2.DepthAnythingV2; reference :237-248)
Checkpoints are looked up in ``model_dir`` under the reference's file names; the reference downloads them when missing
(ensure_file_downloaded) -- this build has no network path and raises FileNotFoundError instead, unless
``allow_random_init`` is set (bench / tests: random weights of the same architecture).
Other ids (midas_v21 5-6, Marigold 10, Depth-Anything v1 11: their networks are not vendored by the reference) are
not built: ``ensure_models`` raises NotImplementedError unless a predictor was registered with ``register_predictor``.
Boost (``boost=True``; src/boost.py + the pix2pix merge network + ds_boost_blend) runs on every built base model.
Nothing ever falls back silently.
"""
import gc
import os

# Appendix B of SURVEY.md: ids whose raw output is near-is-dark (src/depthmap_generation.py:402)
INVERTED_MODEL_IDS = (0, 7, 8, 9, 10)

DEFAULT_NET_SIZES = {          # src/depthmap_generation.py:323-339
    0: [448, 448], 1: [512, 512], 2: [384, 384], 3: [384, 384], 4: [384, 384], 5: [384, 384], 6: [256, 256],
    7: [384, 512], 8: [384, 768], 9: [384, 512], 10: [768, 768], 11: [518, 518], 12: [518, 518], 13: [518, 518],
    14: [518, 518],
}


def _build_dpt_beit(backbone, filename):
    def make():
        from dmidas.dpt_depth import DPTDepthModel
        return DPTDepthModel(path=None, backbone=backbone, non_negative=True), filename
    return make


def _build_dav2(letter):
    cfg = {'s': ('vits', 64, [48, 96, 192, 384]), 'b': ('vitb', 128, [96, 192, 384, 768]), 'l': ('vitl', 256, [256, 512, 1024, 1024])}[letter]

    def make():
        from ddepth_anything_v2 import DepthAnythingV2
        return DepthAnythingV2(encoder=cfg[0], features=cfg[1], out_channels=cfg[2]), f"depth_anything_v2_vit{letter}.pth"
    return make


def _build_zoe(kind):
    def make():
        from dzoedepth import build_zoedepth
        return build_zoedepth(kind)
    return make


def _build_leres():
    from lib.multi_depth_model_woauxi import RelDepthModel
    return RelDepthModel(backbone='resnext101'), "res101.pth"


_BUILDERS = {0: _build_leres, 1: _build_dpt_beit("beitl16_512", "dpt_beit_large_512.pt"), 2: _build_dpt_beit("beitl16_384", "dpt_beit_large_384.pt"),
             3: _build_dpt_beit("vitl16_384", "dpt_large-midas-2f21e586.pt"), 4: _build_dpt_beit("vitb_rn50_384", "dpt_hybrid-midas-501f0c75.pt"),
             7: _build_zoe("zoedepth_n"), 8: _build_zoe("zoedepth_k"), 9: _build_zoe("zoedepth_nk"),
             12: _build_dav2('s'), 13: _build_dav2('b'), 14: _build_dav2('l')}


def apply_tiling_mode(net):
    """reference :250-260: every module whose type is EXACTLY nn.Conv2d / nn.Conv1d (subclasses such as the hybrid stem's
    weight-standardised convolutions are left alone, like in the reference) pads circularly, so that the depth map of a
    tileable texture tiles.  Returns the number of layers switched."""
    import torch
    n = 0
    for m in net.modules():
        if type(m) is torch.nn.Conv2d or type(m) is torch.nn.Conv1d:
            m.padding_mode = 'circular'
            n += 1
    return n


def _load_pix2pix(device, allow_random_init):
    """reference :287-299: './models/pix2pix/latest_net_G.pth' (downloaded there when missing)."""
    from pix2pix.models.pix2pix4depth_model import Pix2Pix4DepthModel
    m = Pix2Pix4DepthModel()
    path = os.path.join(m.save_dir, 'latest_net_G.pth')
    if os.path.exists(path):
        m.load_networks('latest')
    elif not allow_random_init:
        raise FileNotFoundError(f"{path} not found (the reference would download it; this build has no network path)")
    return m.eval().to(device)


class _NetPredictor:
    """One loaded network + its device-resident pre/post-processing (estimatemidas :455-499 / estimatedepthanything_v2
    :548-559).  Callable like a registered predictor: (pil_image, net_width, net_height, device) -> float32 tensor [H,W]."""

    def __init__(self, model_type, device, model_dir, allow_random_init, no_half, tiling_mode=False):
        import torch
        net, filename = _BUILDERS[model_type]()
        self.tiling_mode = bool(tiling_mode)
        if tiling_mode:
            apply_tiling_mode(net)
        self_model_dir_unset = model_dir is None
        if model_dir is None:                                # reference :83-92
            model_dir = {0: "./models/leres", 11: "./models/depth_anything", 12: "./models/depth_anything_v2",
                         13: "./models/depth_anything_v2", 14: "./models/depth_anything_v2"}.get(model_type, "./models/midas")
        if model_type in (7, 8, 9) and self_model_dir_unset:
            # the reference lets torch.hub.load_state_dict_from_url cache ZoeD_M12_*.pt (dzoedepth/models/model_io.py:62-64)
            model_dir = os.path.join(torch.hub.get_dir(), "checkpoints")
        path = os.path.join(model_dir, filename)
        if os.path.exists(path):
            sd = torch.load(path, map_location='cpu')
            if model_type in (7, 8, 9):                      # model_io.py:34-50: optional 'model' wrapper, DataParallel prefix
                sd = sd.get('model', sd)
                sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
            if model_type == 0:                              # reference :108-112: checkpoint['depth_model'], "module." stripped
                sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd['depth_model'].items()}
            elif "optimizer" in sd:
                sd = sd["model"]
            missing, unexpected = net.load_state_dict(sd, strict=False)
            # timm's index buffers / k_bias may or may not be stored; anything else missing is an error
            bad = [k for k in missing if not k.endswith(("relative_position_index", "k_bias"))]
            if bad:
                raise RuntimeError(f"{path}: checkpoint lacks {len(bad)} tensors, e.g. {bad[:3]}")
        elif not allow_random_init:
            raise FileNotFoundError(f"{path} not found (the reference would download it; this build has no network path). "
                                    "Place the checkpoint there or set ModelHolder.allow_random_init for a dry run")
        self.model_type = model_type
        self.no_half = bool(no_half)
        self.hip_graphs = "auto"
        self._graphed = {}
        self.net = net.eval().to(device)
        dev = torch.device(device)
        # reference :266-275: LeReS stays float32, and so does zoedepth_n ("completely trips and generates black images")
        if dev.type == 'cuda' and not no_half and model_type not in (0, 7):
            self.net = self.net.half()

    @property
    def is_dav2(self):
        return self.model_type in (12, 13, 14)

    def __call__(self, pil_image, net_width, net_height, device):
        import numpy as np
        import torch
        img = torch.from_numpy(np.array(pil_image.convert("RGB"), dtype=np.uint8, order="C")).to(next(self.net.parameters()).device)
        return self.predict_batch(img.unsqueeze(0), net_width, net_height)[0]

    def predict_batch(self, batch, net_width, net_height):
        """uint8 RGB [B,H,W,3] on the network's device -> float32 [B,H,W] raw predictions: ONE forward for the batch.
        The launches of that forward are captured once per (shape, net size) into a hipGraph and replayed (src/hip_graph.py) --
        ModelHolder's `hip_graphs` setting: "auto" (default: from the third use of a shape), True (from the first), False (never)."""
        mode = getattr(self, "hip_graphs", False)
        if mode and batch.is_cuda:
            from .hip_graph import GraphedForward
            key = (int(net_width), int(net_height))
            gf = self._graphed.get(key)
            if gf is None:
                while len(self._graphed) >= 4:               # NET_SIZE_MATCH makes the net size follow the images: bounded like the shapes
                    self._graphed.pop(next(iter(self._graphed)))
                # "auto" (the default since round 6): a (shape, net size) runs eager twice, is captured on its third use and
                # replayed from then on, when the replay reproduces the eager result; True: captured on first use
                gf = self._graphed[key] = GraphedForward(lambda b: self._predict_batch_eager(b, net_width, net_height),
                                                         lazy=2 if mode == "auto" else 0)
            return gf(batch)
        return self._predict_batch_eager(batch, net_width, net_height)

    def _predict_batch_eager(self, batch, net_width, net_height):
        if self.model_type == 0:
            return self.net.infer_batch(batch, int(net_width), int(net_height))      # estimateleres (:406-421)
        if self.model_type in (7, 8, 9):
            return self.net.infer_batch(batch, int(net_width), int(net_height))      # estimatezoedepth (:443-452)
        if self.is_dav2:
            return self.net.infer_batch(batch, int(net_width))             # reference passes net_width as input_size (:553)
        mode = "minimal"                                                    # resize_mode of ids 1-4 (:127, :141, :155, :168)
        return self.net.infer_batch(batch, net_size=int(net_width), resize_mode=mode, net_h=int(net_height))


class ModelHolder:
    def __init__(self):
        self.depth_model = None
        self.pix2pix_model = None
        self.depth_model_type = None
        self.device = None
        self.offloaded = False
        self.resize_mode = None
        self.normalization = None
        self.tiling_mode = False
        self._predictors = {}
        self.model_dir = None                    # None: the reference's per-family directories (:83-92); set to override
        self.allow_random_init = False
        self.no_half = False

    def update_settings(self, **kvargs):
        """reference :54-57 -- free-form settings (boost_rmax, precision, no_half, ...) become attributes."""
        for k, v in kvargs.items():
            setattr(self, k, v)

    def register_predictor(self, model_type, fn):
        """fn(pil_image, net_width, net_height, device) -> float32 tensor/ndarray [H,W] at image size (raw model output)."""
        self._predictors[model_type] = fn

    def ensure_models(self, model_type, device, boost: bool, tiling_mode: bool = False):
        """reference :60-74."""
        from . import miopen_db
        miopen_db.seed()                 # the package's MIOpen find results, before the first library convolution (src/miopen_db.py)
        if boost:
            if model_type not in (0, 1, 2, 3, 4, 7, 8, 9, 12, 13, 14):
                raise NotImplementedError(f"Boost with depth model id {model_type!r} is not built (built: 0 LeReS, 1-4 MiDaS "
                                          "DPT, 7-9 ZoeDepth, 12-14 Depth-Anything-V2)")
            if self.pix2pix_model is None:
                self.pix2pix_model = _load_pix2pix(device, self.allow_random_init)
        else:
            self.pix2pix_model = None
        if model_type in self._predictors:
            self.depth_model = self._predictors[model_type]
        elif model_type in _BUILDERS:
            from . import gemm_tuning
            gemm_tuning.enable()                      # tuned hipBLASLt / rocBLAS solutions for the transformer linears
            # Boost never runs a MiDaS/LeReS/ZoeDepth base network in half precision (reference :271); DA-V2 stays half
            # (:273-275).  The reference reloads whenever the boost switch changes (:62-72); here only if the precision does.
            no_half = bool(self.no_half or (boost and model_type not in (12, 13, 14)))
            if (self.depth_model is None or self.depth_model_type != model_type or not isinstance(self.depth_model, _NetPredictor)
                    or self.depth_model.no_half != no_half or self.depth_model.tiling_mode != bool(tiling_mode)):
                self.depth_model = _NetPredictor(model_type, device, self.model_dir, self.allow_random_init, no_half, tiling_mode)
            self.depth_model.hip_graphs = getattr(self, "hip_graphs", "auto")      # "auto" | True | False (see predict_batch)
        else:
            raise NotImplementedError(
                f"depth model {model_type!r} is not available in this build (built: ids {sorted(_BUILDERS)}); register a "
                "predictor with ModelHolder.register_predictor or pass precomputed depthmaps")
        self.depth_model_type = model_type
        self.device = device
        self.tiling_mode = tiling_mode
        self.offloaded = False

    def get_default_net_size(self, model_type):
        if model_type in DEFAULT_NET_SIZES:
            return DEFAULT_NET_SIZES[model_type]
        return [512, 512]

    def get_raw_prediction(self, input, net_width, net_height):
        """reference :375-403 -> (prediction [H,W], invert?)."""
        if self.pix2pix_model is not None:                   # reference :399-401: net size is ignored with Boost
            import numpy as np
            import torch
            from . import boost
            img = torch.from_numpy(np.array(input.convert("RGB"), dtype=np.uint8, order="C")).to(self.device)
            raw = boost.estimateboost(img, self.depth_model.net, self.depth_model_type, self.pix2pix_model,
                                      int(getattr(self, "boost_rmax", 1600)), group=getattr(self, "boost_group", None))
        else:
            raw = self.depth_model(input, net_width, net_height, self.device)
        return raw, (self.depth_model_type in INVERTED_MODEL_IDS)

    def get_raw_prediction_batch(self, pil_images, rgb_u8, net_width, net_height):
        """get_raw_prediction for a device batch of same-size images (the funnel's unit of work on an MI355X).
        pil_images: the PIL inputs; rgb_u8: the same pixels as a uint8 [B,H,W,3] tensor on the device.
        -> (float32 [B,H,W] on the device, invert?).  A built network runs ONE forward for the batch; a registered predictor
        is called through its optional ``batch_tensor(rgb_u8, net_width, net_height)`` (the pixels the funnel has already
        uploaded), its optional ``batch(pils, net_width, net_height, device)``, or image by image; Boost renders image
        by image (estimateboost batches its own patches)."""
        import torch
        invert = self.depth_model_type in INVERTED_MODEL_IDS
        if self.pix2pix_model is not None:
            from . import boost
            # boost_group (update_settings): a torch.distributed group to shard every image's PATCHES over (one process per
            # GPU, all ranks call the funnel with the same images); only rank 0 of the group gets predictions back
            grp = getattr(self, "boost_group", None)
            outs = [boost.estimateboost(rgb_u8[i], self.depth_model.net, self.depth_model_type, self.pix2pix_model,
                                        int(getattr(self, "boost_rmax", 1600)), group=grp) for i in range(rgb_u8.shape[0])]
            if any(o is None for o in outs):
                return None, invert
            return torch.stack([torch.as_tensor(o) for o in outs]), invert
        if isinstance(self.depth_model, _NetPredictor):
            return self.depth_model.predict_batch(rgb_u8, net_width, net_height), invert
        if hasattr(self.depth_model, "batch_tensor"):          # a registered predictor that takes the uploaded pixels, like _NetPredictor
            return torch.as_tensor(self.depth_model.batch_tensor(rgb_u8, net_width, net_height)), invert
        if hasattr(self.depth_model, "batch"):
            return torch.as_tensor(self.depth_model.batch(pil_images, net_width, net_height, self.device)), invert
        outs = []
        for im in pil_images:
            r = self.depth_model(im, net_width, net_height, self.device)
            outs.append(r if torch.is_tensor(r) else torch.from_numpy(__import__("numpy").asarray(r)))
        return torch.stack([o.to(self.device) for o in outs]), invert

    def offload(self):
        """reference :341-350 moves the networks to host RAM between runs to spare a consumer card's VRAM.  On a 288 GB
        MI355X the largest built set (BEiT-L + pix2pix, < 2 GB in float32) stays resident: the call only records the
        state, so the next run starts without a 1-2 GB host -> device copy."""
        self.offloaded = True

    def reload(self):
        """reference :352-360 (counterpart of offload)."""
        self.offloaded = False

    def unload_models(self):
        self.depth_model = None
        self.pix2pix_model = None
        self.depth_model_type = None
        self.device = None
        gc.collect()
