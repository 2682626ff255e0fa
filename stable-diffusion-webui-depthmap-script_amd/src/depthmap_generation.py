"""Model side of the funnel: ``ModelHolder`` (reference: src/depthmap_generation.py:40-403).

Round-1 state: the holder keeps the reference's interface (``ensure_models``, ``get_raw_prediction``,
``offload``/``reload``/``unload_models``, ``update_settings``, ``get_default_net_size``) so
``core_generation_funnel`` is wired exactly like the reference, but the model families themselves
(DPT/BEiT, ViT-hybrid, Depth-Anything-V2, LeReS+pix2pix Boost -- SURVEY.md 8a rows a10-a17) are not built
yet.  A depth predictor can be plugged in with ``register_predictor`` (tests and the bench use a
synthetic one); without it ``ensure_models`` raises -- it never silently falls back to anything.
"""
import gc

# Appendix B of SURVEY.md: ids whose raw output is near-is-dark (src/depthmap_generation.py:402)
INVERTED_MODEL_IDS = (0, 7, 8, 9, 10)

DEFAULT_NET_SIZES = {          # src/depthmap_generation.py:323-339
    0: [448, 448], 1: [512, 512], 2: [384, 384], 3: [384, 384], 4: [384, 384], 5: [384, 384], 6: [256, 256],
    7: [512, 384], 8: [768, 384], 9: [512, 384], 10: [768, 768], 11: [518, 518], 12: [518, 518], 13: [518, 518],
    14: [518, 518],
}


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

    def update_settings(self, **kvargs):
        """reference :54-57 -- free-form settings (boost_rmax, precision, no_half, ...) become attributes."""
        for k, v in kvargs.items():
            setattr(self, k, v)

    def register_predictor(self, model_type, fn):
        """fn(pil_image, net_width, net_height, device) -> float32 tensor/ndarray [H,W] at image size (raw model output)."""
        self._predictors[model_type] = fn

    def ensure_models(self, model_type, device, boost: bool, tiling_mode: bool = False):
        """reference :60-74."""
        if boost:
            raise NotImplementedError("Boost (res101 + pix2pix merge) is not built yet")
        if model_type not in self._predictors:
            raise NotImplementedError(
                f"depth model {model_type!r} is not available in this build: the model families of SURVEY.md 8a "
                "(a10-a17) come after the per-pixel path; register a predictor with ModelHolder.register_predictor "
                "or pass precomputed depthmaps")
        self.depth_model = self._predictors[model_type]
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
        raw = self.depth_model(input, net_width, net_height, self.device)
        return raw, (self.depth_model_type in INVERTED_MODEL_IDS)

    def offload(self):
        self.offloaded = True

    def reload(self):
        self.offloaded = False

    def unload_models(self):
        self.depth_model = None
        self.pix2pix_model = None
        self.depth_model_type = None
        self.device = None
        gc.collect()
