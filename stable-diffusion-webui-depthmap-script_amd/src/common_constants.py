"""Option names and defaults accepted by ``core_generation_funnel``.

The option NAMES, their ORDER and their DEFAULTS are the boundary contract with the reference
(its ``GenerationOptions`` enum, src/common_constants.py:4-66): callers pass dicts keyed by the lower-cased
names, and ``CoreGenerationFunnelInp`` fills whatever is missing from ``.df``.  The table below is that
contract; the enum is generated from it.  The ``scope`` column says what this build does with the
option: "hot" = implemented on the MI355X path, "host" = honoured by host code, "out" = belongs to a
subsystem that is out of scope (rembg, meshes, heatmap) and raises if switched on.
"""
import enum

_OPTION_TABLE = (
    # name, default, scope
    ("COMPUTE_DEVICE", "GPU", "host"),
    ("MODEL_TYPE", "Depth Anything v2 Base", "host"),
    ("BOOST", False, "out"),
    ("NET_SIZE_MATCH", False, "host"),
    ("NET_WIDTH", 448, "host"),
    ("NET_HEIGHT", 448, "host"),
    ("TILING_MODE", False, "host"),
    ("DO_OUTPUT_DEPTH", True, "hot"),
    ("OUTPUT_DEPTH_INVERT", False, "hot"),
    ("OUTPUT_DEPTH_COMBINE", False, "hot"),
    ("OUTPUT_DEPTH_COMBINE_AXIS", "Horizontal", "hot"),
    ("DO_OUTPUT_DEPTH_PREDICTION", False, "hot"),
    ("CLIPDEPTH", False, "hot"),
    ("CLIPDEPTH_MODE", "Range", "hot"),
    ("CLIPDEPTH_FAR", 0.0, "hot"),
    ("CLIPDEPTH_NEAR", 1.0, "hot"),
    ("GEN_STEREO", False, "hot"),
    ("STEREO_MODES", ["left-right", "red-cyan-anaglyph"], "hot"),
    ("STEREO_DIVERGENCE", 2.5, "hot"),
    ("STEREO_SEPARATION", 0.0, "hot"),
    ("STEREO_FILL_ALGO", "polylines_sharp", "hot"),
    ("STEREO_OFFSET_EXPONENT", 1.0, "hot"),
    ("STEREO_BALANCE", 0.0, "hot"),
    ("GEN_NORMALMAP", False, "hot"),
    ("NORMALMAP_PRE_BLUR", False, "hot"),
    ("NORMALMAP_PRE_BLUR_KERNEL", 3, "hot"),
    ("NORMALMAP_SOBEL", True, "hot"),
    ("NORMALMAP_SOBEL_KERNEL", 3, "hot"),
    ("NORMALMAP_POST_BLUR", False, "hot"),
    ("NORMALMAP_POST_BLUR_KERNEL", 3, "hot"),
    ("NORMALMAP_INVERT", False, "hot"),
    ("GEN_HEATMAP", False, "out"),
    ("GEN_SIMPLE_MESH", False, "out"),
    ("SIMPLE_MESH_OCCLUDE", True, "out"),
    ("SIMPLE_MESH_SPHERICAL", False, "out"),
    ("GEN_INPAINTED_MESH", False, "out"),
    ("GEN_INPAINTED_MESH_DEMOS", False, "out"),
    ("GEN_REMBG", False, "out"),
    ("SAVE_BACKGROUND_REMOVAL_MASKS", False, "out"),
    ("PRE_DEPTH_BACKGROUND_REMOVAL", False, "out"),
    ("REMBG_MODEL", "u2net", "out"),
)


class _OptionBase(enum.Enum):
    """Members carry ``.df`` (default value) and ``.scope``; ``.value`` is the 1-based position."""

    @property
    def df(self):
        return _DEFAULTS[self.name]

    @property
    def scope(self):
        return _SCOPES[self.name]


GenerationOptions = _OptionBase("GenerationOptions", [(name, i + 1) for i, (name, _, _) in enumerate(_OPTION_TABLE)])
GenerationOptions.__doc__ = "Options consumed by core_generation_funnel; use these to avoid typos."
_DEFAULTS = {name: df for name, df, _ in _OPTION_TABLE}
_SCOPES = {name: sc for name, _, sc in _OPTION_TABLE}
