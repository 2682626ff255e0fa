"""Drop-in for the reference's ``ddepth_anything_v2`` package (src/depthmap_generation.py:30: ``from
ddepth_anything_v2 import DepthAnythingV2``): same class, same constructor, same checkpoint key names, MI355X-first
forward (padded token sequence, fused MFMA attention, V produced transposed by its GEMM)."""
from .depth_anything_v2.dpt import DepthAnythingV2  # noqa: F401
