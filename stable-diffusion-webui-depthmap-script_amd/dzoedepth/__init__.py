from .zoedepth import ZoeDepth, ZoeDepthNK, MidasCore, build_zoedepth  # noqa: F401
