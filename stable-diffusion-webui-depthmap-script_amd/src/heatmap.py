"""Heat map output of the funnel (reference: src/core.py:271-274 -> dzoedepth/utils/misc.py:97-150 ``colorize``).

``colorize(value, cmap='inferno')`` with the reference's defaults: vmin / vmax are the 2nd / 85th percentile of the depth
(np.percentile, linear interpolation), the normalised value goes through matplotlib's colormap with ``bytes=True``.
Here the percentiles are exact order statistics found on the device and the per-pixel mapping is one HIP pass
(ds_colorize_u16); only the 256-entry colour table comes from matplotlib, like in the reference.
"""
import numpy as np

_LUTS = {}


def colormap_table(cmap='inferno'):
    """The colormap's RGBA byte table [N, 4] (what Colormap.__call__(..., bytes=True) indexes), cached per name."""
    if cmap not in _LUTS:
        import matplotlib
        cm = matplotlib.colormaps[cmap] if hasattr(matplotlib, "colormaps") else matplotlib.cm.get_cmap(cmap)
        _LUTS[cmap] = np.ascontiguousarray(cm(np.arange(cm.N), bytes=True), dtype=np.uint8)
    return _LUTS[cmap]


def colorize_batch(depth_u16, cmap='inferno', lo=2.0, hi=85.0):
    """depth_u16: CUDA tensor [n, h, w] uint16 -> CUDA tensor [n, h, w, 4] uint8 (RGBA)."""
    from . import _native
    from .video_mode import _LOCAL, _global_percentiles
    torch = _native.require_gpu()
    lut = colormap_table(cmap)
    if lut.shape[0] > 256:
        raise NotImplementedError("colormaps with more than 256 entries are not built")
    # uint16 has few torch kernels: widen through the int16 bit pattern (exact)
    wide = (depth_u16.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF).to(torch.float32)
    vmm = [_global_percentiles(wide[i], [lo, hi], _LOCAL) for i in range(depth_u16.shape[0])]
    return _native.colorize_u16(depth_u16.contiguous(), torch.tensor(vmm, dtype=torch.float64),
                                torch.from_numpy(lut))
