#!/usr/bin/env python3
"""Golden heat maps from the REFERENCE's own ``colorize`` (dzoedepth/utils/misc.py:97-150), as the funnel calls it
(src/core.py:271-274: ``colorize(img_output, cmap='inferno')`` on the uint16 depth).

Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_heatmap.py

The reference module is imported unmodified; outside shims: ``torchvision`` (absent, only imported at module top) is a
MagicMock, and ``matplotlib.cm.get_cmap`` (removed in matplotlib 3.9; the reference calls it at :133) is pointed at the
registry that replaced it.  Output: heatmap_cases.npz (depth inputs, RGBA outputs, and the colormap's byte table so that
the CPU tests do not depend on the matplotlib version of the box they run on).
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def main():
    for m in ('torchvision', 'torchvision.transforms'):
        sys.modules[m] = MagicMock()
    import matplotlib
    import matplotlib.cm
    if not hasattr(matplotlib.cm, 'get_cmap'):
        matplotlib.cm.get_cmap = lambda name: matplotlib.colormaps[name]
    sys.path.insert(0, REF)
    from dzoedepth.utils.misc import colorize

    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:48, 0:80]
    cases = {
        'ramp': ((xx * 65535) // 79).astype(np.uint16),
        'noise': rng.integers(0, 65536, (48, 80), dtype=np.uint16),
        'narrow': (30000 + rng.integers(0, 7, (48, 80))).astype(np.uint16),
        'flat': np.full((48, 80), 1234, np.uint16),
        'two_level': np.where(xx < 70, 100, 60000).astype(np.uint16),        # 85th percentile == 2nd percentile == 100
        'smooth': (32768 + 30000 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.uint16),
        'odd': rng.integers(0, 65536, (37, 53), dtype=np.uint16),
    }
    out = {}
    for name, d in cases.items():
        out[f'{name}__depth'] = d
        out[f'{name}__rgba'] = np.asarray(colorize(d.copy(), cmap='inferno'))
    cm = matplotlib.colormaps['inferno']
    out['inferno_lut'] = np.asarray(cm(np.arange(cm.N), bytes=True), dtype=np.uint8)
    out['matplotlib_version'] = np.array(matplotlib.__version__)
    np.savez_compressed(os.path.join(HERE, 'heatmap_cases.npz'), **out)
    print('wrote', len(cases), 'cases;', {k: v.shape for k, v in out.items() if k.endswith('__rgba')})


if __name__ == '__main__':
    main()
