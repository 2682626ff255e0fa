#!/usr/bin/env python3
"""Golden geometry for the simple-mesh output from the REFERENCE's own functions: dzoedepth/utils/geometry.py
(depth_to_points, create_triangles; numpy only, imported as is) and src/core.py (depth_edges_mask,
pano_depth_to_world_points; core imported with the absent third-party modules stubbed, like make_golden.py does).
Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mesh.py   -> mesh_cases.npz
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference')


def main():
    np.float_ = np.float64
    import make_golden
    core = make_golden.load_reference_core()
    from dzoedepth.utils.geometry import create_triangles, depth_to_points
    rng = np.random.default_rng(9)
    out = {}
    for name, (h, w, dt) in {"a": (13, 17, np.float32), "b": (9, 31, np.float64), "c": (24, 24, np.float32)}.items():
        yy, xx = np.mgrid[0:h, 0:w]
        d = (1.0 + 0.12 * np.sin(xx / 3.0) + 0.08 * np.cos(yy / 2.0) + (xx > w // 2) * 0.8 + rng.random((h, w)) * 0.004).astype(dt)
        out[f"{name}__depth"] = d
        out[f"{name}__points"] = depth_to_points(d[None])
        out[f"{name}__pano"] = core.pano_depth_to_world_points(d)
        m = core.depth_edges_mask(d)
        out[f"{name}__edges"] = m
        out[f"{name}__tri_all"] = create_triangles(h, w)
        out[f"{name}__tri_masked"] = create_triangles(h, w, mask=~m)
    np.savez_compressed(os.path.join(HERE, 'mesh_cases.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
