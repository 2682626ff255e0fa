"""Shared helpers for the tests: golden fixtures and synthetic inputs (no reference tree needed)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def load_stereo_golden():
    z = np.load(os.path.join(GOLDEN, "stereo_cases.npz"))
    index = json.loads(bytes(z["__index__"]).decode())
    return z, index


def load_funnel_golden():
    z = np.load(os.path.join(GOLDEN, "funnel_cases.npz"))
    index = json.loads(bytes(z["__index__"]).decode())
    return z, index


def load_normalmap_golden():
    """tests/golden/make_golden_normalmap.py: outputs of the reference's own create_normalmap (stub cv2, see there)."""
    z = np.load(os.path.join(GOLDEN, "normalmap_cases.npz"))
    index = json.loads(bytes(z["__index__"]).decode())
    return z, index


def golden_inputs(case):
    import make_golden as mg        # pure-numpy generators; importing it does not touch /root/reference
    return mg.gen_inputs(case)


def survey_inputs(H, W, seed, n=1):
    """SURVEY.md Appendix A / section 8(d) synthetic input (integer-only depth pattern)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.int64)
    d = (xx * 30000) // (W - 1) + ((xx // 8 + yy // 8) % 2) * 8000
    d[H // 4: H // 2, W // 3: 2 * W // 3] = 60000
    d[(3 * H) // 4:, : W // 5] = 1000
    dep = np.repeat(d.astype(np.uint16)[None], n, axis=0)
    return img, dep


def smooth_depth(H, W, seed):
    """Smooth float field with a few occluders -- stands in for a model prediction."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    f = np.zeros((H, W))
    for _ in range(6):
        cx, cy, s, a = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(W / 16, W / 3), rng.uniform(-1, 1)
        f += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    f += 0.3 * xx / W
    x0, y0 = int(rng.uniform(0, W / 2)), int(rng.uniform(0, H / 2))
    f[y0:y0 + H // 4, x0:x0 + W // 3] += 1.0
    return f.astype(np.float32)
