"""Simple ("occluded") mesh output of the funnel (reference: src/core.py:277-306, create_mesh :740-773,
depth_edges_mask :724-737, pano_depth_to_world_points :695-721; dzoedepth/utils/geometry.py:27-98).

Per-pixel work -- back-projection of every depth sample through a 55-degree pinhole, two triangles per pixel quad, an
edge mask from the depth gradient -- done with device tensor ops in float64 in the reference's operation order (the
3x3 inverse intrinsics come from the same numpy call the reference makes), so vertices and faces match it bit for bit.
The reference hands vertices / faces / colours to `trimesh` and lets it write the .obj; trimesh is not part of this
build, so the file is written here (Wavefront OBJ, "v x y z r g b" vertex colours, 1-based "f a b c" faces): same
geometry, our own text formatting.
"""
import math
import os

import numpy as np


def get_intrinsics(h, w):
    """geometry.py:27-37: fov 55 degrees, central principal point."""
    f = 0.5 * w / np.tan(0.5 * 55 * np.pi / 180.0)
    return np.array([[f, 0, 0.5 * w], [0, f, 0.5 * h], [0, 0, 1]])


def depth_to_points(depth):
    """geometry.py:39-74 with R = I, t = 0.  depth: tensor [H, W] (any float dtype, any device) -> float64 [H, W, 3]."""
    import torch
    h, w = depth.shape
    kinv = np.linalg.inv(get_intrinsics(h, w))                 # the reference's own call (:42); 9 doubles, host side
    d = depth.to(torch.float64)
    u = torch.arange(w, device=depth.device, dtype=torch.float64).view(1, w).expand(h, w)
    v = torch.arange(h, device=depth.device, dtype=torch.float64).view(h, 1).expand(h, w)
    rows = []
    for i in range(3):                                         # (D * Kinv) @ (u, v, 1): products summed in index order (:66)
        acc = (d * float(kinv[i, 0])) * u
        acc = acc + (d * float(kinv[i, 1])) * v
        acc = acc + (d * float(kinv[i, 2])) * 1.0
        rows.append(acc)
    # M = diag(-1, -1, 1) (:51-53), then the identity rotation and zero translation (:70): exact sign flips
    return torch.stack((0.0 - rows[0], 0.0 - rows[1], rows[2]), dim=-1)


def pano_depth_to_world_points(depth):
    """core.py:695-721: equirectangular depth as radius.  -> float64 [H*W, 3]."""
    import torch
    h, w = depth.shape
    lon = torch.from_numpy(np.linspace(-np.pi, np.pi, w)).to(depth.device).view(1, w).expand(h, w).reshape(-1)
    lat = torch.from_numpy(np.linspace(-np.pi / 2, np.pi / 2, h)).to(depth.device).view(h, 1).expand(h, w).reshape(-1)
    r = depth.reshape(-1)
    r = r.to(torch.float64) if r.dtype != torch.float32 else r
    x = r * torch.cos(lat) * torch.cos(lon)
    y = r * torch.cos(lat) * torch.sin(lon)
    z = r * torch.sin(lat)
    return torch.stack([x, y, z], dim=1).to(torch.float64)


def _gradient(a, axis):
    """np.gradient along one axis, unit spacing: central differences inside, one-sided at the two ends."""
    import torch
    a = a.movedim(axis, 0)
    g = torch.empty_like(a)
    g[1:-1] = (a[2:] - a[:-2]) / 2.0
    g[0] = a[1] - a[0]
    g[-1] = a[-1] - a[-2]
    return g.movedim(0, axis)


def depth_edges_mask(depth):
    """core.py:724-737: gradient magnitude > 0.05 (in the dtype of the depth, like numpy)."""
    import torch
    dx, dy = _gradient(depth, 0), _gradient(depth, 1)
    return torch.sqrt(dx ** 2 + dy ** 2) > 0.05


def create_triangles(h, w, mask=None, device=None):
    """geometry.py:77-98: (tl, bl, tr) and (br, tr, bl) per pixel quad, row-major; with a mask, only triangles whose
    three vertices are all kept."""
    import torch
    y, x = torch.meshgrid(torch.arange(h - 1, device=device), torch.arange(w - 1, device=device), indexing='ij')
    tl = y * w + x
    tr = tl + 1
    bl = tl + w
    br = bl + 1
    tri = torch.stack([tl, bl, tr, br, tr, bl], dim=-1).reshape((w - 1) * (h - 1) * 2, 3)
    if mask is not None:
        m = mask.reshape(-1)
        tri = tri[m[tri].all(1)]
    return tri


def mesh_depth(depth, model_type, boost, custom_depth):
    """core.py:283-303: map the raw prediction to "sensible" distances for everything that is not plain ZoeDepth."""
    d = depth
    dmin, dmax = d.min(), d.max()
    if model_type not in (7, 8, 9) or boost or custom_depth:
        if model_type > 0 or custom_depth:
            d = dmax - d + dmin
        if float(dmin) < 0:
            d = d - dmin
        if float(d.max()) > 10.0:
            d = 4.0 * (d - dmin) / (dmax - dmin)
        d = d + 1.0
    return d


def create_mesh_arrays(image_u8, depth, keep_edges=False, spherical=False):
    """create_mesh (:740-773) up to the trimesh call: (vertices float64 [N,3], faces int64 [M,3], colours uint8 [N,3])."""
    import torch
    h, w = depth.shape
    if tuple(image_u8.shape[:2]) != (h, w):
        raise ValueError(f"image {tuple(image_u8.shape[:2])} and depth {(h, w)} differ (the reference only shrinks the IMAGE to "
                         "depthmap_script_mesh_maxsize and then fails inside trimesh)")
    verts = pano_depth_to_world_points(depth) if spherical else depth_to_points(depth).reshape(-1, 3)
    mask = None if keep_edges else ~depth_edges_mask(depth)
    faces = create_triangles(h, w, mask=mask, device=depth.device)
    if spherical:                                               # :765-771: rotate 90 degrees about X
        c, s = math.cos(math.pi / 2), math.sin(math.pi / 2)
        rot = torch.tensor([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]], dtype=torch.float64, device=verts.device)
        verts = verts @ rot.T
    return verts, faces, image_u8.reshape(-1, image_u8.shape[-1])[:, :3]


def unique_filename(outpath, basename, ext, suffix=''):
    """core.py:352-362 without the web UI's sequence bookkeeping: the first free '<basename>-NNNN[-suffix].<ext>'."""
    suffix = f'-{suffix}' if suffix else ''
    for i in range(100000):
        fn = os.path.join(outpath or '.', f"{basename}-{i:04}{suffix}.{ext}")
        if not os.path.exists(fn):
            return fn
    return f"{basename}-99999{suffix}.{ext}"


def write_obj(path, verts, faces, colors):
    v = np.asarray(verts, dtype=np.float64)
    c = np.asarray(colors, dtype=np.float64) / 255.0
    f = np.asarray(faces, dtype=np.int64) + 1
    with open(path, 'w') as fh:
        fh.write("# depthmap simple mesh: v x y z r g b / f a b c\n")
        np.savetxt(fh, np.concatenate([v, c], axis=1), fmt="v %.8f %.8f %.8f %.5f %.5f %.5f")
        np.savetxt(fh, f, fmt="f %d %d %d")
    return path
