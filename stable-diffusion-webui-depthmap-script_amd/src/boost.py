"""Boost ("BoostingMonocularDepth") multi-resolution merge, device resident and batched over patches.

Reference: src/depthmap_generation.py -- estimateboost :774-941, generatemask :944-953, rgb2gray :956-958,
resizewithpool :960-965, calculateprocessingres :969-1025, doubleestimate :1028-1049, singleestimate :1053-1066,
generatepatchs :1070-1099, applyGridpatch :1102-1116, adaptiveselection :1120-1167, getGF_fromintegral :1170-1177,
ImageandPatchs :562-608, impatch :663-670; pix2pix/models/pix2pix4depth_model.py:96-116.

What is different from the reference on purpose (results are the same up to float rounding):
  * every patch's double estimation depends only on the RGB patch and on the BASE estimate, never on the running blend
    (`estimation_base_image` is written once, :580-583,868), so all patches go through the depth network and the merge
    network as BATCHES; only the blend itself is order dependent;
  * the blend of all patches is ONE kernel launch (ds_boost_blend): polynomial mapping, cubic resize of the merged patch,
    bilinear resize of the Gaussian mask and `dst*(1-mask) + merged*mask` happen per output pixel, in patch order, with
    no temporaries -- the reference makes four full-size arrays and a read-modify-write of the estimate per patch;
  * image-processing steps that the reference does with OpenCV / scikit-image on the host (Sobel, resize, dilate,
    block_reduce, integral, GaussianBlur) are torch operations on the device, restated from the libraries' documented
    behaviour (cv2 / skimage are not installable here: PARITY UNPINNED for those steps; the patch-selection logic, the
    merge network and the depth networks are pinned against the reference's own code in tests/).
Note: `cv2.resize(grad, (n, n), cv2.INTER_AREA)` and `cv2.resize(x, (p, p), cv2.INTER_NEAREST)` in the reference pass
the flag in the `dst` slot (:988,1008), so both are plain bilinear resizes; reproduced.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _native
from . import vit_mi355x as vm

PIX2PIX_SIZE = 1024
MASK_SIZE = 3000
_mask_cache = {}


# ---- small image-processing helpers (torch restatements of the cv2 / skimage calls) --------------------------------------
def _resize(t, size_hw, mode):
    """t [..., H, W] -> [..., h, w]; cv2.resize semantics: half-pixel centres, no antialiasing, replicated border;
    mode 'bicubic' = INTER_CUBIC (a = -0.75), 'bilinear' = INTER_LINEAR."""
    lead = t.shape[:-2]
    x = t.reshape((-1, 1) + tuple(t.shape[-2:]))
    y = F.interpolate(x, size=(int(size_hw[0]), int(size_hw[1])), mode=mode, align_corners=False)
    return y.reshape(lead + (int(size_hw[0]), int(size_hw[1])))


def _resize_hwc(img, size_hw, mode):
    return _resize(img.permute(2, 0, 1), size_hw, mode).permute(1, 2, 0)


def rgb2gray(rgb):
    """:956-958"""
    return rgb[..., 0] * 0.2989 + rgb[..., 1] * 0.5870 + rgb[..., 2] * 0.1140


def sobel_abs_sum(gray):
    """|Sobel dy| + |Sobel dx|, ksize 3, float64, BORDER_REFLECT_101 (:987, :1073-1074)."""
    g = F.pad(gray[None, None], (1, 1, 1, 1), mode='reflect')
    kx = vm.device_constant((-1., 0., 1., -2., 0., 2., -1., 0., 1.), gray.device, gray.dtype).view(3, 3)
    gx = F.conv2d(g, kx[None, None])[0, 0]
    gy = F.conv2d(g, kx.t().contiguous()[None, None])[0, 0]
    return gy.abs() + gx.abs()


def _dilate_ones(x, k):
    """cv2.dilate with a k x k kernel of ones, anchor at (k//2, k//2), outside = -inf."""
    if k <= 1:
        return x
    lo, hi = k // 2, k - 1 - k // 2
    p = F.pad(x[None, None], (lo, hi, lo, hi), value=float('-inf'))
    return F.max_pool2d(p, kernel_size=k, stride=1)[0, 0]


def resizewithpool(img, size):
    """:960-965: skimage.measure.block_reduce(img, (n, n), np.max) (zero padded to a multiple of n)."""
    n = int(math.floor(img.shape[0] / size))
    h, w = img.shape
    ph, pw = (-h) % n, (-w) % n
    p = F.pad(img, (0, pw, 0, ph), value=0.0)
    # the maximum of every n x n block as a reduction over a [H/n, n, W/n, n] view (exact: a maximum rounds nothing) -- torch's float64
    # max_pool2d on a single-channel 2160 x 3840 plane took 1 ms per call, 13 calls per image of BASELINE config 4
    return p.view(p.shape[0] // n, n, p.shape[1] // n, n).amax(dim=(1, 3))


def generatemask(size, device):
    """:944-953: ones in the central 70 %, Gaussian blur (k = 2*ceil(2*s/16)+1, sigma = s/16), min-max normalised."""
    key = (size, str(device))
    if key in _mask_cache:
        return _mask_cache[key]
    mask = torch.zeros((size, size), dtype=torch.float32, device=device)
    sigma = int(size / 16)
    k = int(2 * math.ceil(2 * int(size / 16)) + 1)
    lo = int(0.15 * size)
    mask[lo:size - lo, lo:size - lo] = 1
    xs = torch.arange(k, dtype=torch.float64, device=device) - (k - 1) / 2
    g = torch.exp(-(xs * xs) / (2.0 * sigma * sigma))
    g = (g / g.sum()).float()                                # cv2.getGaussianKernel, CV_32F for a float32 image
    r = k // 2
    m = F.pad(mask[None, None], (r, r, r, r), mode='reflect')                      # BORDER_REFLECT_101
    m = F.conv2d(m, g.view(1, 1, 1, k))
    m = F.conv2d(m, g.view(1, 1, k, 1))[0, 0]
    m = (m - m.min()) / (m.max() - m.min())
    _mask_cache[key] = m.contiguous()
    return _mask_cache[key]


# ---- resolution search and patch selection -----------------------------------------------------------------------------------
def calculateprocessingres(img, basesize, confidence=0.1, scale_threshold=3, whole_size_threshold=3000):
    """:969-1025.  img: float64 [H, W, 3] on the device."""
    speed_scale = 32
    image_dim = int(min(img.shape[0:2]))
    grad = sobel_abs_sum(rgb2gray(img))
    grad = _resize(grad, (image_dim, image_dim), 'bilinear')
    m, M = grad.min(), grad.max()
    middle = m + (0.4 * (M - m))
    grad = (grad >= middle).to(grad.dtype)
    k1 = int(basesize / speed_scale)
    k2 = int(basesize / (4 * speed_scale))
    threshold = min(whole_size_threshold, scale_threshold * max(img.shape[:2]))
    outputsize_scale = basesize / speed_scale
    grad_resized = None
    for p_size in range(int(basesize / speed_scale), int(threshold / speed_scale), int(basesize / (2 * speed_scale))):
        grad_resized = resizewithpool(grad, p_size)
        grad_resized = _resize(grad_resized, (p_size, p_size), 'bilinear')
        grad_resized = (grad_resized >= 0.5).to(grad.dtype)
        dilated = _dilate_ones(grad_resized, k1)
        meanvalue = float((1 - dilated).mean())
        if meanvalue > confidence:
            break
        outputsize_scale = p_size
    if grad_resized is None:                                 # the reference would raise NameError here (:1021)
        raise ValueError("Boost: image too small for the resolution search")
    patch_scale = float(_dilate_ones(grad_resized, k2).mean())
    return int(outputsize_scale * speed_scale), patch_scale


def applyGridpatch(blsize, stride, img_shape, box):
    """:1102-1116"""
    counter1 = 0
    patch_bound_list = {}
    for k in range(blsize, img_shape[1] - blsize, stride):
        for j in range(blsize, img_shape[0] - blsize, stride):
            patchbounds = [j - blsize, k - blsize, j - blsize + 2 * blsize, k - blsize + 2 * blsize]
            patch_bound = [box[0] + patchbounds[1], box[1] + patchbounds[0], patchbounds[3] - patchbounds[1],
                           patchbounds[2] - patchbounds[0]]
            patch_bound_list[str(counter1)] = {'rect': patch_bound, 'size': patch_bound[2]}
            counter1 += 1
    return patch_bound_list


def getGF_fromintegral(integralimage, rect):
    """:1170-1177"""
    x1, x2 = rect[1], rect[1] + rect[3]
    y1, y2 = rect[0], rect[0] + rect[2]
    return integralimage[x2, y2] - integralimage[x1, y2] - integralimage[x2, y1] + integralimage[x1, y1]


def adaptiveselection(integral_grad, patch_bound_list, gf, factor):
    """:1120-1167 (host logic on the integral image)."""
    patchlist = {}
    count = 0
    height, width = integral_grad.shape
    search_step = int(32 / factor)
    for c in range(len(patch_bound_list)):
        bbox = patch_bound_list[str(c)]['rect']
        cgf = getGF_fromintegral(integral_grad, bbox) / (bbox[2] * bbox[3])
        if cgf >= gf:
            bbox_test = bbox.copy()
            while True:
                bbox_test[0] = bbox_test[0] - int(search_step / 2)
                bbox_test[1] = bbox_test[1] - int(search_step / 2)
                bbox_test[2] = bbox_test[2] + search_step
                bbox_test[3] = bbox_test[3] + search_step
                if bbox_test[0] < 0 or bbox_test[1] < 0 or bbox_test[1] + bbox_test[3] >= height \
                        or bbox_test[0] + bbox_test[2] >= width:
                    break
                cgf = getGF_fromintegral(integral_grad, bbox_test) / (bbox_test[2] * bbox_test[3])
                if cgf < gf:
                    break
                bbox = bbox_test.copy()
            patchlist[str(count)] = {'rect': bbox, 'size': bbox[2]}
            count += 1
    return patchlist


def generatepatchs(img, base_size, factor):
    """:1070-1099.  img: float64 [H, W, 3] on the device.  Returns [(id, {'rect', 'size'})] sorted largest first."""
    whole_grad = sobel_abs_sum(rgb2gray(img))
    threshold = whole_grad[whole_grad > 0].mean()
    whole_grad = torch.where(whole_grad < threshold, torch.zeros_like(whole_grad), whole_grad)
    gf = float(whole_grad.sum()) / whole_grad.numel()
    integ = torch.zeros((whole_grad.shape[0] + 1, whole_grad.shape[1] + 1), dtype=torch.float64, device=img.device)
    integ[1:, 1:] = whole_grad.cumsum(0).cumsum(1)                                # cv2.integral
    integ = integ.cpu().numpy()
    blsize = int(round(base_size / 2))
    stride = int(round(blsize * 0.75))
    patch_bound_list = applyGridpatch(blsize, stride, img.shape, [0, 0, 0, 0])
    patch_bound_list = adaptiveselection(integ, patch_bound_list, gf, factor)
    return sorted(patch_bound_list.items(), key=lambda x: x[1]['size'], reverse=True)


# ---- depth network on patches -------------------------------------------------------------------------------------------------
def _single_estimates(patches, msize, net, model_type, chunk):
    """singleestimate (:1053-1066) for a list of float64 [h, w, 3] patches (channel order as get_raw_prediction leaves it:
    R and B swapped).  Returns a list of float32 [h, w] predictions at patch size."""
    outs = []
    dev = patches[0].device
    if model_type == 0:                                      # estimateleres (:406-421)
        mean = vm.device_constant(vm.IMAGENET_MEAN, dev).view(1, 3, 1, 1)
        std = vm.device_constant(vm.IMAGENET_STD, dev).view(1, 3, 1, 1)
        for s in range(0, len(patches), chunk):
            part = patches[s:s + chunk]
            x = torch.stack([_resize(p.flip(-1).permute(2, 0, 1), (msize, msize), 'bilinear') for p in part]).float()
            pred = net.depth_model((x - mean) / std).float()[:, 0]
            outs += [_resize(pred[i], part[i].shape[:2], 'bicubic') for i in range(len(part))]
        return outs
    if model_type in (12, 13, 14):                           # estimatedepthanything_v2 (:548-559)
        for p in patches:
            u8 = (p * 255.1).clamp(0, 255).to(torch.uint8).flip(-1)    # (:550) the second channel swap; net swaps again
            outs.append(net.infer_batch(u8.unsqueeze(0), int(msize))[0])
        return outs
    if model_type in (1, 2, 3, 4):                           # estimatemidasBoost (:1180-1220)
        from dmidas.dpt_depth import midas_net_size
        mean = vm.device_constant(vm.IMAGENET_MEAN, dev).view(1, 3, 1, 1)      # ImageNet statistics for EVERY MiDaS
        std = vm.device_constant(vm.IMAGENET_STD, dev).view(1, 3, 1, 1)       # model here, unlike estimatemidas
        for p in patches:
            h, w = p.shape[:2]
            nw, nh = midas_net_size(w, h, msize, msize, "upper_bound")              # keep aspect, multiple of 32 (:1184-1191)
            x = _resize(p.permute(2, 0, 1), (nh, nw), 'bicubic').unsqueeze(0).float()
            pred = net((x - mean) / std).float()[0]                                   # float32: Boost never runs half (:271)
            pred = _resize(pred, (h, w), 'bicubic')                                   # cv2.resize INTER_CUBIC (:1209)
            lo, hi = pred.min(), pred.max()
            if float(hi - lo) > float(np.finfo("float").eps):                         # :1212-1218
                pred = (pred - lo) / (hi - lo)
            else:
                pred = torch.zeros_like(pred)
            outs.append(pred)
        return outs
    if model_type in (7, 8, 9):                              # estimatezoedepth on np.uint8(img * 255) (:1062-1064)
        for p in patches:
            u8 = (p * 255).to(torch.uint8)                   # truncation like np.uint8; no channel swap on this path
            outs.append(net.infer_batch(u8.unsqueeze(0), int(msize), int(msize))[0])
        return outs
    raise NotImplementedError(f"Boost with depth model id {model_type} is not built (built: 0 LeReS, 1-4 MiDaS DPT, 7-9 ZoeDepth, "
                              "12-14 Depth-Anything-V2)")


def doubleestimate(patches, size1, size2, net, model_type, pix2pix, chunk=8):
    """:1028-1049 for a list of patches: low-res + high-res estimate -> merge network -> min-max normalised
    float32 [P, 1024, 1024]."""
    # both estimates in batches of `chunk` CONSECUTIVE patches: the batch composition is then a function of the patch list
    # alone, whatever the number of ranks the list is sharded over (estimateboost shards in whole chunks)
    e1 = _single_estimates(patches, size1, net, model_type, chunk)
    e2 = _single_estimates(patches, size2, net, model_type, chunk)
    outs = []
    for s in range(0, len(patches), chunk):
        a = torch.stack([_resize(t, (PIX2PIX_SIZE, PIX2PIX_SIZE), 'bicubic') for t in e1[s:s + chunk]])
        b = torch.stack([_resize(t, (PIX2PIX_SIZE, PIX2PIX_SIZE), 'bicubic') for t in e2[s:s + chunk]])
        m = pix2pix.merge(a, b)
        mn = m.amin(dim=(-2, -1), keepdim=True)
        mx = m.amax(dim=(-2, -1), keepdim=True)
        outs.append((m - mn) / (mx - mn))
    return torch.cat(outs)


def _polyfit1(x, y):
    """np.polyfit(x, y, deg=1) over the last two dims, float64 closed form.  x, y: [P, S, S] -> (p0, p1) each [P]."""
    x = x.double().flatten(1)
    y = y.double().flatten(1)
    n = x.shape[1]
    sx, sy = x.sum(1), y.sum(1)
    sxx, sxy = (x * x).sum(1), (x * y).sum(1)
    den = n * sxx - sx * sx
    p0 = (n * sxy - sx * sy) / den
    p1 = (sy - p0 * sx) / n
    return p0, p1


# ---- the pipeline -----------------------------------------------------------------------------------------------------------------
def _patch_merge(patches, base_patches, rf, patch_netsize, net, model_type, pix2pix, chunk):
    """Per-patch half of estimateboost (:879-915) for a list of patches: double estimation, merge against the base patch,
    degree-1 polyfit.  -> (merged [P, 1024, 1024] float32, coefficients [P, 2] float64).  Depends only on the patch and
    on the base estimate, never on the running blend: this is what shards over GPUs."""
    dev = patches[0].device
    est = doubleestimate(patches, rf, patch_netsize, net, model_type, pix2pix, chunk)                          # :887-888
    mapped_all, coefs = [], []
    for s in range(0, len(patches), chunk):
        b1024 = torch.stack([_resize(t, (PIX2PIX_SIZE, PIX2PIX_SIZE), 'bicubic') for t in base_patches[s:s + chunk]])
        mapped = pix2pix.merge(b1024, est[s:s + chunk])                                                        # :896-909
        p0, p1 = _polyfit1(mapped, b1024)                                                                      # :915
        mapped_all.append(mapped)
        coefs.append(torch.stack((p0, p1), dim=1))
    return torch.cat(mapped_all).to(dev), torch.cat(coefs).to(dev)


def _gather_ragged(local, counts, group, dst):
    """One gather of per-rank blocks of different lengths (counts[r] rows on rank r), padded to the longest."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = max(max(counts), 1)
    send = local.new_zeros((per,) + tuple(local.shape[1:]))
    send[:local.shape[0]] = local
    from .multigpu import global_rank
    gdst = global_rank(group, dst)          # `dst` is a rank of the group; the collective wants the global rank
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=gdst, group=group)
        return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0)
    dist.gather(send, None, dst=gdst, group=group)
    return None


@torch.no_grad()
def estimateboost(image_u8, net, model_type, pix2pix, whole_size_threshold=1600, chunk=8, stats=None, group=None, dst=0,
                  blend=None, trace=None):
    """:774-941.  image_u8: uint8 [H, W, 3] on the device, channel order as the funnel hands it over (RGB).
    Returns the boosted float32 [H, W] prediction (device).

    group: a torch.distributed process group (one process per GPU) to shard the PATCHES over (BASELINE config 4,
    SURVEY.md 8e).  Rank `dst` runs the whole-image passes and the patch selection and broadcasts the base estimate and
    the patch rectangles; every rank then renders a contiguous run of whole chunks of the ordered patch list (double
    estimation + merge network + polyfit: all that depends on the patch and the base only), ONE gather brings the merged
    1024^2 patches and their coefficients to `dst`, and `dst` blends them in the original order (:1098, :936) with one
    launch.  Chunk-aligned shards keep every network batch identical to the single-rank run, so the result does not
    depend on the number of ranks.  Ranks other than `dst` return None.
    blend: test hook replacing ds_boost_blend (CPU runs of the sharding logic); the product always uses the HIP kernel.
    trace: optional dict that receives the stage outputs (float32 / float64 CPU copies): the whole-image double estimate, the
    base at merge size, the merged patches and their polyfit coefficients, the blended estimate -- what the per-stage error
    budget of the GPU test compares between the device run and its CPU twin."""
    import torch.distributed as dist
    multi = group is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if multi else 0
    world = dist.get_world_size(group) if multi else 1
    if blend is None:
        _native.require_gpu()
        blend = _native.boost_blend
    rf = {0: 448, 1: 512, 11: 518, 12: 518, 13: 518, 14: 518}.get(model_type, 384)          # :777-786
    patch_netsize = 2 * rf
    dev = image_u8.device
    img = image_u8.flip(-1).double() / 255.0                 # get_raw_prediction: cvtColor(BGR2RGB) / 255 (:381)
    H, W = img.shape[:2]

    head = torch.zeros(8, dtype=torch.int64, device=dev)
    rect_t = None
    if rank == dst:
        whole_image_optimal_size, patch_scale = calculateprocessingres(img, rf, 0.2, 3, whole_size_threshold)
        whole_estimate = doubleestimate([img], rf, whole_image_optimal_size, net, model_type, pix2pix, chunk)[0]
        factor = max(min(1, 4 * patch_scale * whole_image_optimal_size / whole_size_threshold), 0.2)       # :819
        if H > W:
            a = 2 * whole_image_optimal_size
            b = round(2 * whole_image_optimal_size * W / H)
        else:
            a = round(2 * whole_image_optimal_size * H / W)
            b = 2 * whole_image_optimal_size
        b, a = int(round(b / factor)), int(round(a / factor))
        img_r = _resize_hwc(img, (a, b), 'bicubic')                                                           # :846
        patchset = generatepatchs(img_r, rf * 2, factor)
        rect_t = torch.tensor([list(info['rect']) for _, info in patchset], dtype=torch.int64, device=dev).reshape(-1, 4)
        head[0], head[1], head[2] = a, b, rect_t.shape[0]
    if multi:
        from .multigpu import global_rank
        gsrc = global_rank(group, dst)       # `dst` / `rank` are ranks of the group; broadcast's src is a global rank
        dist.broadcast(head, src=gsrc, group=group)
        a, b = int(head[0]), int(head[1])
        if rank != dst:
            rect_t = torch.zeros((int(head[2]), 4), dtype=torch.int64, device=dev)
            img_r = _resize_hwc(img, (a, b), 'bicubic')
        if rect_t.shape[0]:
            dist.broadcast(rect_t, src=gsrc, group=group)
    mergein_scale = H / img_r.shape[0]                                                                         # :866
    size_m = (round(img_r.shape[0] * mergein_scale), round(img_r.shape[1] * mergein_scale))
    rgb_image = _resize_hwc(img_r, size_m, 'bicubic')
    if rank == dst:
        base = _resize(whole_estimate, size_m, 'bicubic').float().contiguous()
        if trace is not None:
            trace["whole_estimate"], trace["base"] = whole_estimate.float().cpu(), base.cpu()
    else:
        base = torch.empty(size_m, dtype=torch.float32, device=dev)
    if multi:
        dist.broadcast(base, src=gsrc, group=group)          # every rank merges against the SAME base estimate
    dst_img = base.clone() if rank == dst else None

    rects, patches, base_patches = [], [], []
    for rect in rect_t.tolist():
        r = np.round(np.array(rect) * mergein_scale).astype('int')                                            # :595-597
        w1, h1, w2, h2 = int(r[0]), int(r[1]), int(r[0] + r[2]), int(r[1] + r[3])
        prgb = rgb_image[max(h1, 0):h2, max(w1, 0):w2]
        pbase = base[max(h1, 0):h2, max(w1, 0):w2]
        if prgb.shape[0] == 0 or prgb.shape[1] == 0:
            continue
        rects.append((max(w1, 0), max(h1, 0), pbase.shape[1], pbase.shape[0]))
        patches.append(prgb)
        base_patches.append(pbase)
    if stats is not None and rank == dst:
        stats.update({"whole_image_optimal_size": whole_image_optimal_size, "patch_scale": patch_scale, "factor": factor,
                      "target": (a, b), "patches": len(rects), "ranks": world})
    if rects:
        # whole chunks of the ordered list, contiguous per rank
        from .multigpu import shard_bounds
        n_chunks = (len(rects) + chunk - 1) // chunk
        bounds = [(min(s * chunk, len(rects)), min(e * chunk, len(rects))) for s, e in shard_bounds(n_chunks, world)]
        s, e = bounds[rank]
        if e > s:
            mapped, coef = _patch_merge(patches[s:e], base_patches[s:e], rf, patch_netsize, net, model_type, pix2pix, chunk)
        else:
            mapped = torch.zeros((0, PIX2PIX_SIZE, PIX2PIX_SIZE), dtype=torch.float32, device=dev)
            coef = torch.zeros((0, 2), dtype=torch.float64, device=dev)
        if multi:
            # ONE collective on the data path: a patch travels as its 1024^2 merged estimate followed by the two float64
            # polyfit coefficients (four float32 words, bit patterns untouched)
            counts = [b1 - b0 for b0, b1 in bounds]
            payload = torch.cat((mapped.flatten(1), coef.contiguous().view(torch.float32).reshape(-1, 4)), dim=1)
            payload = _gather_ragged(payload.contiguous(), counts, group, dst)
            if rank == dst:
                n_px = PIX2PIX_SIZE * PIX2PIX_SIZE
                mapped = payload[:, :n_px].reshape(-1, PIX2PIX_SIZE, PIX2PIX_SIZE)
                coef = payload[:, n_px:].contiguous().view(torch.float64).reshape(-1, 2)
        if rank == dst:
            if trace is not None:
                trace["mapped"], trace["coef"] = mapped[:, ::4, ::4].float().cpu(), coef.double().cpu()
            blend(dst_img, rects, [tuple(c) for c in coef.tolist()], mapped, generatemask(MASK_SIZE, dev))     # :916-937
            if trace is not None:
                trace["blended"] = dst_img.float().cpu()
    if rank != dst:
        return None
    return _resize(dst_img, (H, W), 'bicubic')                                                                # :940
