"""Drop-in for the hot path of the reference's ``src/core.py``: ``core_generation_funnel`` (alias
``run_depthmap``), ``CoreGenerationFunnelInp``, ``convert_to_i16``, ``convert_i16_to_rgb``.

Reference: src/core.py:44-58 (i16 helpers), :61-80 (option bundle), :83-349 (the funnel generator).
Same signature, same yielded ``(input_index, kind, result)`` triples in the same order, same option
names/defaults (``common_constants.GenerationOptions``).  What differs is where the pixels are made:
depth post-processing, uint16 quantisation, stereo views and normal maps run as HIP kernels on an
MI355X (``libdepthstereo_hip.so``) and stay in HBM between stages; host code only converts to PIL at
the moment a result is yielded.

Out of scope here: background removal, the inpainted (3D-photo) mesh and
video assembly.  Asking for them raises NotImplementedError instead of silently doing nothing.
"""
import gc

import numpy as np
from PIL import Image

from . import _native
from .common_constants import GenerationOptions
from .common_constants import GenerationOptions as go
from .depthmap_generation import ModelHolder
from .normalmap_generation import create_normalmap_batch
from .stereoimage_generation import create_stereoimages_batch

SCRIPT_NAME = "DepthMap (MI355X-native hot path)"

model_holder = ModelHolder()

_OUT_OF_SCOPE = (go.GEN_REMBG, go.GEN_INPAINTED_MESH)


def convert_to_i16(arr):
    """reference: src/core.py:44-50 -- clip(arr*65536 + 1e-4, 0, 65535.9) truncated to uint16, on the device.
    ndarray in -> ndarray out; float32 input is processed in float32, anything else in float64 (numpy rules)."""
    torch = _native.require_gpu()
    a = np.asarray(arr)
    if a.dtype != np.float32:
        a = a.astype(np.float64)
    t = torch.from_numpy(np.array(a, order='C')).cuda()
    return _native.convert_to_i16(t).cpu().numpy()


def convert_i16_to_rgb(image, like):
    """reference: src/core.py:52-58 -- three 8-bit channels holding image/256 (truncated)."""
    output = np.zeros_like(like)
    v = (np.asarray(image) / 256.0)
    output[:, :, 0] = v
    output[:, :, 1] = v
    output[:, :, 2] = v
    return output


class CoreGenerationFunnelInp:
    """Option bundle of the funnel (reference: src/core.py:61-80).  Keys are lower-cased option names or
    GenerationOptions members; unknown keys are dropped silently, missing ones take the enum default."""

    def __init__(self, values):
        if isinstance(values, CoreGenerationFunnelInp):
            values = values.values
        lowered = {}
        for k, v in values.items():
            lowered[(k.name if isinstance(k, GenerationOptions) else k).lower()] = v
        self.values = {}
        for setting in GenerationOptions:
            name = setting.name.lower()
            self.values[name] = lowered[name] if name in lowered else setting.df

    def __getitem__(self, item):
        if isinstance(item, GenerationOptions):
            return self.values[item.name.lower()]
        return self.values[item]

    def __getattr__(self, item):
        if item == 'values':
            raise AttributeError(item)
        return self[item]


def _custom_depth_to_float(dp, image):
    """Custom depthmap ingest (reference: src/core.py:145-174).  Host code: PIL resampling must be PIL's."""
    if isinstance(dp, Image.Image):
        if dp.width != image.width or dp.height != image.height:
            try:
                dp = dp.resize((image.width, image.height), Image.Resampling.LANCZOS)
            except Exception:
                dp = dp.resize((image.width, image.height))
        if len(dp.getbands()) == 1:
            out = np.asarray(dp, dtype="float")
            out_max = out.max()
            if out_max < 256:
                bit_depth = 8
            elif out_max < 65536:
                bit_depth = 16
            else:
                bit_depth = 32
            out = out / (2.0 ** bit_depth)
        else:
            out = np.asarray(dp, dtype="float")[:, :, 0]
            out = out / 256.0
        return out
    out = np.asarray(dp, dtype="float")
    assert image.height == out.shape[0], "Custom depthmap height mismatch"
    assert image.width == out.shape[1], "Custom depthmap width mismatch"
    return out


def _postprocess_prediction(pred, invert, inp):
    """Depth post-processing of src/core.py:189-206 on the device.
    pred: float32 cuda tensor [H,W].  Returns (out_f32 [H,W] in [0,1], prediction_copy or None)."""
    torch = _native._torch()
    pred = pred.contiguous()
    pmin, pmax = pred.min(), pred.max()
    if not bool(abs(float(pmax) - float(pmin)) > np.finfo("float").eps):
        return torch.zeros(pred.shape, dtype=torch.float64, device=pred.device), None, True
    out = pred.clone()
    if invert:
        out = out * -1
    prediction_copy = out
    if inp[go.CLIPDEPTH]:
        if inp[go.CLIPDEPTH_MODE] == 'Range':
            out = (out - out.min()) / (out.max() - out.min())
            out = torch.clamp(out, min=float(inp[go.CLIPDEPTH_FAR]), max=float(inp[go.CLIPDEPTH_NEAR]))
        elif inp[go.CLIPDEPTH_MODE] == 'Outliers':                                              # :199-201
            # np.percentile on the device: exact order statistics by bisection on the float32 bit pattern, numpy's
            # linear interpolation; np.clip(float32 array, float64 bounds) promotes to float64 (NumPy >= 2)
            from .video_mode import _global_percentiles
            fb, nb = _global_percentiles(out, [float(inp[go.CLIPDEPTH_FAR]) * 100.0, float(inp[go.CLIPDEPTH_NEAR]) * 100.0], None)
            out = torch.clamp(out.double(), min=fb, max=nb)
    return out, prediction_copy, False


def core_generation_funnel(outpath, inputimages, inputdepthmaps, inputnames, inp, ops=None):
    """Generator yielding ``(input_index, kind, result)`` (reference: src/core.py:83-349)."""
    if len(inputimages) == 0 or inputimages[0] is None:
        return
    if inputdepthmaps is None or len(inputdepthmaps) == 0:
        inputdepthmaps = [None for _ in range(len(inputimages))]
    inputdepthmaps_complete = all([x is not None for x in inputdepthmaps])

    inp = CoreGenerationFunnelInp(inp)
    for opt in _OUT_OF_SCOPE:
        if inp[opt]:
            raise NotImplementedError(f"{opt.name} belongs to a subsystem that is out of scope of the MI355X hot path "
                                      "(SURVEY.md section 8); use the reference for it")
    if ops is None:
        ops = {}
    model_holder.update_settings(**ops)

    torch = _native.require_gpu()        # the hot path has no CPU implementation, whatever COMPUTE_DEVICE says
    device = torch.device('cuda', torch.cuda.current_device())

    try:
        if not inputdepthmaps_complete:
            model_holder.ensure_models(inp[go.MODEL_TYPE], device, inp[go.BOOST], inp[go.TILING_MODE])
        for count in range(0, len(inputimages)):
            if inputimages[count].mode == 'I':                      # :135-137
                inputimages[count].point(lambda p: p * 0.0039063096, mode='RGB')
                inputimages[count] = inputimages[count].convert('RGB')

            image = inputimages[count]
            mesh_source = None      # what :281 calls depthi: the raw prediction, or the custom depth map
            if inputdepthmaps is not None and inputdepthmaps[count] is not None:
                out = _custom_depth_to_float(inputdepthmaps[count], image)                     # :145-174
                out_t = torch.from_numpy(np.array(out, dtype=np.float64, order='C')).to(device)
                img_output_t = _native.convert_to_i16(out_t)                                   # :211
                mesh_source = out_t
            else:
                if inp[go.NET_SIZE_MATCH]:                                                     # :177-184
                    net_width = (image.width + 31) // 32 * 32
                    net_height = (image.height + 31) // 32 * 32
                else:
                    net_width = inp[go.NET_WIDTH]
                    net_height = inp[go.NET_HEIGHT]
                raw_prediction, raw_prediction_invert = model_holder.get_raw_prediction(image, net_width, net_height)
                pred_t = raw_prediction if torch.is_tensor(raw_prediction) else torch.from_numpy(np.asarray(raw_prediction))
                pred_t = pred_t.to(device=device, dtype=torch.float32)
                mesh_source = pred_t
                if not inp[go.CLIPDEPTH] and not inp[go.DO_OUTPUT_DEPTH_PREDICTION]:
                    # fused device path: min/max -> normalise -> uint16 in one pass (:189-211)
                    img_output_t = _native.depth_to_u16(pred_t.unsqueeze(0), raw_prediction_invert)[0]
                else:
                    out_t, prediction_copy, broken = _postprocess_prediction(pred_t, raw_prediction_invert, inp)
                    if not broken:
                        if inp[go.DO_OUTPUT_DEPTH_PREDICTION]:
                            yield count, 'depth_prediction', prediction_copy.cpu().numpy().copy()
                        if out_t.dtype == torch.float64:                                        # 'Outliers': numpy promoted
                            out_t = (out_t - out_t.min()) / (out_t.max() - out_t.min())                 # :202
                            img_output_t = _native.convert_to_i16(out_t.contiguous())                   # :211
                        else:
                            img_output_t = _native.depth_to_u16(out_t.to(torch.float32).unsqueeze(0), False)[0]   # :202,:211
                    else:
                        img_output_t = torch.zeros(pred_t.shape, dtype=torch.uint16, device=device)          # :206

            img_output = None       # host copy of the uint16 depth, made only if a host consumer needs it

            if inp[go.DO_OUTPUT_DEPTH]:                                                        # :240-249
                img_output = img_output_t.cpu().numpy()
                img_depth = np.bitwise_not(img_output) if inp[go.OUTPUT_DEPTH_INVERT] else img_output   # cv2.bitwise_not
                if inp[go.OUTPUT_DEPTH_COMBINE]:
                    axis = 1 if inp[go.OUTPUT_DEPTH_COMBINE_AXIS] == 'Horizontal' else 0
                    img_concat = Image.fromarray(np.concatenate(
                        (image, convert_i16_to_rgb(img_depth, np.asarray(image))), axis=axis))
                    yield count, 'concat_depth', img_concat
                else:
                    yield count, 'depth', Image.fromarray(img_depth)

            if inp[go.GEN_STEREO]:                                                             # :251-259
                modes = inp[go.STEREO_MODES]
                img_t = torch.from_numpy(np.array(image, dtype=np.uint8, order='C')).to(device)
                if img_t.dim() != 3:
                    raise ValueError('not enough values to unpack (expected 3, got %d)' % img_t.dim())
                stereo = create_stereoimages_batch(
                    img_t.unsqueeze(0), img_output_t.unsqueeze(0),
                    inp[go.STEREO_DIVERGENCE], inp[go.STEREO_SEPARATION], modes,
                    inp[go.STEREO_BALANCE], inp[go.STEREO_OFFSET_EXPONENT], inp[go.STEREO_FILL_ALGO])
                for c in range(0, len(stereo)):
                    yield count, inp[go.STEREO_MODES][c], Image.fromarray(stereo[c][0].cpu().numpy())

            if inp[go.GEN_NORMALMAP]:                                                          # :261-269
                nm = create_normalmap_batch(
                    img_output_t.unsqueeze(0),
                    inp[go.NORMALMAP_PRE_BLUR_KERNEL] if inp[go.NORMALMAP_PRE_BLUR] else None,
                    inp[go.NORMALMAP_SOBEL_KERNEL] if inp[go.NORMALMAP_SOBEL] else None,
                    inp[go.NORMALMAP_POST_BLUR_KERNEL] if inp[go.NORMALMAP_POST_BLUR] else None,
                    inp[go.NORMALMAP_INVERT])
                yield count, 'normalmap', Image.fromarray(nm[0].cpu().numpy())

            if inp[go.GEN_HEATMAP]:                                                            # :271-274
                from .heatmap import colorize_batch
                yield count, 'heatmap', Image.fromarray(colorize_batch(img_output_t.unsqueeze(0))[0].cpu().numpy())

            if inp[go.GEN_SIMPLE_MESH]:                                                        # :277-306
                from . import mesh_generation as mg
                custom = inputdepthmaps[count] is not None
                mt = inp[go.MODEL_TYPE]          # callers pass the numeric id; the option's default is a display name
                depthi = mg.mesh_depth(mesh_source, mt if isinstance(mt, int) else -1, bool(inp[go.BOOST]), custom)
                rgb_t = torch.from_numpy(np.array(image.convert('RGB'), dtype=np.uint8, order='C')).to(device)
                verts, faces, colors = mg.create_mesh_arrays(rgb_t, depthi, keep_edges=not inp[go.SIMPLE_MESH_OCCLUDE],
                                                             spherical=bool(inp[go.SIMPLE_MESH_SPHERICAL]))
                fn = mg.unique_filename(outpath, 'depthmap', 'obj', 'simple')
                yield count, 'simple_mesh', mg.write_obj(fn, verts.cpu().numpy(), faces.cpu().numpy(), colors.cpu().numpy())
    except Exception as e:
        if 'out of memory' in str(e).lower():                                                  # :308-326
            suggestion = "out of GPU memory, could not generate depthmap! " \
                         "Here are some suggestions to work around this issue:\n"
            if inp[go.MODEL_TYPE] != 6:
                suggestion += " * Use a different model (generally, more memory-consuming models produce better depthmaps)\n"
            if not inp[go.BOOST]:
                suggestion += " * Reduce net size (this could reduce quality)\n"
            raise Exception(suggestion)
        raise e
    finally:
        if ops.get('keepmodels', True):
            model_holder.offload()
        else:
            model_holder.unload_models()
        gc.collect()


# the name BASELINE.json's north_star uses (a pre-0.4.0 name of the same function, see SURVEY.md naming note)
run_depthmap = core_generation_funnel


def unload_models():
    """reference: src/core.py:669."""
    model_holder.unload_models()
