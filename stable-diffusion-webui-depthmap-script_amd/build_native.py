"""Build libdepthstereo_hip.so (gfx950) in-tree with hipcc.

    python build_native.py [--force]
    DS_EXPERIMENTS=1 python build_native.py [--force]      -> libdepthstereo_hip_experiments.so (load it with DS_NATIVE_LIB=<path>)

Flags that matter for bit-exactness: -ffp-contract=off (no FMA contraction: every float64 operation
rounds like numpy/numba), no -ffast-math.  f32/f64 division and sqrt are the correctly rounded
expansions (hipcc default).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
EXPERIMENTS = bool(os.environ.get("DS_EXPERIMENTS"))
# the experiments build is a SECOND library beside the product one (src/_native.py loads it only when DS_NATIVE_LIB names it)
OUT = os.path.join(HERE, "libdepthstereo_hip_experiments.so" if EXPERIMENTS else "libdepthstereo_hip.so")
OBJ_DIR = os.path.join(HERE, "build", "experiments") if EXPERIMENTS else os.path.join(HERE, "build")
SOURCES = ["ds_api.hip", "ds_stereo.hip", "ds_stereo_polylines.hip", "ds_normalmap.hip", "ds_attention.hip", "ds_attention4.hip", "ds_encoder_ops.hip", "ds_boost.hip", "ds_heatmap.hip", "ds_linear.hip", "ds_gconv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         # MFMA results land in ordinary VGPRs (gfx950 has one unified file): no v_accvgpr_read/write around the softmax
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


if EXPERIMENTS:
    # timing ablations, superseded kernel generations and kernels that have not run on hardware yet (DS_ATT_ABLATE, DS_ATT_V1,
    # DS_ATT_GEN=3, DS_LIN_ABLATE, DS_PL_DEBUG ...): some of them produce WRONG results by design, so they are compiled only on
    # request, into a library of their own, never into the default one
    FLAGS.append("-DDS_EXPERIMENTS")


# per-file flags.  ds_attention4.hip: its phases are interleaved by hand (one MFMA, one fragment read, a share of the softmax per
# step); the SLP vectoriser would pack the row-sum adds into v_pk_add_f32, which costs more than two plain adds beside MFMAs
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs").  Same values either way.
EXTRA_FLAGS = {"ds_attention4.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "depthstereo.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    objs = []
    procs = []
    os.makedirs(OBJ_DIR, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
