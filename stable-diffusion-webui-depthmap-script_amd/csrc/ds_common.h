// Internal helpers shared by the HIP translation units of libdepthstereo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/depthstereo.h"

#define DS_API extern "C" __attribute__((visibility("default")))

// ---- error plumbing -------------------------------------------------------------------------
void ds_set_error(const char *fmt, ...);

#define DS_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ds_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DS_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define DS_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            ds_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

// ---- context ----------------------------------------------------------------------------------
#define DS_KT_KINDS 24          // DS_KT_* of include/depthstereo.h (0..11) and + 12 for the ragged round of a GEMM kind
#define DS_KT_RING 1024
struct ds_ctx {
    int device;
    // growable scratch blocks (device memory)
    void *minmax;      size_t minmax_bytes;      // n * 2 doubles
    void *partials;    size_t partials_bytes;    // block partials of the min/max reduction
    void *row_flags;   size_t row_flags_bytes;   // one int per (image, eye, row) + counters
    void *row_list;    size_t row_list_bytes;    // compacted list of flagged rows
    void *exact_ws;    size_t exact_ws_bytes;    // scratch of the exact sequential sweep
    void *tmp_a;       size_t tmp_a_bytes;       // generic temporaries (normal-map blur planes ...)
    void *tmp_b;       size_t tmp_b_bytes;
    void *zero_line;   size_t zero_line_bytes;   // 256 zero bytes: the padding ring of ds_conv3x3_nhwc
    int zero_line_cleared;
    void *lin_ws;      size_t lin_ws_bytes;      // ds_linear's ragged round with a K split: arrival counters + fp32 partials
    void *lin_ws_cleared;                        // the block whose counters have been zeroed
    void *gn_ws;       size_t gn_ws_bytes;       // ds_group_norm_nchw: float32 moments per (image, group, slice); ONE size (graphs hold it)
    int ncu;                                     // CU count of `device` rounded down to a multiple of 8 (0 = not read yet)
    int64_t last_exact_rows_valid;
    // optional kernel timing (ds_profile_enable)
    int profile;
    hipEvent_t ev[4];          // render start/stop, exact start/stop
    int ev_created, ev_recorded;
    // in-step kernel timers (ds_kernel_timer_enable): per kind a ring of event pairs recorded around the launches on the caller's stream
    int ktimer;
    hipEvent_t (*kt_ev[DS_KT_KINDS])[2];
    int kt_n[DS_KT_KINDS];
};

// ds_kt_begin: -1 when the timer is off, the ring of this kind is full, or the stream is being captured; else the slot whose start
// event was recorded.  ds_kt_end records the slot's stop event.  Two calls around ONE kernel launch on the same stream.
int ds_kt_begin(ds_ctx *ctx, int kind, hipStream_t st);
void ds_kt_end(ds_ctx *ctx, int kind, int slot, hipStream_t st);

int ds_ctx_reserve(ds_ctx *ctx, void **slot, size_t *cur, size_t need);

// ---- device helpers -----------------------------------------------------------------------------
// float64 -> uint8 exactly as numpy/numba do on x86-64: truncate toward zero, keep the low byte.
__device__ __forceinline__ uint8_t ds_f64_to_u8(double v)
{
    if (!(v == v)) return 0;
    if (v >= 9.2e18 || v <= -9.2e18) return 0;
    return (uint8_t)(long long)v;
}

__device__ __forceinline__ double ds_wave_min(double v)
{
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ double ds_wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

// Depth normalisation of stereoimage_generation.py:79-81 for one element.
//   U16: (depth - min) stays uint16, '/' promotes both operands to float64
//   F32: subtraction and division in float32, the kernel then reads the value as float64
//   F64: float64 throughout
template <int DT> struct ds_depth_traits;
template <> struct ds_depth_traits<DS_DEPTH_U16> {
    typedef uint16_t T;
    __device__ static __forceinline__ double norm(T v, double mn, double mx) {
        return (double)(uint16_t)(v - (uint16_t)mn) / (double)(uint16_t)((uint16_t)mx - (uint16_t)mn);
    }
};
template <> struct ds_depth_traits<DS_DEPTH_F32> {
    typedef float T;
    __device__ static __forceinline__ double norm(T v, double mn, double mx) {
        float den = (float)mx - (float)mn;
        float q = (v - (float)mn) / den;
        return (double)q;
    }
};
template <> struct ds_depth_traits<DS_DEPTH_F64> {
    typedef double T;
    __device__ static __forceinline__ double norm(T v, double mn, double mx) {
        return (v - mn) / (mx - mn);
    }
};
