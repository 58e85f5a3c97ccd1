// Shared helpers for libnerfhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nerf_hip.h"

#define NERF_API extern "C" __attribute__((visibility("default")))

namespace nerf {

void set_error(const char* fmt, ...);

// Records the message and returns the code (used as `return fail(NERF_E_ARG, "...")`).
#define NERF_FAIL(code, ...) (nerf::set_error(__VA_ARGS__), (code))

#define NERF_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return NERF_FAIL(NERF_E_ARG, __VA_ARGS__);   \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline hipStream_t as_stream(nerf_stream_t s) { return (hipStream_t)s; }

// Compute units of the current device (cached per device; 0 when there is no device to ask).
inline int device_cu_count() {
    static thread_local int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// Makes the runtime load, on the current device, the code object that holds `kernel` (HIP loads a
// translation unit's device code when something in it is first used, and takes device memory for
// it then): set-up paths of entry points that promise to allocate nothing call it for every unit
// whose kernels they will launch.
template <class K>
inline int preload_code_of(K kernel, const char* what) {
    hipFuncAttributes attr;
    const hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel));
    if (e != hipSuccess) return NERF_FAIL((int)e, "%s: loading the device code: %s", what, hipGetErrorString(e));
    return 0;
}
// One per translation unit whose kernels nerf_render_rays launches.
int preload_rays_code(const char* what);
int preload_mlp_forward_code(const char* what);
int preload_composite_code(const char* what);
int preload_sample_pdf_code(const char* what);

constexpr int kWave = 64;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                           CTRL, ROW_MASK, 0xf, false));
}

// Inclusive product scan over the wave's 64 lanes (lane i gets v0*...*vi).
__device__ __forceinline__ float wave_scan_mul(float v) {
    v *= dpp_or<0x111, 0xf>(1.0f, v);  // row_shr:1
    v *= dpp_or<0x112, 0xf>(1.0f, v);  // row_shr:2
    v *= dpp_or<0x114, 0xf>(1.0f, v);  // row_shr:4
    v *= dpp_or<0x118, 0xf>(1.0f, v);  // row_shr:8
    v *= dpp_or<0x142, 0xa>(1.0f, v);  // row_bcast:15 -> rows 1 and 3
    v *= dpp_or<0x143, 0xc>(1.0f, v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// Inclusive sum scan, same shape.
__device__ __forceinline__ float wave_scan_add(float v) {
    v += dpp_or<0x111, 0xf>(0.0f, v);
    v += dpp_or<0x112, 0xf>(0.0f, v);
    v += dpp_or<0x114, 0xf>(0.0f, v);
    v += dpp_or<0x118, 0xf>(0.0f, v);
    v += dpp_or<0x142, 0xa>(0.0f, v);
    v += dpp_or<0x143, 0xc>(0.0f, v);
    return v;
}

__device__ __forceinline__ float lane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum across the 64 lanes of a wave; every lane gets the result (six DPP adds and one lane
// read: no LDS traffic, unlike a butterfly of ds_bpermute).
__device__ __forceinline__ float wave_sum(float v) { return lane63(wave_scan_add(v)); }

// Lane i gets lane i-1's value; lane 0 gets `first`.
__device__ __forceinline__ float wave_shr1(float first, float v) {
    return dpp_or<0x138, 0xf>(first, v);  // wave_shr:1
}

// Lane i gets lane i+1's value; lane 63 gets `last`.
__device__ __forceinline__ float wave_shl1(float last, float v) {
    return dpp_or<0x130, 0xf>(last, v);  // wave_shl:1
}

__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Inclusive suffix sum over the wave's 64 lanes (lane i gets v_i + ... + v_63), summing only the
// later lanes (no total-minus-prefix cancellation).
__device__ __forceinline__ float wave_rscan_add(float v) {
    v += dpp_or<0x101, 0xf>(0.0f, v);  // row_shl:1
    v += dpp_or<0x102, 0xf>(0.0f, v);  // row_shl:2
    v += dpp_or<0x104, 0xf>(0.0f, v);  // row_shl:4
    v += dpp_or<0x108, 0xf>(0.0f, v);  // row_shl:8
    const float t1 = read_lane(v, 16), t2 = read_lane(v, 32), t3 = read_lane(v, 48);
    const int row = (threadIdx.x & 63) >> 4;
    const float add = (row == 0) ? (t1 + t2) + t3 : (row == 1) ? t2 + t3 : (row == 2) ? t3 : 0.0f;
    return v + add;
}

}  // namespace nerf
