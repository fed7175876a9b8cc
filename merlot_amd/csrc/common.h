// Shared device/host helpers for the MI355X (gfx950 / CDNA4) MERLOT kernels.
// wave = 64 lanes, 4 SIMDs per CU, 160 KiB LDS per CU.  gfx950 only -- no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/merlot_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error plumbing (host) -----------------------------------------------------------------
void merlot_set_error(const char* fmt, ...);
#define MERLOT_CHECK(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            merlot_set_error(__VA_ARGS__);        \
            return (code);                        \
        }                                         \
    } while (0)

static inline int merlot_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        merlot_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MERLOT_ELAUNCH;
    }
    return MERLOT_OK;
}

// ---- device math ---------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) {
    // exact erf form, utils/model_utils.py:96-110
    return x * (0.5f * (1.0f + erff(x * 0.70710678118654752440f)));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// counter-based dropout keep decision: one 32-bit hash per element (site seed + linear index).
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    // thresh = p * 2^32 ; keep iff hash >= thresh
    uint32_t h = hash32((uint32_t)idx ^ (uint32_t)seed) ^ hash32((uint32_t)(idx >> 32) + (uint32_t)(seed >> 32) + 0x9e3779b9U);
    h = hash32(h);
    return h >= thresh;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
