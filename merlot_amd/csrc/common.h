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

// GEMM-epilogue forms (bf16 outputs): Abramowitz-Stegun 7.1.26 for erf (|error| <= 1.5e-7 absolute, i.e. far below
// the bf16 rounding of the result) on raw v_rcp / v_exp -- ~14 VALU instructions instead of ocml erff's two-branch
// evaluation; the epilogue arithmetic of a K=768 GEMM was costing a third of its main loop
// (profiles/r01_f_epilogue_decomposition.txt).  gelu'(u) re-uses the same exponential: exp(-z^2), z = |u|/sqrt(2).
__device__ __forceinline__ float erf_abs_fast(float z, float& e) {          // z >= 0; returns erf(z), e = exp(-z*z)
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return __builtin_fmaf(-poly * t, e, 1.0f);
}
__device__ __forceinline__ float gelu_fast(float x) {
    float e;
    const float er = erf_abs_fast(__builtin_fabsf(x) * 0.70710678118654752440f, e);
    return x * __builtin_fmaf(0.5f, __builtin_copysignf(er, x), 0.5f);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float e;
    const float er = erf_abs_fast(__builtin_fabsf(x) * 0.70710678118654752440f, e);
    const float cdf = __builtin_fmaf(0.5f, __builtin_copysignf(er, x), 0.5f);
    return __builtin_fmaf(x * 0.39894228040143267794f, e, cdf);
}

// counter-based dropout keep decision (site seed + linear element index): ONE 32-bit hash per PAIR of elements, its
// two 16-bit halves compared with a 16-bit threshold (p is resolved to 2^-16).  Every user -- GEMM residual epilogue,
// merlot_dropout_apply, the fused mask in ln_bwd -- goes through these functions, so forward and backward agree.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t dropout_word(uint64_t seed, uint64_t pair) {
    return hash32((uint32_t)pair ^ (uint32_t)seed) ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    // thresh = p * 2^32 ; keep iff this element's 16-bit field >= p * 2^16
    const uint32_t w = dropout_word(seed, idx >> 1);
    return ((idx & 1) ? (w >> 16) : (w & 0xffffu)) >= (thresh >> 16);
}
// N consecutive elements starting at an EVEN index: N/2 hashes
template <int N>
__device__ __forceinline__ void dropout_keep_n(uint64_t seed, uint64_t idx0, uint32_t thresh, bool (&keep)[N]) {
    static_assert(N % 2 == 0, "pairs");
    const uint32_t t16 = thresh >> 16;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        const uint32_t w = dropout_word(seed, (idx0 >> 1) + j);
        keep[2 * j] = (w & 0xffffu) >= t16;
        keep[2 * j + 1] = (w >> 16) >= t16;
    }
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
