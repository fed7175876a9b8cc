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
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// Experiment switches (scripts/ only).  The product library is compiled WITHOUT MERLOT_EXPERIMENTS: no getenv, no
// debug bits in any kernel, no probe exports -- `build.sh exp` builds libmerlot_hip_exp.so with them.
#ifdef MERLOT_EXPERIMENTS
#define DBG_BIT(p, bit) ((((p).dbg) & (bit)) != 0)
#else
#define DBG_BIT(p, bit) (false)
#endif

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((address_space(3))) int lds_int_t;
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

// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a kernel ON ONE DEVICE: a process that drives several GPUs has
// to set it on each of them.  `LdsAttrOnce` remembers, per device, the largest size already granted to one kernel (a cache of an
// idempotent driver attribute, not library state: losing it only costs one more hipFuncSetAttribute); host threads may race.
struct LdsAttrOnce {
    int granted[64];                                     // zero-initialised (static storage); index = device ordinal & 63
};
static inline int merlot_ensure_lds(LdsAttrOnce& st, const void* kern, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int* slot = &st.granted[dev & 63];
    if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) >= bytes) return MERLOT_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        merlot_set_error("%s: hipFuncSetAttribute(LDS=%d) failed on device %d: %s", what, bytes, dev, hipGetErrorString(e));
        return MERLOT_ELAUNCH;
    }
    int cur = __atomic_load_n(slot, __ATOMIC_RELAXED);   // published only AFTER the attribute is in place
    while (cur < bytes && !__atomic_compare_exchange_n(slot, &cur, bytes, true, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) {
    }
    return MERLOT_OK;
}
#define MERLOT_ENSURE_LDS(kern, bytes, what)                                                         \
    do {                                                                                              \
        static LdsAttrOnce lds_once_;                                                                 \
        const int rc_ = merlot_ensure_lds(lds_once_, reinterpret_cast<const void*>(kern), bytes, what); \
        if (rc_ != MERLOT_OK) return rc_;                                                             \
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

// GEMM-epilogue forms of GELU and GELU' (bf16 outputs).  A K=768 GEMM spends 96 matrix-pipe cycles per output element
// and lane; erff() + the rest cost ~100 VALU cycles per element on top, and even an Abramowitz-Stegun erf still pays
// 2 quarter-rate transcendentals (profiles/r01_f_epilogue_decomposition.txt).  Here, with a = min(|x|, 4.5),
//   gelu(x)  = max(x, 0) - r(a),            r(a) = a * Phi(-a)            (smooth hump, -> 0)
//   gelu'(x) = x >= 0 ? 1 - d(a) : d(a),    d(a) = Phi(-a) - a * phi(a)   (smooth, -> 0)
// and r, d are degree-10 polynomials in t = a * 2/4.5 - 1 (Chebyshev fit, fp32 Horner): max abs error 1.6e-5 / 1.1e-4 over the
// whole line -- what the clamp at 4.5 leaves anyway (r(4.5) = 1.5e-5, d(4.5) = -6.9e-5: the degree-12 fits of rounds 1-3 were
// 2.5e-6 / 4.2e-6 inside the interval and no better than that outside), below half a bf16 ulp of every gelu output with
// |y| >= 0.008 -- no transcendental, evaluated two elements per instruction with v_pk_fma_f32.  The GELU epilogues are VALU-bound
// (profiles/r02_f_epilogue_timeline.txt: the polynomial was ~10 k of the fc1 tile's 23 k epilogue cycles), so every FMA counts.
// Coefficients: scripts/fit_gelu_poly.py.
__device__ constexpr float GELU_R[11] = {2.749903500e-02f, -1.330170631e-01f, 2.464785576e-01f, -1.475407928e-01f, -2.019351274e-01f, 4.360252321e-01f, -2.080824375e-01f, -1.780532300e-01f, 1.802777648e-01f, 2.258405089e-02f, -4.423601553e-02f};
__device__ constexpr float GELU_D[11] = {-5.916317180e-02f, 2.194041312e-01f, -1.940523237e-01f, -3.639891744e-01f, 9.440367818e-01f, -5.339142084e-01f, -4.742373228e-01f, 6.090298295e-01f, -1.214299630e-02f, -1.806256324e-01f, 4.554631189e-02f};
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int GELU_DEG = 10;
__device__ __forceinline__ f32x2 gelu_poly_pk(f32x2 a, const float (&c)[GELU_DEG + 1]) {
    const f32x2 lim = {4.5f, 4.5f}, sc = {2.0f / 4.5f, 2.0f / 4.5f}, m1 = {-1.0f, -1.0f};
    a = __builtin_elementwise_min(a, lim);
    const f32x2 t = __builtin_elementwise_fma(a, sc, m1);
    f32x2 acc = {c[GELU_DEG], c[GELU_DEG]};
#pragma unroll
    for (int k = GELU_DEG - 1; k >= 0; --k) {
        const f32x2 ck = {c[k], c[k]};
        acc = __builtin_elementwise_fma(acc, t, ck);
    }
    return acc;
}
// in-place on 2 elements
__device__ __forceinline__ void gelu_fast2(float& x0, float& x1) {
    const f32x2 a = {__builtin_fabsf(x0), __builtin_fabsf(x1)};
    const f32x2 r = gelu_poly_pk(a, GELU_R);
    x0 = __builtin_fmaxf(x0, 0.f) - r[0];
    x1 = __builtin_fmaxf(x1, 0.f) - r[1];
}
__device__ __forceinline__ void gelu_grad_fast2(float u0, float u1, float& g0, float& g1) {
    const f32x2 a = {__builtin_fabsf(u0), __builtin_fabsf(u1)};
    const f32x2 d = gelu_poly_pk(a, GELU_D);
    g0 = u0 >= 0.f ? 1.0f - d[0] : d[0];
    g1 = u1 >= 0.f ? 1.0f - d[1] : d[1];
}
// (Round 4 measured the alternative the ISA suggested -- hipcc separates the dependent v_pk_fma_f32 of a chain by s_nop and
// evaluates pair after pair --: eight independent scalar chains of v_fmaak_f32 interleaved step by step.  Same-box A/B: fc1 622 ->
// 632 us, GELU' level, step +0.5 %: the packed form stays.  The epilogue is bound by VALU throughput (2 cycles per FMA and element
// either way), not by the chain's latency; profiles/r04_b_gelu_scalar_ab.txt.)
__device__ __forceinline__ float gelu_fast(float x) {
    float y0 = x, y1 = x;
    gelu_fast2(y0, y1);
    return y0;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float g0, g1;
    gelu_grad_fast2(x, x, g0, g1);
    return g0;
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

// saturating clamp into e4m3's finite range that PROPAGATES NaN (fminf / fmaxf return the other operand for a NaN input and
// would turn a diverged activation into +-448 silently; v_cvt_pk_fp8_f32 maps NaN to e4m3fn's NaN encoding)
__device__ __forceinline__ float clamp_e4m3(float x) {
    const float c = __builtin_fminf(__builtin_fmaxf(x, -448.f), 448.f);
    return x != x ? x : c;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
