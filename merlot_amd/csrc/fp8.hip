// fp8 (OCP e4m3fn) operand preparation for BASELINE config #5's GEMM path: per-tensor scaling, "current" (the scale comes
// from the tensor being quantised, not from a history).  The reference has no fp8 path at all (its precision policy is
// bf16 compute / fp32 params, utils/model_utils.py:572-602); the contract is SURVEY.md 7(vii): per-tensor scales, fp32
// accumulation, loss within 2e-2 of the bf16 path.
//
//   scale[0] = s = 448 / amax(|x|)      (1 if the tensor is all zeros)
//   scale[1] = 1 / s                    -- what merlot_gemm_fp8_nt multiplies its accumulators by
//   scale[2] = amax (bit pattern of a non-negative float, max-ed with integer atomics)
//   y = e4m3( clamp(x * s, -448, 448) ) round-to-nearest-even
//
// Two passes over x (HBM-bound, 2 + 2 + 1 bytes per element): amax, then convert.  x is read as 16-B vectors, y written
// as 8-B vectors; rows may be strided (ldx, ldy in elements).
#include "common.h"

namespace {

constexpr float E4M3_MAX = 448.f;

__global__ __launch_bounds__(256) void amax_bf16_kernel(const bf16* __restrict__ x, int64_t rows, int cols8, int64_t ldx,
                                                        unsigned int* __restrict__ amax_bits) {
    const int64_t total = rows * cols8;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols8;
        const int c = (int)(i - r * cols8);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ldx + (int64_t)c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax(amax_bits, __float_as_uint(m));        // non-negative floats order like their bit patterns
    }
}

__device__ __forceinline__ float scale_from_amax(float amax) { return amax > 0.f ? E4M3_MAX / amax : 1.f; }

__global__ __launch_bounds__(256) void quantize_e4m3_kernel(const bf16* __restrict__ x, int64_t rows, int cols8, int64_t ldx,
                                                            uint8_t* __restrict__ y, int64_t ldy, float* __restrict__ scale) {
    const float s = scale_from_amax(scale[2]);
    const int64_t total = rows * cols8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols8;
        const int c = (int)(i - r * cols8);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ldx + (int64_t)c * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = clamp_e4m3((float)v[e] * s);
        u32x2 o;
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
        o[0] = (uint32_t)w;
        w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w, true);
        o[1] = (uint32_t)w;
        *reinterpret_cast<u32x2*>(y + r * ldy + (int64_t)c * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {           // every reader derives s from scale[2] itself: no ordering needed
        scale[0] = s;
        scale[1] = 1.f / s;
    }
}

}  // namespace

extern "C" int merlot_quantize_e4m3(const void* x, int64_t rows, int64_t cols, int64_t ldx, void* y, int64_t ldy,
                                    float* scale, merlot_stream_t stream) {
    MERLOT_CHECK(x && y && scale, MERLOT_ESHAPE, "merlot_quantize_e4m3: null argument");
    MERLOT_CHECK(rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= cols && ldy >= cols, MERLOT_ESHAPE,
                 "merlot_quantize_e4m3: rows=%lld cols=%lld ldx=%lld ldy=%lld (cols, ldx, ldy multiples of 8)", (long long)rows,
                 (long long)cols, (long long)ldx, (long long)ldy);
    MERLOT_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, MERLOT_EALIGN, "merlot_quantize_e4m3: x 16-byte, y 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scale, 0, 3 * sizeof(float), s);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "merlot_quantize_e4m3: memset failed: %s", hipGetErrorString(e));
    const int cols8 = (int)(cols / 8);
    const int64_t total = rows * cols8;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(amax_bf16_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, cols8, ldx,
                       reinterpret_cast<unsigned int*>(scale + 2));
    hipLaunchKernelGGL(quantize_e4m3_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, cols8, ldx, (uint8_t*)y, ldy, scale);
    return merlot_launch_status("merlot_quantize_e4m3");
}

// ---- ABI v9: operand preparation for the 8-bit weight-gradient kernel (merlot_gemm_f8_tn, gemm_q8.inc).  e4m3 or e5m2 (fmt 0 / 1), zero rows appended up
// to rows_pad (the kernel's K-tile is 128 reduction rows; zeros add nothing), and two ways of getting the per-tensor scale:
//   current (delayed = 0): as merlot_quantize_e4m3 -- an amax pass, then the convert pass (2 + 2 + 1 bytes per element);
//   delayed (delayed = 1): ONE pass (2 + 1 bytes per element) with the scale from the amax this block recorded in the PREVIOUS call (scale[3]), while the
//     same pass records this tensor's amax for the next one; values beyond the old range saturate.  The first call of a block has to be a current one.
// scale = device float[4]: {s, 1/s, amax the scale was made from, amax of the tensor just quantised}.
namespace {

constexpr float E5M2_MAX = 57344.f;

template <int FMT>
__device__ __forceinline__ uint32_t cvt4_f8(float a, float b, float c, float d) {
    int w = 0;
    if (FMT == 0) {
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    } else {
        w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
    }
    return (uint32_t)w;
}

__global__ void f8_scale_rotate_kernel(float* scale, float fmax, int current) {   // delayed: the recorded amax (if any: merlot_f8_scale_rotate may have consumed
    const float amax = scale[3];                                                   // it already) becomes the scale's; current: the amax pass has just run
    if (amax > 0.f || current) {
        const float s = amax > 0.f ? fmax / amax : 1.f;
        scale[0] = s;
        scale[1] = 1.f / s;
        scale[2] = amax;
    }
}

template <int FMT, bool RECORD>
__global__ __launch_bounds__(256) void quantize_f8_kernel(const bf16* __restrict__ x, int64_t rows, int64_t rows_pad, int cols8, int64_t ldx,
                                                          uint8_t* __restrict__ y, int64_t ldy, float* __restrict__ scale) {
    constexpr float FMAX = FMT == 0 ? E4M3_MAX : E5M2_MAX;
    const float s = scale[0];
    const int64_t total = rows * cols8, total_pad = rows_pad * cols8;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pad; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols8;
        const int c = (int)(i - r * cols8);
        u32x2 o = {0u, 0u};
        if (i < total) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ldx + (int64_t)c * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (float)v[e];
                if (RECORD) m = fmaxf(m, fabsf(t));
                f[e] = fminf(fmaxf(t * s, -FMAX), FMAX);
            }
            o[0] = cvt4_f8<FMT>(f[0], f[1], f[2], f[3]);
            o[1] = cvt4_f8<FMT>(f[4], f[5], f[6], f[7]);
        }
        *reinterpret_cast<u32x2*>(y + r * ldy + (int64_t)c * 8) = o;
    }
    if (RECORD) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
            atomicMax(reinterpret_cast<unsigned int*>(scale + 3), __float_as_uint(m));
        }
    }
}

}  // namespace

// A producer that writes its own 8-bit copy (merlot_gemm_*_nt_q8, merlot_ln_*_q8t) scales by block[0] and max-es its amax into block[3]: this call, on the
// stream IN FRONT of the producer, turns the amax the previous step recorded into the step's scale and clears the record -- the delayed-scaling step of
// merlot_quantize_f8 without the pass.  n blocks of 4 floats, consecutive.
namespace {
__global__ void f8_scale_rotate_n_kernel(float* blocks, int n, float fmax_e4m3, float fmax_e5m2, const int* fmts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* b = blocks + 4 * i;
    const float amax = b[3];
    if (amax > 0.f) {                                    // (a block whose producer has not run since the last rotation keeps its scale)
        const float fm = fmts[i] == 0 ? fmax_e4m3 : fmax_e5m2;
        b[0] = fm / amax;
        b[1] = amax / fm;
        b[2] = amax;
        b[3] = 0.f;
    }
}
}  // namespace
extern "C" int merlot_f8_scale_rotate(float* blocks, int n, const int32_t* fmts, merlot_stream_t stream) {
    MERLOT_CHECK(blocks && fmts && n > 0, MERLOT_ESHAPE, "merlot_f8_scale_rotate: null argument");
    hipLaunchKernelGGL(f8_scale_rotate_n_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, blocks, n, E4M3_MAX, E5M2_MAX, (const int*)fmts);
    return merlot_launch_status("merlot_f8_scale_rotate");
}

extern "C" int merlot_quantize_f8(const void* x, int64_t rows, int64_t cols, int64_t ldx, void* y, int64_t ldy, int64_t rows_pad, int fmt,
                                  int delayed, float* scale, merlot_stream_t stream) {
    MERLOT_CHECK(x && y && scale, MERLOT_ESHAPE, "merlot_quantize_f8: null argument");
    MERLOT_CHECK(fmt == 0 || fmt == 1, MERLOT_ESHAPE, "merlot_quantize_f8: fmt is 0 (e4m3) or 1 (e5m2)");
    MERLOT_CHECK(rows > 0 && rows_pad >= rows && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= cols && ldy >= cols, MERLOT_ESHAPE,
                 "merlot_quantize_f8: rows=%lld rows_pad=%lld cols=%lld ldx=%lld ldy=%lld (cols, ldx, ldy multiples of 8)", (long long)rows,
                 (long long)rows_pad, (long long)cols, (long long)ldx, (long long)ldy);
    MERLOT_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, MERLOT_EALIGN, "merlot_quantize_f8: x 16-byte, y 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float fmax = fmt == 0 ? E4M3_MAX : E5M2_MAX;
    const int cols8 = (int)(cols / 8);
    const int64_t total = rows_pad * cols8;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (!delayed) {
        hipError_t e = hipMemsetAsync(scale, 0, 4 * sizeof(float), s);
        MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "merlot_quantize_f8: memset failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(amax_bf16_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, cols8, ldx, reinterpret_cast<unsigned int*>(scale + 3));
        hipLaunchKernelGGL(f8_scale_rotate_kernel, dim3(1), dim3(1), 0, s, scale, fmax, 1);
        if (fmt == 0) hipLaunchKernelGGL((quantize_f8_kernel<0, false>), dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, rows_pad, cols8, ldx, (uint8_t*)y, ldy, scale);
        else hipLaunchKernelGGL((quantize_f8_kernel<1, false>), dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, rows_pad, cols8, ldx, (uint8_t*)y, ldy, scale);
    } else {
        hipLaunchKernelGGL(f8_scale_rotate_kernel, dim3(1), dim3(1), 0, s, scale, fmax, 0);
        hipError_t e = hipMemsetAsync(scale + 3, 0, sizeof(float), s);
        MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "merlot_quantize_f8: memset failed: %s", hipGetErrorString(e));
        if (fmt == 0) hipLaunchKernelGGL((quantize_f8_kernel<0, true>), dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, rows_pad, cols8, ldx, (uint8_t*)y, ldy, scale);
        else hipLaunchKernelGGL((quantize_f8_kernel<1, true>), dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, rows_pad, cols8, ldx, (uint8_t*)y, ldy, scale);
    }
    return merlot_launch_status("merlot_quantize_f8");
}

namespace {
// per column GROUP maxima in one pass: cols = groups * gcols8 * 8, amax_bits[g] = max|x[:, g-th group]|
__global__ __launch_bounds__(256) void amax_groups_bf16_kernel(const bf16* __restrict__ x, int64_t rows, int gcols8, int groups,
                                                               int64_t ldx, unsigned int* __restrict__ amax_bits) {
    const int cols8 = gcols8 * groups;
    const int64_t total = rows * cols8;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols8;
        const int c = (int)(i - r * cols8);
        const int g = c / gcols8;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ldx + (int64_t)c * 8);
        float mm = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) mm = fmaxf(mm, fabsf((float)v[e]));
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = (k == g) ? fmaxf(m[k], mm) : m[k];
    }
    __shared__ float part[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[k] = fmaxf(m[k], __shfl_xor(m[k], o, 64));
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        const int k = threadIdx.x;
        atomicMax(amax_bits + k, __float_as_uint(fmaxf(fmaxf(part[0][k], part[1][k]), fmaxf(part[2][k], part[3][k]))));
    }
}
}  // namespace

// amax[g] = max|x[:, g * cols/groups : (g + 1) * cols/groups]| over a [rows, cols] bf16 view (row stride ldx), g < groups <= 4;
// amax (device float[groups]) is zeroed here, on the stream.
extern "C" int merlot_amax_bf16(const void* x, int64_t rows, int64_t cols, int64_t ldx, int groups, float* amax, merlot_stream_t stream) {
    MERLOT_CHECK(x && amax, MERLOT_ESHAPE, "merlot_amax_bf16: null argument");
    MERLOT_CHECK(rows > 0 && cols > 0 && groups >= 1 && groups <= 4 && cols % (8 * groups) == 0 && ldx % 8 == 0 && ldx >= cols, MERLOT_ESHAPE,
                 "merlot_amax_bf16: rows=%lld cols=%lld ldx=%lld groups=%d (cols a multiple of 8 * groups, ldx of 8, groups <= 4)",
                 (long long)rows, (long long)cols, (long long)ldx, groups);
    MERLOT_CHECK(((uintptr_t)x & 15) == 0, MERLOT_EALIGN, "merlot_amax_bf16: x must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(amax, 0, groups * sizeof(float), s);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "merlot_amax_bf16: memset failed: %s", hipGetErrorString(e));
    const int gcols8 = (int)(cols / 8 / groups);
    const int64_t total = rows * (cols / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(amax_groups_bf16_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)x, rows, gcols8, groups, ldx,
                       reinterpret_cast<unsigned int*>(amax));
    return merlot_launch_status("merlot_amax_bf16");
}
