// LayerNorm forward / backward for gfx950 (utils/model_utils.py:113-130): fp32 statistics, population
// variance, eps inside the rsqrt, y = x*s - mean*s + beta with s = rstd*gamma.
// HBM-bound: one wave per row, each lane owns 4 contiguous features per 256-wide slab (8-byte bf16 /
// 16-byte f32 accesses, 512 B / 1 KiB coalesced per wave instruction); row statistics by wave shuffles.
// (Round 5 measured the 16-byte variant for bf16 rows of 768 features -- every lane one 16-B chunk, the lower half-wave a second one: two
// instructions per row and operand instead of three --: forward level (252 vs 254 us at 405 504 rows, 4.9 TB/s), backward SLOWER (512 vs
// 450 - 467 us: half the wave idles in every second instruction and the per-lane partials double); not kept, scripts/exp_ln_wide.py's
// numbers are in profiles/r05_q_ln_wide.txt.)
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<bf16>(const bf16* p, float (&v)[4]) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
}
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
}
__device__ __forceinline__ void store4(bf16* p, const float (&v)[4]) {
    bf16x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (bf16)v[e];
    *reinterpret_cast<bf16x4*>(p) = t;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = v[e];
    *reinterpret_cast<f32x4*>(p) = t;
}

// 4 elements as they lie in memory (bf16: 8 bytes, f32: 16): the NEXT row's operands wait in this form while the current row is worked on (round 6)
template <typename T> struct Raw4;
template <> struct Raw4<bf16> { typedef bf16x4 type; };
template <> struct Raw4<float> { typedef f32x4 type; };
template <typename T>
__device__ __forceinline__ typename Raw4<T>::type load_raw(const T* p) { return *reinterpret_cast<const typename Raw4<T>::type*>(p); }
template <typename R>
__device__ __forceinline__ void unpack4(const R& t, float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
}

// Round 6: both LayerNorm kernels walk the rows with a grid-stride loop of at most 2 048 workgroups (the guide's sizing for memory-bound kernels) and request
// row r + stride's operands BEFORE they work on row r.  Until round 5 the forward launched one short-lived wave per row (101 376 workgroups at the bench's ViT
// row count) and the backward asked for the residual gradient only after its two wave reductions: 4.9 / 5.4 TB/s, bound by the chain load -> reduce -> load -> store
// of one row per wave, not by HBM.  Same-box A/B against the old build (profiles/r06_h_ln_prefetch.txt): forward 266 - 276 -> 256 us at 405 504 rows (4.6 -> 4.87
// TB/s), backward 561 - 587 -> 541 - 588 us there and 116 -> 102 us at 65 536 rows.  The arithmetic is unchanged; the compiler contracts it into FMAs slightly
// differently in the new loop (checksums agree to 6 digits, not bit for bit).
template <int NS, typename TX>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y16,
                                                     float* __restrict__ y32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int64_t rows, float eps,
                                                     uint8_t* __restrict__ y8 = nullptr, float* __restrict__ row_scale = nullptr,
                                                     float* __restrict__ tscale = nullptr) {
    constexpr int H = NS * 256;
    const int lane = threadIdx.x & 63;
    // tscale (round 6, ABI v9): the e4m3 copy with ONE per-tensor factor tscale[0] (delayed: from an earlier step's amax) instead of per-row ones -- the copy
    // the 8-bit weight gradient can read as stored (a per-row factor lies along ITS reduction index); max|y| of this launch is max-ed into tscale[3]
    const float ts = tscale ? tscale[0] : 0.f;
    float run_amax = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    typename Raw4<TX>::type nxt[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) nxt[i] = load_raw<TX>(x + row * H + i * 256 + lane * 4);
  for (; row < rows; row += stride) {
    float v[NS][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        unpack4(nxt[i], v[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[i][e];
    }
    if (row + stride < rows) {
#pragma unroll
        for (int i = 0; i < NS; ++i) nxt[i] = load_raw<TX>(x + (row + stride) * H + i * 256 + lane * 4);
    }
    const float mean = wave_sum(s) * (1.0f / H);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            ss += d * d;
        }
    const float var = wave_sum(ss) * (1.0f / H);
    const float rstd = rsqrtf(var + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        float g[4], bt[4], o[4];
        load4<float>(gamma + i * 256 + lane * 4, g);
        load4<float>(beta + i * 256 + lane * 4, bt);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = rstd * g[e];
            o[e] = v[i][e] * sc - mean * sc + bt[e];
        }
        if (y16) store4(y16 + row * H + i * 256 + lane * 4, o);
        if (y32) store4(y32 + row * H + i * 256 + lane * 4, o);
        if (y8) {                                        // keep what the bf16 consumer sees: quantise the bf16-ROUNDED row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = (float)(bf16)o[e];
                amax = fmaxf(amax, fabsf(v[i][e]));
            }
        }
    }
    if (y8) {
        // per-ROW e4m3 copy for the fp8 GEMM that consumes this LayerNorm (QKV / fc1): the wave owns the whole row, so the
        // scale costs one wave reduction and the copy one extra byte per element in the pass that is already running
        float sq;
        if (tscale) {
            run_amax = fmaxf(run_amax, amax);
            sq = ts;
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
            sq = amax > 0.f ? 448.f / amax : 1.f;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = clamp_e4m3(v[i][e] * sq);
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
            *reinterpret_cast<int*>(y8 + row * H + i * 256 + lane * 4) = w;
        }
        if (lane == 0 && !tscale) row_scale[row] = 1.f / sq;
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
  }
    if (tscale) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) run_amax = fmaxf(run_amax, __shfl_xor(run_amax, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(tscale + 3), __float_as_uint(run_amax));
    }
}

// backward: grid-stride over rows; each wave keeps per-lane partial dgamma/dbeta for its 4*NS columns,
// reduced across the 4 waves through LDS and pushed with one atomicAdd per column per block.
// Optional fused tail (transformer layers): the residual-stream gradient dx that this kernel produces is exactly the
// gradient of the PREVIOUS sub-layer's `h + dropout(branch)`; so the kernel can also emit d_branch = dropout'(dx)
// (same counter-based mask as the forward GEMM epilogue) and accumulate its column sums = that sub-layer's bias
// gradient, saving one dropout pass and one column-sum pass over [T, H] per sub-layer.
template <int NS, typename TDY, typename TX, typename TR, typename TDX>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const TR* __restrict__ dres,
                                                     TDX* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, bf16* __restrict__ dx_drop,
                                                     uint32_t drop_thresh, float drop_scale, uint64_t drop_seed,
                                                     float* __restrict__ dcolsum, uint8_t* __restrict__ db8 = nullptr, float* __restrict__ db8_scale = nullptr,
                                                     int db8_fmt = 1) {
    constexpr int H = NS * 256;
    __shared__ float red[3][4][H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // db8 (round 6, ABI v9): an 8-bit float copy (db8_fmt 0 = e4m3, 1 = e5m2) of the BRANCH gradient as the next kernels read it (bf16-rounded, behind the
    // dropout mask) with the per-tensor factor db8_scale[0] (delayed) -- the A operand of the sub-layer's 8-bit weight gradient; max|.| goes to db8_scale[3]
    const float b8s = db8 ? db8_scale[0] : 0.f;
    const float b8max = db8_fmt == 0 ? 448.f : 57344.f;
    float b8_amax = 0.f;
    float g[NS][4], dg[NS][4], db[NS][4], dc[NS][4];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        load4<float>(gamma + i * 256 + lane * 4, g[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dg[i][e] = db[i][e] = dc[i][e] = 0.f;
    }
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    typename Raw4<TDY>::type ndy[NS];
    typename Raw4<TX>::type nx[NS];
    typename Raw4<TR>::type nr[NS];
    float nmu = 0.f, nrs = 0.f;
    auto request = [&](const int64_t r) {                // every operand of row r, before the current row's reductions
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            ndy[i] = load_raw<TDY>(dy + r * H + i * 256 + lane * 4);
            nx[i] = load_raw<TX>(x + r * H + i * 256 + lane * 4);
            if (dres) nr[i] = load_raw<TR>(dres + r * H + i * 256 + lane * 4);
        }
        nmu = mean[r];
        nrs = rstd[r];
    };
    if (row < rows) request(row);
    for (; row < rows; row += stride) {
        const float mu = nmu, rs = nrs;
        float dyv[NS][4], xh[NS][4], rv[NS][4];
        float c1 = 0.f, c2 = 0.f;
        float xvv[NS][4];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            unpack4(ndy[i], dyv[i]);
            unpack4(nx[i], xvv[i]);
            if (dres) unpack4(nr[i], rv[i]);
        }
        if (row + stride < rows) request(row + stride);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float (&xv)[4] = xvv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (xv[e] - mu) * rs;
                const float gd = dyv[i][e] * g[i][e];
                c1 += gd;
                c2 += gd * xh[i][e];
                dg[i][e] += dyv[i][e] * xh[i][e];
                db[i][e] += dyv[i][e];
            }
        }
        c1 = wave_sum(c1) * (1.0f / H);
        c2 = wave_sum(c2) * (1.0f / H);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (dyv[i][e] * g[i][e] - c1 - xh[i][e] * c2);
            if (dres) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rv[i][e];
            }
            store4(dx + row * H + i * 256 + lane * 4, o);
            if (dcolsum) {
                if (drop_thresh) {
                    bool keep[4];                         // H % 4 == 0: the index of element 0 is even
                    dropout_keep_n<4>(drop_seed, (uint64_t)row * H + (uint64_t)(i * 256 + lane * 4), drop_thresh, keep);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // the branch gradient is taken from the value as the next kernels will read it (bf16)
                        const float ob = sizeof(TDX) == 2 ? (float)(bf16)o[e] : o[e];
                        o[e] = keep[e] ? ob * drop_scale : 0.f;
                    }
                    store4(dx_drop + row * H + i * 256 + lane * 4, o);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dc[i][e] += sizeof(TDX) == 2 ? (float)(bf16)o[e] : o[e];
                if (db8) {
                    float f[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (float)(bf16)o[e];
                        b8_amax = fmaxf(b8_amax, fabsf(t));
                        f[e] = fminf(fmaxf(t * b8s, -b8max), b8max);
                    }
                    int w = 0;
                    if (db8_fmt == 0) {
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
                    } else {
                        w = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w, false);
                        w = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w, true);
                    }
                    *reinterpret_cast<int*>(db8 + row * H + i * 256 + lane * 4) = w;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][i * 256 + lane * 4 + e] = dg[i][e];
            red[1][wave][i * 256 + lane * 4 + e] = db[i][e];
            red[2][wave][i * 256 + lane * 4 + e] = dc[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        const float sg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (dgamma) atomicAdd(dgamma + c, sg);
        if (dbeta) atomicAdd(dbeta + c, sb);
        if (dcolsum) atomicAdd(dcolsum + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
    }
    if (db8) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b8_amax = fmaxf(b8_amax, __shfl_xor(b8_amax, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(db8_scale + 3), __float_as_uint(b8_amax));
    }
}

template <int NS>
int ln_fwd_launch(const void* x, int x_f32, const float* gamma, const float* beta, void* y16, float* y32, float* mean,
                  float* rstd, int64_t rows, float eps, hipStream_t s, void* y8 = nullptr, float* row_scale = nullptr, float* tscale = nullptr) {
    int nblk = cdiv(rows, 4);
    if (nblk > 2048) nblk = 2048;                        // 8 workgroups per CU; the rest of the rows by the grid-stride loop
    const dim3 grid(nblk), block(256);
    if (x_f32)
        hipLaunchKernelGGL((ln_fwd_kernel<NS, float>), grid, block, 0, s, (const float*)x, gamma, beta, (bf16*)y16, y32,
                           mean, rstd, rows, eps, (uint8_t*)y8, row_scale, tscale);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<NS, bf16>), grid, block, 0, s, (const bf16*)x, gamma, beta, (bf16*)y16, y32,
                           mean, rstd, rows, eps, (uint8_t*)y8, row_scale, tscale);
    return merlot_launch_status("merlot_ln_fwd");
}

// supported dtype combinations of the backward (dy, x, dres, dx):
//   0: bf16 bf16 bf16 bf16   (transformer layers)
//   1: f32  f32  f32  f32    (heads / embedding sites)
//   2: bf16 f32  f32  f32    (embedding LN whose output was bf16, input f32)
//   3: f32  bf16 bf16 bf16
template <int NS>
int ln_bwd_launch(int combo, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                  const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, void* dx_drop, float drop_p,
                  uint64_t drop_seed, float* dcolsum, hipStream_t s, void* db8 = nullptr, float* db8_scale = nullptr, int db8_fmt = 1) {
    const uint32_t thresh = (dx_drop && drop_p > 0.f) ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float dscale = 1.0f / (1.0f - drop_p);
    int nblk = cdiv(rows, 4);
    // Every block ends in 2 - 3 atomics per column.  Fewer blocks (512 / 256 = whole multiples of the CUs, as the GroupNorm backward's sweep suggested) are 5 - 22 % faster per
    // launch in an isolated sweep (profiles/r06_z11_ln_bwd_blocks.txt) and LEVEL in the step: its main variant takes 294 - 296 us at either cap under rocprofv3, the step
    // 426.5 / 426.6 ms at either (profiles/r06_z15_ln_bwd_in_step.txt; an earlier same-box pair read +0.3 %, r06_z12) -- the cap stays.
    int cap = 2048;
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_LN_BWD_BLOCKS")) cap = atoi(e);      // scripts/exp_ln_bwd_blocks.py: every block ends in 2 - 3 atomics per column
#endif
    if (nblk > cap) nblk = cap;
    const dim3 grid(nblk), block(256);
#define LN_BWD(TDY, TX, TR, TDX)                                                                                      \
    hipLaunchKernelGGL((ln_bwd_kernel<NS, TDY, TX, TR, TDX>), grid, block, 0, s, (const TDY*)dy, (const TX*)x, mean,  \
                       rstd, gamma, (const TR*)dres, (TDX*)dx, dgamma, dbeta, rows, (bf16*)dx_drop, thresh,     \
                       dscale, drop_seed, dcolsum, (uint8_t*)db8, db8_scale, db8_fmt)
    switch (combo) {
        case 0: LN_BWD(bf16, bf16, bf16, bf16); break;
        case 1: LN_BWD(float, float, float, float); break;
        case 2: LN_BWD(bf16, float, float, float); break;
        case 3: LN_BWD(float, bf16, bf16, bf16); break;
        default: merlot_set_error("merlot_ln_bwd: unsupported dtype combination"); return MERLOT_EDTYPE;
    }
#undef LN_BWD
    return merlot_launch_status("merlot_ln_bwd");
}

}  // namespace

#define LN_DISPATCH_H(H, CALL)                                                                  \
    switch (H) {                                                                                \
        case 256: { constexpr int NS = 1; return CALL; }                                        \
        case 512: { constexpr int NS = 2; return CALL; }                                        \
        case 768: { constexpr int NS = 3; return CALL; }                                        \
        case 1024: { constexpr int NS = 4; return CALL; }                                       \
        default: merlot_set_error("LayerNorm: H=%d unsupported (256/512/768/1024)", H); return MERLOT_ESHAPE; \
    }

extern "C" int merlot_ln_fwd(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                             float* mean, float* rstd, int64_t rows, int H, float eps, merlot_stream_t stream) {
    MERLOT_CHECK(x && gamma && beta && (y_bf16 || y_f32), MERLOT_ESHAPE, "merlot_ln_fwd: null operand");
    MERLOT_CHECK(rows > 0, MERLOT_ESHAPE, "merlot_ln_fwd: rows must be > 0");
    hipStream_t s = (hipStream_t)stream;
    LN_DISPATCH_H(H, (ln_fwd_launch<NS>(x, x_f32, gamma, beta, y_bf16, y_f32, mean, rstd, rows, eps, s)));
}

extern "C" int merlot_ln_fwd_q8(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, void* y_fp8,
                                float* row_scale, float* mean, float* rstd, int64_t rows, int H, float eps, merlot_stream_t stream) {
    MERLOT_CHECK(x && gamma && beta && y_fp8 && row_scale, MERLOT_ESHAPE, "merlot_ln_fwd_q8: null operand");
    MERLOT_CHECK(rows > 0, MERLOT_ESHAPE, "merlot_ln_fwd_q8: rows must be > 0");
    hipStream_t s = (hipStream_t)stream;
    LN_DISPATCH_H(H, (ln_fwd_launch<NS>(x, x_f32, gamma, beta, y_bf16, nullptr, mean, rstd, rows, eps, s, y_fp8, row_scale)));
}

// ABI v9: merlot_ln_fwd_q8 with ONE per-tensor factor (delayed scaling: tscale = merlot_quantize_f8's block of this tensor; [0] is read, max|y| goes to [3])
extern "C" int merlot_ln_fwd_q8t(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, void* y_fp8,
                                 float* tscale, float* mean, float* rstd, int64_t rows, int H, float eps, merlot_stream_t stream) {
    MERLOT_CHECK(x && gamma && beta && y_fp8 && tscale, MERLOT_ESHAPE, "merlot_ln_fwd_q8t: null operand");
    MERLOT_CHECK(rows > 0, MERLOT_ESHAPE, "merlot_ln_fwd_q8t: rows must be > 0");
    hipStream_t s = (hipStream_t)stream;
    LN_DISPATCH_H(H, (ln_fwd_launch<NS>(x, x_f32, gamma, beta, y_bf16, nullptr, mean, rstd, rows, eps, s, y_fp8, nullptr, tscale)));
}

extern "C" int merlot_ln_bwd(const void* dy, int dy_f32, const void* x, int x_f32, const float* mean, const float* rstd,
                             const float* gamma, const void* dres, int dres_f32, void* dx, int dx_f32, float* dgamma,
                             float* dbeta, int64_t rows, int H, void* dx_drop, float drop_p, uint64_t drop_seed,
                             float* dcolsum, merlot_stream_t stream) {
    MERLOT_CHECK(dy && x && mean && rstd && gamma && dx, MERLOT_ESHAPE, "merlot_ln_bwd: null operand");
    MERLOT_CHECK(drop_p >= 0.f && drop_p < 1.f, MERLOT_ESHAPE, "merlot_ln_bwd: drop_p out of range");
    MERLOT_CHECK(!(drop_p > 0.f && dx_drop) || dcolsum, MERLOT_ESHAPE, "merlot_ln_bwd: dx_drop needs dcolsum");
    MERLOT_CHECK(rows > 0, MERLOT_ESHAPE, "merlot_ln_bwd: rows must be > 0");
    if (!dres) dres_f32 = dx_f32;
    int combo = -1;
    if (!dy_f32 && !x_f32 && !dres_f32 && !dx_f32) combo = 0;
    else if (dy_f32 && x_f32 && dres_f32 && dx_f32) combo = 1;
    else if (!dy_f32 && x_f32 && dres_f32 && dx_f32) combo = 2;
    else if (dy_f32 && !x_f32 && !dres_f32 && !dx_f32) combo = 3;
    hipStream_t s = (hipStream_t)stream;
    LN_DISPATCH_H(H, (ln_bwd_launch<NS>(combo, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, dx_drop, drop_p,
                                        drop_seed, dcolsum, s)));
}

// ABI v9: merlot_ln_bwd that also writes the 8-bit float copy of the branch gradient (see ln_bwd_kernel "db8")
extern "C" int merlot_ln_bwd_q8(const void* dy, int dy_f32, const void* x, int x_f32, const float* mean, const float* rstd,
                             const float* gamma, const void* dres, int dres_f32, void* dx, int dx_f32, float* dgamma,
                             float* dbeta, int64_t rows, int H, void* dx_drop, float drop_p, uint64_t drop_seed,
                             float* dcolsum, void* db8, int db8_fmt, float* db8_scale, merlot_stream_t stream) {
    MERLOT_CHECK(db8 && db8_scale && dcolsum && (db8_fmt == 0 || db8_fmt == 1), MERLOT_ESHAPE, "merlot_ln_bwd_q8: the 8-bit copy is of the branch gradient (needs dcolsum), db8_fmt 0 / 1");
    MERLOT_CHECK(dy && x && mean && rstd && gamma && dx, MERLOT_ESHAPE, "merlot_ln_bwd: null operand");
    MERLOT_CHECK(drop_p >= 0.f && drop_p < 1.f, MERLOT_ESHAPE, "merlot_ln_bwd: drop_p out of range");
    MERLOT_CHECK(!(drop_p > 0.f && dx_drop) || dcolsum, MERLOT_ESHAPE, "merlot_ln_bwd: dx_drop needs dcolsum");
    MERLOT_CHECK(rows > 0, MERLOT_ESHAPE, "merlot_ln_bwd: rows must be > 0");
    if (!dres) dres_f32 = dx_f32;
    int combo = -1;
    if (!dy_f32 && !x_f32 && !dres_f32 && !dx_f32) combo = 0;
    else if (dy_f32 && x_f32 && dres_f32 && dx_f32) combo = 1;
    else if (!dy_f32 && x_f32 && dres_f32 && dx_f32) combo = 2;
    else if (dy_f32 && !x_f32 && !dres_f32 && !dx_f32) combo = 3;
    hipStream_t s = (hipStream_t)stream;
    LN_DISPATCH_H(H, (ln_bwd_launch<NS>(combo, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, dx_drop, drop_p,
                                        drop_seed, dcolsum, s, db8, db8_scale, db8_fmt)));
}
