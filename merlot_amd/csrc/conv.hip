// HBM-bound pieces of the ResNet-hybrid stem (utils/vision_transformer.py:8-170,
// utils/model_utils.py:133-222): 3x3 im2col / col2im around the MFMA GEMMs, GroupNorm(32) forward / backward with the
// ReLU and the residual add of the bottleneck fused, 2x2 average pool.  Activations are NHWC bf16, C % 8 == 0 (C == 3
// only for the very first convolution, scalar path).  One lane moves 16 B (8 channels).
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int per_block, int cap = 65535) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- 3x3 im2col: out[(n,yo,xo)][(ky,kx,c)] = x[n][yo*s+ky-1][xo*s+kx-1][c] (0 outside), columns K..Kp-1 zero ----------
// pad (1,1) is both TF's SAME at stride 1 and fixed_padding(kernel 3) + VALID at stride 2 (utils/vision_transformer.py:8-19,43)
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int N, int H,
                                                        int W, int C, int s, int Ho, int Wo, int Kp, float shift) {
    const int cpr = Kp / 8;                               // 16-B chunks per output row (Kp % 8 == 0)
    const int64_t total = (int64_t)N * Ho * Wo * cpr;
    const int K = 9 * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int k0 = (int)(i - row * cpr) * 8;
        const int xo = (int)(row % Wo);
        const int yo = (int)((row / Wo) % Ho);
        const int n = (int)(row / ((int64_t)Wo * Ho));
        bf16x8 v;
        if ((C & 7) == 0 && k0 < K) {                     // one (ky,kx) tap, 8 consecutive channels
            const int tap = k0 / C, c = k0 - tap * C;
            const int y = yo * s + tap / 3 - 1, xx = xo * s + tap % 3 - 1;
            if (y >= 0 && y < H && xx >= 0 && xx < W) {
                v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)n * H + y) * W + xx) * C + c);
                if (shift != 0.f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + shift);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            }
        } else {                                          // C == 3 (first convolution) or the zero padding columns
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float val = 0.f;
                if (k < K) {
                    const int tap = k / C, c = k - tap * C;
                    const int y = yo * s + tap / 3 - 1, xx = xo * s + tap % 3 - 1;
                    if (y >= 0 && y < H && xx >= 0 && xx < W) val = (float)x[(((int64_t)n * H + y) * W + xx) * C + c] + shift;
                }
                v[e] = (bf16)val;
            }
        }
        *reinterpret_cast<bf16x8*>(out + row * Kp + k0) = v;
    }
}

// ---- patch im2col: out[(n,ph,pw)][(py,px,c)] = image[n][ph*P+py][pw*P+px][c] + shift, non-overlapping PxP patches ----------
// (utils/vision_transformer.py:193-205: `image - 0.5`, 16x16/16 VALID conv; k order = HWIO flattening).  A patch row is
// P segments of 3P contiguous bf16 (96 B at P = 16): one lane moves 16 B.
__global__ __launch_bounds__(256) void im2col_patch_kernel(const bf16* __restrict__ img, bf16* __restrict__ out, int N, int H,
                                                          int W, int P, int h1, int w1, float shift) {
    const int seg = 3 * P / 8;                            // 16-B chunks per patch row segment (6 at P = 16)
    const int cpr = P * seg;                              // chunks per output row
    const int64_t total = (int64_t)N * h1 * w1 * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        const int py = ch / seg, cx = ch - py * seg;
        const int pw = (int)(row % w1), ph = (int)((row / w1) % h1);
        const int64_t n = row / ((int64_t)w1 * h1);
        bf16x8 v = *reinterpret_cast<const bf16x8*>(img + ((n * H + ph * P + py) * W + pw * P) * 3 + cx * 8);
        if (shift != 0.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + shift);
        }
        *reinterpret_cast<bf16x8*>(out + row * (3 * P * P) + py * 3 * P + cx * 8) = v;
    }
}

// ---- col2im (input gradient of the 3x3 convolution): dx[n][y][x][c] = sum over taps of dP[(n,yo,xo)][(ky,kx,c)] ----------
__global__ __launch_bounds__(256) void col2im3x3_kernel(const bf16* __restrict__ dp, bf16* __restrict__ dx, int N, int H,
                                                        int W, int C, int s, int Ho, int Wo, int Kp) {
    const int cpr = C / 8;
    const int64_t total = (int64_t)N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / cpr;
        const int c = (int)(i - pix * cpr) * 8;
        const int xx = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const int n = (int)(pix / ((int64_t)W * H));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = y + 1 - ky;
            if (ty < 0 || ty % s != 0 || ty / s >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = xx + 1 - kx;
                if (tx < 0 || tx % s != 0 || tx / s >= Wo) continue;
                const int64_t row = ((int64_t)n * Ho + ty / s) * Wo + tx / s;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dp + row * Kp + (ky * 3 + kx) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
        *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
    }
}

// ---- GroupNorm ----------------------------------------------------------------------------------------------------------
// stats[n][g] = {sum x, sum x^2} over (H*W, C/G): the one-pass moments the reference asks for (mean_close_to_zero=True,
// utils/model_utils.py:196-201).  A block owns one sample and one slice of its positions; a thread always touches the
// same 8 channels, so it reduces in registers and the block folds channel chunks into groups through LDS.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16* __restrict__ x, float* __restrict__ stats, int HW, int C,
                                                       int G, int pos_per_block) {
    extern __shared__ float gred[];                        // [G][2]
    const int n = blockIdx.x;
    const int cpr = C / 8, cpg = C / G;
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) gred[i] = 0.f;
    __syncthreads();
    const int p0 = blockIdx.y * pos_per_block, p1 = min(HW, p0 + pos_per_block);
    const int chunk = threadIdx.x % cpr;                   // blockDim.x % cpr == 0 (host guarantees)
    const int prow = threadIdx.x / cpr, pstep = blockDim.x / cpr;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (prow < pstep) {
        for (int p = p0 + prow; p < p1; p += pstep) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((int64_t)n * HW + p) * C + chunk * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s1[e] += f;
                s2[e] += f * f;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (chunk * 8 + e) / cpg;
        atomicAdd(&gred[2 * g], s1[e]);
        atomicAdd(&gred[2 * g + 1], s2[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) atomicAdd(stats + (int64_t)n * 2 * G + i, gred[i]);
}

// {sum, sum of squares} -> {mean, rstd} in place (var = E[x^2] - E[x]^2, the reference's one-pass form)
__global__ void gn_finalize_kernel(float* __restrict__ stats, int64_t n_groups, float inv_cnt, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups) {
        const float mean = stats[2 * i] * inv_cnt;
        const float var = stats[2 * i + 1] * inv_cnt - mean * mean;
        stats[2 * i] = mean;
        stats[2 * i + 1] = rsqrtf(var + eps);
    }
}

// GroupNorm's affine output y = x * sc + sh and its ReLU mask, in ONE place: gn_apply_kernel writes y, and the two backward kernels recompute
// the mask from x where no residual joins (no second read of y).  The three are separately compiled loops: the mask is only the
// forward's if they evaluate the same expression with the same roundings -- explicit FMAs (no contraction left to the compiler), and
// the test is made on the value the forward STORED (bf16), so an element that rounds to zero is masked on both sides.
__device__ __forceinline__ void gn_affine(float mean, float rstd, float gamma, float beta, float& sc, float& sh) {
    sc = rstd * gamma;
    sh = __builtin_fmaf(-mean, sc, beta);
}
__device__ __forceinline__ float gn_y(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }
__device__ __forceinline__ bool gn_relu_passes(float yv) { return (float)(bf16)yv > 0.f; }

// y = [relu]( (x - mean) * rstd * gamma + beta [+ res] ).  Grid (sample, position slice) like the statistics kernel: a thread always
// touches the same 8 channels, so gamma, beta and the (mean, rstd) of its channels' groups are loaded ONCE per thread -- the first
// version looked them up per element with an integer division by the run-time group width each, and ran at 60-70 % of the HBM rate
// (round 4, profiles/r04_r_groupnorm_kernels.txt).
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const bf16* __restrict__ res, bf16* __restrict__ y, int HW,
                                                       int C, int G, int relu, int pos_per_block) {
    const int n = blockIdx.x;
    const int cpr = C / 8, cpg = C / G;
    const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr, pstep = blockDim.x / cpr;
    const int p0 = blockIdx.y * pos_per_block, p1 = min(HW, p0 + pos_per_block);
    float sc[8], sh[8];                                    // y = x * sc + sh
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e, g = c / cpg;
        const float mean = stats[((int64_t)n * G + g) * 2], rstd = stats[((int64_t)n * G + g) * 2 + 1];
        gn_affine(mean, rstd, gamma[c], beta[c], sc[e], sh[e]);
    }
    for (int p = p0 + prow; p < p1; p += pstep) {
        const int64_t off = ((int64_t)n * HW + p) * C + chunk * 8;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + off);
        bf16x8 r8;
        if (res) r8 = *reinterpret_cast<const bf16x8*>(res + off);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = gn_y((float)v[e], sc[e], sh[e]);
            if (res) f += (float)r8[e];
            if (relu) f = fmaxf(f, 0.f);
            o[e] = (bf16)f;
        }
        *reinterpret_cast<bf16x8*>(y + off) = o;
    }
}

// backward, pass 1: with dy' = relu ? dy * (y > 0) : dy and xhat = (x - mean) * rstd:
//   dgamma[c] += sum dy' * xhat, dbeta[c] += sum dy'          (over samples and positions; atomics per block)
//   gsum[n][g] = {sum_c gamma_c * dbeta_c(n), sum_c gamma_c * dgamma_c(n)}   (per sample)
// (relu with y == nullptr: the ReLU mask is recomputed from x -- y = (x - mean) * rstd * gamma + beta, the forward's own expression --
// instead of reading the stored output: one tensor pass less in each of the two backward kernels; layers with a residual add pass y)
template <int U>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y,
                                                           const bf16* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ gsum, int HW, int C,
                                                           int G, float eps, int relu, int pos_per_block) {
    extern __shared__ float sm[];                          // [C][2] per-channel partials of this block
    const int n = blockIdx.x;
    const int cpr = C / 8, cpg = C / G;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int p0 = blockIdx.y * pos_per_block, p1 = min(HW, p0 + pos_per_block);
    const int chunk = threadIdx.x % cpr;
    const int prow = threadIdx.x / cpr, pstep = blockDim.x / cpr;
    float mean[8], rstd[8], gam[8], bet[8], dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool from_x = relu && y == nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (chunk * 8 + e) / cpg;
        mean[e] = stats[((int64_t)n * G + g) * 2];
        rstd[e] = stats[((int64_t)n * G + g) * 2 + 1];
        gam[e] = bet[e] = 0.f;
        if (from_x) gn_affine(mean[e], rstd[e], gamma[chunk * 8 + e], beta[chunk * 8 + e], gam[e], bet[e]);     // y = x * gam + bet
    }
    if (prow < pstep) {
        // U positions per iteration, every load of the iteration issued before the first use (one block per sample since round 6: 14 waves per CU have to keep the
        // memory pipe full on their own; profiles/r06_z18_gn_unroll.txt)
        for (int pb = p0 + prow; pb < p1; pb += U * pstep) {
            bf16x8 d8[U], x8[U], y8[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = pb + u * pstep;
                if (p < p1) {
                    const int64_t off = ((int64_t)n * HW + p) * C + chunk * 8;
                    d8[u] = *reinterpret_cast<const bf16x8*>(dy + off);
                    x8[u] = *reinterpret_cast<const bf16x8*>(x + off);
                    if (relu && !from_x) y8[u] = *reinterpret_cast<const bf16x8*>(y + off);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pb + u * pstep < p1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float d = (float)d8[u][e];
                        const float yv = from_x ? gn_y((float)x8[u][e], gam[e], bet[e]) : (float)y8[u][e];
                        if (relu && !gn_relu_passes(yv)) d = 0.f;
                        dg[e] += d * ((float)x8[u][e] - mean[e]) * rstd[e];
                        db[e] += d;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(&sm[2 * (chunk * 8 + e)], dg[e]);
        atomicAdd(&sm[2 * (chunk * 8 + e) + 1], db[e]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float g_ = sm[2 * c], b_ = sm[2 * c + 1];
        atomicAdd(dgamma + c, g_);
        atomicAdd(dbeta + c, b_);
        const int g = c / cpg;
        atomicAdd(gsum + ((int64_t)n * G + g) * 2, gamma[c] * b_);
        atomicAdd(gsum + ((int64_t)n * G + g) * 2 + 1, gamma[c] * g_);
    }
}

// backward, pass 2: dx = rstd * (dy' * gamma - gsum0 / cnt - xhat * gsum1 / cnt); optionally dres = dy' (residual branch).
// Same indexing as gn_apply_kernel: per-thread constants k1 = rstd * gamma, k2 = rstd * gsum0 / cnt, k3 = rstd * gsum1 / cnt, and
// xhat = x * rstd - mean * rstd.
template <int U>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y,
                                                           const bf16* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ gsum, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16* __restrict__ dx,
                                                           bf16* __restrict__ dres, int HW, int C, int G, int relu,
                                                           int pos_per_block) {
    const int n = blockIdx.x;
    const int cpr = C / 8, cpg = C / G;
    const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
    const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr, pstep = blockDim.x / cpr;
    const int p0 = blockIdx.y * pos_per_block, p1 = min(HW, p0 + pos_per_block);
    const bool from_x = relu && y == nullptr;
    float rs[8], mr[8], k1[8], k2[8], k3[8], gam[8], bet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e, g = c / cpg;
        const float mean = stats[((int64_t)n * G + g) * 2], rstd = stats[((int64_t)n * G + g) * 2 + 1];
        rs[e] = rstd;
        mr[e] = mean * rstd;
        gam[e] = gamma[c];
        k1[e] = rstd * gam[e];
        bet[e] = 0.f;
        if (from_x) gn_affine(mean, rstd, gam[e], beta[c], k1[e], bet[e]);      // from_x: y = x * k1 + bet, gn_apply_kernel's own expression
        k2[e] = rstd * gsum[((int64_t)n * G + g) * 2] * inv_cnt;
        k3[e] = rstd * gsum[((int64_t)n * G + g) * 2 + 1] * inv_cnt;
    }
    for (int pb = p0 + prow; pb < p1; pb += U * pstep) {     // (U positions per iteration, loads first: see gn_bwd_stats_kernel)
        bf16x8 d8[U], x8[U], y8[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = pb + u * pstep;
            if (p < p1) {
                const int64_t off = ((int64_t)n * HW + p) * C + chunk * 8;
                d8[u] = *reinterpret_cast<const bf16x8*>(dy + off);
                x8[u] = *reinterpret_cast<const bf16x8*>(x + off);
                if (relu && !from_x) y8[u] = *reinterpret_cast<const bf16x8*>(y + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = pb + u * pstep;
            if (p < p1) {
                const int64_t off = ((int64_t)n * HW + p) * C + chunk * 8;
                bf16x8 o, r;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = (float)d8[u][e];
                    const float xhat = __builtin_fmaf((float)x8[u][e], rs[e], -mr[e]);
                    const float yv = from_x ? gn_y((float)x8[u][e], k1[e], bet[e]) : (float)y8[u][e];
                    if (relu && !gn_relu_passes(yv)) d = 0.f;
                    o[e] = (bf16)(d * k1[e] - k2[e] - xhat * k3[e]);
                    r[e] = (bf16)d;
                }
                *reinterpret_cast<bf16x8*>(dx + off) = o;
                if (dres) *reinterpret_cast<bf16x8*>(dres + off) = r;
            }
        }
    }
}

// ---- GroupNorm in ONE launch per direction (round 6, ABI v10) ---------------------------------------------------------------
// The two-launch forms above read x twice forward (moments, then apply) and x | dy twice backward: 8 tensor passes per layer where
// 5 are algorithmic, and between the launches the whole activation (up to 1.9 GB at the as-shipped geometry) passes through the
// caches, so the second read is an HBM read.  Here a workgroup keeps its slice of the sample IN REGISTERS between the two phases:
//   claim an item (sample n, slice) -> load the slice, reduce its moments -> add them to the sample's sums (device-scope atomics)
//   -> count itself in arrive[n] -> wait until the sample's `split` slices are counted -> normalise the held slice, store.
// Items are CLAIMED from a counter, not derived from blockIdx: a workgroup only ever waits for items with smaller claim numbers
// than slices it could itself still be waiting for -- items that running workgroups hold and finish without waiting for anybody --
// so the wait terminates whatever order the hardware dispatches workgroups in and however many of them are resident.
// Workspace (caller-owned, its control words zeroed by the entry point): u32 ctr[16] | u32 ctl[N][1024] (a sample's arrival counter and "published" flag: one 4 KiB block
// per sample, so that the polls and arrivals of the few dozen samples in flight spread over the memory channels) | forward: f32 sums[N][G][2]; backward:
// f32 slots[N][split][C + G][2] -- one slot per (sample, slice) for the slice's channel sums {dgamma_c, dbeta_c} and its two sums per group.
// FORWARD: every access to the shared sums is a RELAXED device-scope atomic (performed at the memory side, coherent between the XCDs' L2s) and the order "my additions,
// then my arrival" comes from WAITING FOR THE ADDITIONS' RETURN VALUES (2G per workgroup), not from fences: a release / acquire pair at device scope is an L2 write-back +
// invalidate (buffer_wbl2 / buffer_inv sc1) per wave -- with tens of thousands of workgroups per launch, each between other workgroups' output stores, the first version of
// these kernels (release on the arrival, __threadfence() in front of it) ran the as-shipped step into a 600 s timeout where the two-launch form takes 0.4 s.
// BACKWARD: 4C returning atomics per workgroup were the whole cost of its first version; now a workgroup STORES its sums to its slot (write-through, waited for with
// s_waitcnt), arrives with one atomic, the sample's last arriver adds the slots up -- channel sums into dgamma / dbeta (N atomics per address, not N * split), group sums into
// gsum, published with a flag the other slices poll.
__device__ __forceinline__ float gn_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gn_add(float* p, float v) {   // returns only when the addition has been performed
    const float old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
}
// one thread, behind a workgroup barrier that follows every thread's gn_add: count this workgroup in, wait for the sample's other slices; returns the arrival number
__device__ __forceinline__ unsigned gn_arrive_and_wait(unsigned* slot, unsigned split) {
    const unsigned old = __hip_atomic_fetch_add(slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split) __builtin_amdgcn_s_sleep(24);
    return old;
}
// one arrival counter per 4 KiB: a few dozen samples are in flight at a time and every waiting workgroup polls its sample's counter -- with the counters of
// consecutive samples 4 B apart, every poll and every arrival of the launch landed in one or two memory channels (calls 34 / 35: some shapes did not finish)
constexpr int GN_ARRIVE_STRIDE = 1024;
// a pair of floats as ONE relaxed device-scope access (8-B aligned): a write-through store / a load that does not trust this XCD's L2
__device__ __forceinline__ void gn_st2(float* p, float a, float b) {
    const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gn_ld2(const float* p, float& a, float& b) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((unsigned)v);
    b = __uint_as_float((unsigned)(v >> 32));
}
__host__ __device__ constexpr int64_t gn_ws_arrive_words(int N) { return (int64_t)N * GN_ARRIVE_STRIDE; }

template <int ITER, bool RES, bool HOLD = false>
__global__ __launch_bounds__(256, 2) void gn_fwd_fused_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const bf16* __restrict__ res,
                                                           bf16* __restrict__ y, float* __restrict__ stats, unsigned* __restrict__ ws,
                                                           int N, int HW, int C, int G, int relu, int split, float inv_cnt, float eps, int static_items) {
    extern __shared__ float gsm[];                         // [G][2] this slice's sums, then [G][2] {mean, rstd} of the sample
    __shared__ int s_item;
    unsigned* arrive = ws + 16;
    float* sums = reinterpret_cast<float*>(ws + 16 + gn_ws_arrive_words(N));
    const int tid = threadIdx.x, bd = blockDim.x;
    // (static_items: experiments only -- item = workgroup id instead of a claim, scripts/exp_gn_slices.py: the test of the dispatch model in the GN_FUSED_MAX_SLICES comment)
    if (tid == 0) s_item = static_items ? (int)blockIdx.x : (int)__hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < 2 * G; i += bd) gsm[i] = 0.f;
    __syncthreads();
    const int item = s_item, n = item / split, sl = item - n * split;
    const int cpr = C / 8, cpg = C / G;
    const int chunk = tid % cpr, prow = tid / cpr, pstep = bd / cpr;
    const int pbase = sl * (ITER * pstep) + prow;
    const int64_t off0 = ((int64_t)n * HW) * C + chunk * 8;
    bf16x8 v[ITER], rh[(RES && HOLD) ? ITER : 1];          // HOLD: the residual requested in front of the wait and held across it
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int p = pbase + i * pstep;
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
        v[i] = z;
        if (p < HW) v[i] = *reinterpret_cast<const bf16x8*>(x + off0 + (int64_t)p * C);
    }
    if (RES && HOLD) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int p = pbase + i * pstep;
            if (p < HW) rh[i] = *reinterpret_cast<const bf16x8*>(res + off0 + (int64_t)p * C);
        }
    }
    float gam[8], bet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gam[e] = gamma[chunk * 8 + e];
        bet[e] = beta[chunk * 8 + e];
    }
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[i][e];                // rows beyond HW hold zeros
            s1[e] += f;
            s2[e] = __builtin_fmaf(f, f, s2[e]);
        }
    int gidx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gidx[e] = (chunk * 8 + e) / cpg;
    {                                                      // channels of one group first summed in the thread: 2 LDS atomics per group it touches
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a1 += s1[e];
            a2 += s2[e];
            if (e == 7 || gidx[e + 1 < 8 ? e + 1 : 7] != gidx[e]) {
                atomicAdd(&gsm[2 * gidx[e]], a1);
                atomicAdd(&gsm[2 * gidx[e] + 1], a2);
                a1 = a2 = 0.f;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * G; i += bd) gn_add(sums + (int64_t)n * 2 * G + i, gsm[i]);
    __syncthreads();                                       // every thread's additions are performed (returned) before thread 0 counts the workgroup in
    if (tid == 0) gn_arrive_and_wait(arrive + (int64_t)n * GN_ARRIVE_STRIDE, (unsigned)split);
    __syncthreads();
    for (int g = tid; g < G; g += bd) {
        const float mean = gn_ld(sums + ((int64_t)n * G + g) * 2) * inv_cnt;
        const float var = gn_ld(sums + ((int64_t)n * G + g) * 2 + 1) * inv_cnt - mean * mean;
        const float rstd = rsqrtf(var + eps);
        gsm[2 * G + 2 * g] = mean;
        gsm[2 * G + 2 * g + 1] = rstd;
        if (sl == 0) {
            stats[((int64_t)n * G + g) * 2] = mean;
            stats[((int64_t)n * G + g) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gn_affine(gsm[2 * G + 2 * gidx[e]], gsm[2 * G + 2 * gidx[e] + 1], gam[e], bet[e], sc[e], sh[e]);
    // The residual: either held across the wait (HOLD, 8 positions per thread) or read HERE, behind it, with 16 positions per thread -- half as many slices per sample.
    // Measured (profiles/r06_z3_gn_fused_fwd.txt | r06_z4_gn_fused_fwd.txt): 48 x 88 x 256 (66 | 33 slices) 1 520 | 1 330 us, 24 x 44 x 512 (33 | 17) 618 | 638,
    // 12 x 22 x 1024 (17 | 9) 297 | 333: the host takes the late read where holding would cut a sample into more than 40 slices.
    constexpr int RB = RES ? 4 : 1;                        // residual loads in flight per thread
#pragma unroll
    for (int i0 = 0; i0 < ITER; i0 += RB) {
        bf16x8 r8[RB];
        if (RES) {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int p = pbase + (i0 + j) * pstep;
                if (!HOLD && p < HW) r8[j] = *reinterpret_cast<const bf16x8*>(res + off0 + (int64_t)p * C);
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int i = i0 + j, p = pbase + i * pstep;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = gn_y((float)v[i][e], sc[e], sh[e]);
                if (RES) f += (float)(HOLD ? rh[i][e] : r8[j][e]);
                if (relu) f = fmaxf(f, 0.f);
                o[e] = (bf16)f;
            }
            if (p < HW) *reinterpret_cast<bf16x8*>(y + off0 + (int64_t)p * C) = o;
        }
    }
}

// backward in one launch: phase 1 = gn_bwd_stats_kernel's sums on the held slice (x, dy' kept as bf16: dy' is dy or 0), phase 2 = gn_bwd_apply_kernel's
// expression.  HAS_Y: the ReLU mask from the stored output (layers with a residual add); otherwise recomputed from x (relu) or absent.
template <int ITER, bool HAS_Y>
__global__ __launch_bounds__(256, 2) void gn_bwd_fused_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y, const bf16* __restrict__ x,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ gsum, bf16* __restrict__ dx, bf16* __restrict__ dres,
                                                           unsigned* __restrict__ ws, int N, int HW, int C, int G, int relu, int split, int static_items) {
    extern __shared__ float gsm[];                         // [C][2] per-channel partials of this slice, then [G][2] the sample's gsum
    __shared__ int s_item, s_last;
    unsigned* arrive = ws + 16;
    float* slots = reinterpret_cast<float*>(ws + 16 + gn_ws_arrive_words(N));     // [N][split][C + G][2]
    const int tid = threadIdx.x, bd = blockDim.x;
    // (static_items: experiments only -- item = workgroup id instead of a claim, scripts/exp_gn_slices.py: the test of the dispatch model in the GN_FUSED_MAX_SLICES comment)
    if (tid == 0) s_item = static_items ? (int)blockIdx.x : (int)__hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < 2 * (C + G); i += bd) gsm[i] = 0.f;
    __syncthreads();
    const int item = s_item, n = item / split, sl = item - n * split;
    const int cpr = C / 8, cpg = C / G;
    const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
    const int chunk = tid % cpr, prow = tid / cpr, pstep = bd / cpr;
    const int pbase = sl * (ITER * pstep) + prow;
    const int64_t off0 = ((int64_t)n * HW) * C + chunk * 8;
    const bool from_x = relu && !HAS_Y;
    bf16x8 xv[ITER], dv[ITER];
    bf16x8 zero8;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero8[e] = (bf16)0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int p = pbase + i * pstep;
        xv[i] = zero8;
        dv[i] = zero8;
        if (p < HW) {
            xv[i] = *reinterpret_cast<const bf16x8*>(x + off0 + (int64_t)p * C);
            dv[i] = *reinterpret_cast<const bf16x8*>(dy + off0 + (int64_t)p * C);
        }
    }
    float mean[8], rs[8], gam[8], k1[8], bet[8];
    int gidx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e;
        gidx[e] = c / cpg;
        mean[e] = stats[((int64_t)n * G + gidx[e]) * 2];
        rs[e] = stats[((int64_t)n * G + gidx[e]) * 2 + 1];
        gam[e] = gamma[c];
        k1[e] = rs[e] * gam[e];
        bet[e] = 0.f;
        if (from_x) gn_affine(mean[e], rs[e], gam[e], beta[c], k1[e], bet[e]);     // y = x * k1 + bet, gn_apply_kernel's own expression
    }
    float dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        bf16x8 y8 = zero8;
        if (HAS_Y) {
            const int p = pbase + i * pstep;
            if (p < HW) y8 = *reinterpret_cast<const bf16x8*>(y + off0 + (int64_t)p * C);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float d = (float)dv[i][e];
            const float yv = from_x ? gn_y((float)xv[i][e], k1[e], bet[e]) : (float)y8[e];
            if (relu && !gn_relu_passes(yv)) {
                d = 0.f;
                dv[i][e] = (bf16)0.f;
            }
            dg[e] += d * ((float)xv[i][e] - mean[e]) * rs[e];
            db[e] += d;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(&gsm[2 * (chunk * 8 + e)], dg[e]);
        atomicAdd(&gsm[2 * (chunk * 8 + e) + 1], db[e]);
    }
    __syncthreads();
    // This slice's sums go to ITS slot of the workspace with write-through stores, not into shared sums with atomics (the first version: four RETURNING device-scope
    // atomics per channel and workgroup -- x1.4 ... x15 the two-launch time, profiles/r06_z2_gn_fused_shapes.txt): per workgroup one claim, one arrival, one poll loop.
    float* gs = gsm + 2 * C;                               // [G][2]: this slice's {sum gamma * dbeta_c, sum gamma * dgamma_c} per group
    const int P = C + G;                                   // pairs per slot: [C] {dgamma_c, dbeta_c} | [G] the two group sums
    float* const slot0 = slots + (int64_t)n * split * 2 * P;
    float* const slot = slot0 + (int64_t)sl * 2 * P;
    for (int c = tid; c < C; c += bd) {
        const float g_ = gsm[2 * c], b_ = gsm[2 * c + 1];
        gn_st2(slot + 2 * c, g_, b_);
        const int g = c / cpg;
        atomicAdd(&gs[2 * g], gamma[c] * b_);
        atomicAdd(&gs[2 * g + 1], gamma[c] * g_);
    }
    __syncthreads();
    for (int g = tid; g < G; g += bd) gn_st2(slot + 2 * (C + g), gs[2 * g], gs[2 * g + 1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's stores are performed (sc1: written through) ...
    __syncthreads();                                       // ... and so are every thread's, before thread 0 counts the workgroup in
    unsigned* const ctl = arrive + (int64_t)n * GN_ARRIVE_STRIDE;      // [0] arrivals, [32] "the sample's group sums are in gsum" (another 128-B line)
    if (tid == 0) s_last = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)split - 1u;
    __syncthreads();
    if (s_last) {
        // the sample's LAST arriver adds up the slots (its own included): channel sums into the layer's gradient, group sums into gsum for everybody
        for (int i = tid; i < 2 * P; i += bd) gsm[i] = 0.f;
        __syncthreads();
        const int SG = bd / P > 0 ? bd / P : 1;            // slice subsets walked side by side when there are fewer pairs than threads
        for (int i = tid; i < P * SG; i += bd) {
            const int pr = i % P, s0 = i / P;
            float a0 = 0.f, a1 = 0.f;
            int sidx = s0;
            for (; sidx + 3 * SG < split; sidx += 4 * SG) {            // four loads in flight
                float u[4], w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) gn_ld2(slot0 + (int64_t)(sidx + k * SG) * 2 * P + 2 * pr, u[k], w[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0 += u[k];
                    a1 += w[k];
                }
            }
            for (; sidx < split; sidx += SG) {
                float u, w;
                gn_ld2(slot0 + (int64_t)sidx * 2 * P + 2 * pr, u, w);
                a0 += u;
                a1 += w;
            }
            atomicAdd(&gsm[2 * pr], a0);
            atomicAdd(&gsm[2 * pr + 1], a1);
        }
        __syncthreads();
        for (int c = tid; c < C; c += bd) {
            atomicAdd(dgamma + c, gsm[2 * c]);
            atomicAdd(dbeta + c, gsm[2 * c + 1]);
        }
        for (int g = tid; g < G; g += bd) gn_st2(gsum + ((int64_t)n * G + g) * 2, gs[2 * g], gs[2 * g + 1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ctl + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (tid == 0)
            while (__hip_atomic_load(ctl + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(24);
        __syncthreads();
        for (int g = tid; g < G; g += bd) gn_ld2(gsum + ((int64_t)n * G + g) * 2, gs[2 * g], gs[2 * g + 1]);
    }
    __syncthreads();
    float mr[8], k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mr[e] = mean[e] * rs[e];
        k1[e] = rs[e] * gam[e];
        k2[e] = rs[e] * gs[2 * gidx[e]] * inv_cnt;
        k3[e] = rs[e] * gs[2 * gidx[e] + 1] * inv_cnt;
    }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int p = pbase + i * pstep;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)dv[i][e];
            const float xhat = __builtin_fmaf((float)xv[i][e], rs[e], -mr[e]);
            o[e] = (bf16)(d * k1[e] - k2[e] - xhat * k3[e]);
        }
        if (p < HW) {
            *reinterpret_cast<bf16x8*>(dx + off0 + (int64_t)p * C) = o;
            if (dres) *reinterpret_cast<bf16x8*>(dres + off0 + (int64_t)p * C) = dv[i];
        }
    }
}

// ---- 2x2 / stride 2 average pool (tf.nn.avg_pool2d, even H and W) --------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t N, int H,
                                                           int W, int C) {
    const int cpr = C / 8, Ho = H / 2, Wo = W / 2;
    const int64_t total = N * Ho * Wo * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * 8;
        const int64_t pix = i / cpr;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho);
        const int64_t n = pix / ((int64_t)Wo * Ho);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((n * H + 2 * yo + dy) * W + 2 * xo + dx) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(0.25f * acc[e]);
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int64_t N, int H,
                                                           int W, int C) {
    const int cpr = C / 8, Ho = H / 2, Wo = W / 2;
    const int64_t total = N * H * W * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * 8;
        const int64_t pix = i / cpr;
        const int xx = (int)(pix % W), y = (int)((pix / W) % H);
        const int64_t n = pix / ((int64_t)W * H);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + ((n * Ho + y / 2) * Wo + xx / 2) * C + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(0.25f * (float)v[e]);
        *reinterpret_cast<bf16x8*>(dx + i * 8) = o;
    }
}

int gn_block_threads(int C) {                             // multiple of C/8 chunks, <= 256
    const int cpr = C / 8;
    int t = (256 / cpr) * cpr;
    return t < cpr ? cpr : t;
}

}  // namespace

#define CONV_CHECK_GEOM(name)                                                                                     \
    MERLOT_CHECK(x && N > 0 && H > 0 && W > 0 && C > 0, MERLOT_ESHAPE, name ": bad geometry");                   \
    MERLOT_CHECK((reinterpret_cast<uintptr_t>(x) & 15) == 0, MERLOT_EALIGN, name ": operands must be 16-B aligned")

extern "C" int merlot_im2col3x3(const void* x, void* out, int N, int H, int W, int C, int stride, int Kp, float shift,
                                merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_im2col3x3");
    MERLOT_CHECK(out && (stride == 1 || stride == 2) && H % stride == 0 && W % stride == 0, MERLOT_ESHAPE,
                 "merlot_im2col3x3: stride must be 1 or 2 and divide H, W");
    MERLOT_CHECK((C % 8 == 0 || C == 3) && Kp % 8 == 0 && Kp >= 9 * C, MERLOT_ESHAPE,
                 "merlot_im2col3x3: C must be 3 or a multiple of 8, Kp a multiple of 8 and >= 9*C");
    const int Ho = H / stride, Wo = W / stride;
    const int64_t total = (int64_t)N * Ho * Wo * (Kp / 8);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out,
                       N, H, W, C, stride, Ho, Wo, Kp, shift);
    return merlot_launch_status("merlot_im2col3x3");
}

extern "C" int merlot_im2col_patches(const void* image, void* patches, int n_img, int H, int W, int P, float shift,
                                     merlot_stream_t stream) {
    MERLOT_CHECK(image && patches && n_img > 0, MERLOT_ESHAPE, "merlot_im2col_patches: null operand");
    // a lane moves 16 B of a patch row segment (3 P bf16) and the GEMMs take K = 3 P^2 in tiles of 64: P = 8, 16 (merlot.yaml), 24, 32, ...
    MERLOT_CHECK(P >= 8 && P % 8 == 0 && (3 * P * P) % 64 == 0, MERLOT_ESHAPE,
                 "patch embed: patch_size must be a multiple of 8 with 3 * P * P a multiple of 64 (got %d)", P);
    MERLOT_CHECK(H % P == 0 && W % P == 0, MERLOT_ESHAPE, "patch embed: H, W must be multiples of P");
    MERLOT_CHECK((W * 3) % 8 == 0 && (reinterpret_cast<uintptr_t>(image) & 15) == 0, MERLOT_EALIGN,
                 "patch embed: W*3 must be a multiple of 8 and the image 16-B aligned");
    const int h1 = H / P, w1 = W / P;
    const int64_t total = (int64_t)n_img * h1 * w1 * P * (3 * P / 8);
    hipLaunchKernelGGL(im2col_patch_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)image,
                       (bf16*)patches, n_img, H, W, P, h1, w1, shift);
    return merlot_launch_status("merlot_im2col_patches");
}

extern "C" int merlot_col2im3x3(const void* x, void* dx, int N, int H, int W, int C, int stride, int Kp, merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_col2im3x3");
    MERLOT_CHECK(dx && (stride == 1 || stride == 2) && H % stride == 0 && W % stride == 0 && C % 8 == 0 && Kp >= 9 * C && Kp % 8 == 0,
                 MERLOT_ESHAPE, "merlot_col2im3x3: bad arguments");
    const int64_t total = (int64_t)N * H * W * (C / 8);
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)dx, N,
                       H, W, C, stride, H / stride, W / stride, Kp);
    return merlot_launch_status("merlot_col2im3x3");
}

// Round 6: the two passes of a direction can walk the batch in GROUPS OF SAMPLES whose tensors fit the 256 MiB Infinity Cache, so that the
// second pass finds what the first one read (x forward; x | dy [| y] backward) on the die instead of in HBM -- 8 tensor passes per layer would become 5 HBM
// passes without any workgroup waiting for another (the one-launch form above does that, and loses: profiles/r06_y_gn_fused.txt).  Same kernels, pointer
// offsets per group.  MEASURED (profiles/r06_z_gn_groups.txt): with 64 MiB groups in both directions the as-shipped step went 400 -> 711 ms.  A group is a
// few dozen samples and still has to fill 256 CUs, so a sample is cut into up to 64 slices instead of 3 -- and every backward block ends in 4C atomics on
// dgamma / dbeta: 24 x 44 x 512 backward 2.0 -> 20 ms per call.  Groups are therefore OFF in the product (0 = the whole batch per pass, rounds 3 - 5); the
// experiments build reads MERLOT_GN_GROUP_MB / MERLOT_GN_GROUP_MB_BWD for the per-shape sweep (scripts/exp_gn_groups.py).
constexpr int64_t GN_GROUP_BYTES_FWD = 0, GN_GROUP_BYTES_BWD = 0;
static int64_t gn_group_bytes(bool bwd) {
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv(bwd ? "MERLOT_GN_GROUP_MB_BWD" : "MERLOT_GN_GROUP_MB")) return (int64_t)atoi(e) << 20;
#endif
    return bwd ? GN_GROUP_BYTES_BWD : GN_GROUP_BYTES_FWD;
}
// samples per group: whole samples, at least one; 0 bytes = everything in one group
static int gn_group_samples(int N, int64_t bytes_per_sample, bool bwd) {
    const int64_t lim = gn_group_bytes(bwd);
    if (lim <= 0) return N;
    int64_t nb = lim / bytes_per_sample;
    if (nb < 1) nb = 1;
    return nb > N ? N : (int)nb;
}
constexpr int GN_BWD_UNROLL = 1;                          // positions per loop iteration of the two-launch backward (sweep: profiles/r06_z18_gn_unroll.txt)
static int gn_bwd_unroll() {
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_GN_UNROLL")) return atoi(e);
#endif
    return GN_BWD_UNROLL;
}
// Slices per sample = blocks per launch / samples.  Forward (2G trailing atomics per block): enough blocks to fill the chip, >= ~2048.  BACKWARD: every block ends in 4C
// atomics on dgamma / dbeta / the group sums, and the sweep over the thirteen as-shipped shapes at 896 / 448 / 224 / 64 frames (scripts/exp_gn_blocks.py,
// profiles/r06_z8_gn_blocks.txt, r06_z9_gn_blocks_n.txt) says FEWER blocks at every batch size: 3 slices per sample (the >= 2048 rule of rounds 3 - 6) -> 1 at 896 frames =
// 47.4 -> 40.7 ms per step over the stem's 54 layers; at 64 frames 12.1 -> 5.5 ... 6.3 ms.  Rule: 512 blocks for samples of >= 1 MiB (they stream long enough to want two
// rounds of the chip), 256 = one per CU otherwise.
static int gn_split(int nb, int HW, int C, int threads, bool bwd = false) {
    int target = !bwd ? 2048 : ((int64_t)HW * C >= 512 * 1024 ? 512 : 256);
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_GN_BLOCKS")) target = atoi(e);       // scripts/exp_gn_blocks.py: blocks per launch of the two-launch entries
#endif
    int split = (target + nb - 1) / nb;
    const int max_split = (HW * (C / 8) + threads * 4 - 1) / (threads * 4);
    if (split > max_split) split = max_split;
    if (split > 64) split = 64;
    if (split < 1) split = 1;
    return split;
}

extern "C" int merlot_groupnorm_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* stats,
                                    int N, int H, int W, int C, int G, float eps, int relu, merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_groupnorm_fwd");
    MERLOT_CHECK(gamma && beta && y && stats && G > 0 && C % G == 0 && C % 8 == 0 && C <= 2048, MERLOT_ESHAPE,
                 "merlot_groupnorm_fwd: C must be a multiple of 8 and of G");
    const int HW = H * W;
    const int threads = gn_block_threads(C);
    const int64_t per = (int64_t)HW * C;                   // elements of one sample
    const int nb_max = gn_group_samples(N, per * 2, false);
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * 2 * (size_t)N * G, (hipStream_t)stream);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    for (int n0 = 0; n0 < N; n0 += nb_max) {
        const int nb = N - n0 < nb_max ? N - n0 : nb_max;
        const int split = gn_split(nb, HW, C, threads), ppb = (HW + split - 1) / split;
        const bf16* xg = (const bf16*)x + n0 * per;
        float* sg = stats + (int64_t)n0 * 2 * G;
        hipLaunchKernelGGL(gn_stats_kernel, dim3(nb, split), dim3(threads), sizeof(float) * 2 * G, (hipStream_t)stream, xg, sg, HW, C, G, ppb);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(((int64_t)nb * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, sg, (int64_t)nb * G,
                           1.0f / ((float)HW * (float)(C / G)), eps);
        hipLaunchKernelGGL(gn_apply_kernel, dim3(nb, split), dim3(threads), 0, (hipStream_t)stream, xg, sg, gamma, beta,
                           res ? (const bf16*)res + n0 * per : nullptr, (bf16*)y + n0 * per, HW, C, G, relu, ppb);
    }
    return merlot_launch_status("merlot_groupnorm_fwd");
}

extern "C" int merlot_groupnorm_bwd(const void* dy, const void* y, const void* x, const float* stats, const float* gamma,
                                    const float* beta, float* dgamma, float* dbeta, float* gsum, void* dx, void* dres, int N, int H,
                                    int W, int C, int G, float eps, int relu, merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_groupnorm_bwd");
    MERLOT_CHECK(dy && stats && gamma && dgamma && dbeta && gsum && dx && (!relu || y || beta) && G > 0 && C % G == 0 && C % 8 == 0 &&
                     C <= 2048, MERLOT_ESHAPE, "merlot_groupnorm_bwd: bad arguments (relu needs y, or beta to recompute the mask from x)");
    const int HW = H * W;
    const int threads = gn_block_threads(C);
    const int64_t per = (int64_t)HW * C;
    const int nb_max = gn_group_samples(N, per * 2 * (y ? 3 : 2), true);      // both passes read x | dy [| y]
    hipError_t e = hipMemsetAsync(gsum, 0, sizeof(float) * 2 * (size_t)N * G, (hipStream_t)stream);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    for (int n0 = 0; n0 < N; n0 += nb_max) {
        const int nb = N - n0 < nb_max ? N - n0 : nb_max;
        const int split = gn_split(nb, HW, C, threads, true), ppb = (HW + split - 1) / split;
        int split2 = split;                                // the apply pass has no trailing atomics: its own slicing (profiles/r06_z10_gn_blocks_apply.txt)
#ifdef MERLOT_EXPERIMENTS
        if (const char* e2 = getenv("MERLOT_GN_BLOCKS_APPLY")) {
            split2 = (atoi(e2) + nb - 1) / nb;
            const int mx = (HW * (C / 8) + threads * 4 - 1) / (threads * 4);
            split2 = split2 > mx ? mx : split2;
            split2 = split2 > 64 ? 64 : (split2 < 1 ? 1 : split2);
        }
#endif
        const int ppb2 = (HW + split2 - 1) / split2;
        const bf16 *dyg = (const bf16*)dy + n0 * per, *yg = y ? (const bf16*)y + n0 * per : nullptr, *xg = (const bf16*)x + n0 * per;
        const float* sg = stats + (int64_t)n0 * 2 * G;
        float* gg = gsum + (int64_t)n0 * 2 * G;
#define GN_BWD_LAUNCH(U)                                                                                                                             \
    hipLaunchKernelGGL(gn_bwd_stats_kernel<U>, dim3(nb, split), dim3(threads), sizeof(float) * 2 * C, (hipStream_t)stream, dyg, yg, xg, sg, gamma,     \
                       beta, dgamma, dbeta, gg, HW, C, G, eps, relu, ppb);                                                                             \
    hipLaunchKernelGGL(gn_bwd_apply_kernel<U>, dim3(nb, split2), dim3(threads), 0, (hipStream_t)stream, dyg, yg, xg, sg, gg, gamma, beta,              \
                       (bf16*)dx + n0 * per, dres ? (bf16*)dres + n0 * per : nullptr, HW, C, G, relu, ppb2)
        switch (gn_bwd_unroll()) {
            case 4: GN_BWD_LAUNCH(4); break;
            case 2: GN_BWD_LAUNCH(2); break;
            default: GN_BWD_LAUNCH(1); break;
        }
#undef GN_BWD_LAUNCH
    }
    return merlot_launch_status("merlot_groupnorm_bwd");
}

// One launch per direction (ABI v10; the backward's slots ABI v11): see gn_fwd_fused_kernel.  ws: merlot_groupnorm_fused_workspace_bytes(N, C, G) bytes forward,
// merlot_groupnorm_bwd_fused_workspace_bytes(N, H, W, C, G) backward; caller-owned, any content (control words zeroed here, on the stream); stats / gsum as in the two-launch entries.
extern "C" int64_t merlot_groupnorm_fused_workspace_bytes(int N, int C, int G) {
    if (N <= 0 || C <= 0 || G <= 0) return 0;
    return 4 * (16 + gn_ws_arrive_words(N) + (int64_t)N * 2 * G);
}

// Slices per sample the one-launch entries accept, and why there is a limit at all.  The wait of these kernels terminates if a free workgroup slot ANYWHERE on the chip gets the
// next workgroup.  Measured (profiles/r06_z7_gn_slices.txt, r06_z6_gn_stress.txt): a kernel stops returning once a sample has more slices than the kernel has resident
// workgroups PER XCD (backward, 2 per CU = 64 per XCD: 65 slices run, 70 hang; forward holding 8 positions, 3 per CU = 96: 90 run, 100 stalls for 21 s, 132 hang).  That is what
// in-order round-robin dispatch with head-of-line blocking does: workgroups go to XCD (id mod 8) in id order, and a workgroup whose XCD is full holds up every later one.  The
// slots a finished sample frees on one XCD are refilled in a burst, the burst claims consecutive items, so samples become XCD-local; a sample with more slices than the XCD has
// slots fills it with waiters, the next workgroup in line is (within 8 ids) one for that XCD, and nothing starts any more -- the other XCDs drain and idle.  (Stalls that
// resolve do so after ~20 s: something outside the kernel re-dispatches the waves.)  With S slices <= the per-XCD capacity C the same argument gives termination: a blocked
// dispatcher means a full XCD; if the oldest unfinished sample still had unclaimed slices every resident workgroup would hold one of ITS slices, so C <= claimed < S.
// The model predicts that with item = workgroup id (no claims) the round-robin itself spreads a sample over the eight XCDs and the limit becomes 8 C: measured with
// MERLOT_GN_STATIC=1 in the experiments build (profiles/r06_z13_gn_static.txt) -- the 64-per-XCD backward runs at 70 / 132 / 256 / 480 slices and hangs at 520 and 640.
// Hence: both kernels are compiled for >= 2 workgroups per CU (__launch_bounds__(256, 2): C >= 64; tests/test_loop_isa.py checks the register counts) and both entries
// refuse more than 40 slices.  Every as-shipped forward shape is <= 33; the forward default ran 3 900 individually timed calls without an outlier.
constexpr int GN_FUSED_MAX_SLICES = 40;
static int gn_static_items() {
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_GN_STATIC")) return atoi(e);
#endif
    return 0;
}
static int gn_fused_max_slices() {
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_GN_MAX_SLICES")) return atoi(e);       // scripts/exp_gn_slices.py: where the stall begins
#endif
    return GN_FUSED_MAX_SLICES;
}
// the backward's workspace: control words + one slot of (C + G) float pairs per (sample, slice); the slicing is the launch's own (GN_BWD_ITER positions per thread)
constexpr int GN_BWD_ITER = 8;
extern "C" int64_t merlot_groupnorm_bwd_fused_workspace_bytes(int N, int H, int W, int C, int G) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || G <= 0 || C % 8 != 0) return 0;
    const int threads = gn_block_threads(C), pstep = threads / (C / 8);
    const int64_t split = ((int64_t)H * W + GN_BWD_ITER * pstep - 1) / (GN_BWD_ITER * pstep);
    return 4 * (16 + gn_ws_arrive_words(N) + (int64_t)N * split * 2 * (C + G));
}

extern "C" int merlot_groupnorm_fwd_fused(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* stats,
                                          int N, int H, int W, int C, int G, float eps, int relu, void* ws, int64_t ws_bytes,
                                          merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_groupnorm_fwd_fused");
    MERLOT_CHECK(gamma && beta && y && stats && G > 0 && C % G == 0 && C % 8 == 0 && C <= 2048 && G <= 1024, MERLOT_ESHAPE,
                 "merlot_groupnorm_fwd_fused: C must be a multiple of 8 and of G, C <= 2048");
    MERLOT_CHECK(ws && ws_bytes >= merlot_groupnorm_fused_workspace_bytes(N, C, G) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, MERLOT_ESHAPE,
                 "merlot_groupnorm_fwd_fused: workspace of merlot_groupnorm_fused_workspace_bytes(N, C, G) = %lld bytes required (got %lld)",
                 (long long)merlot_groupnorm_fused_workspace_bytes(N, C, G), (long long)ws_bytes);
    MERLOT_CHECK(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) == 0, MERLOT_EALIGN,
                 "merlot_groupnorm_fwd_fused: operands must be 16-B aligned");
    const int HW = H * W;
    const int threads = gn_block_threads(C);
    const int pstep = threads / (C / 8);
    const bool hold = res && (HW + 8 * pstep - 1) / (8 * pstep) <= gn_fused_max_slices();      // (see the kernel: the residual held across the wait, or read behind it)
    const int iter = hold ? 8 : 16;
    const int split = (HW + iter * pstep - 1) / (iter * pstep);
    MERLOT_CHECK((int64_t)N * split < (1LL << 31), MERLOT_ESHAPE, "merlot_groupnorm_fwd_fused: too many slices");
    MERLOT_CHECK(split <= gn_fused_max_slices(), MERLOT_ESHAPE, "merlot_groupnorm_fwd_fused: %d slices per sample (H * W = %d positions of %d channels); measured up to %d -- use merlot_groupnorm_fwd",
                 split, HW, C, GN_FUSED_MAX_SLICES);
    hipError_t e = hipMemsetAsync(ws, 0, 4 * (16 + gn_ws_arrive_words(N) + (size_t)N * 2 * G), (hipStream_t)stream);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    const float inv_cnt = 1.0f / ((float)HW * (float)(C / G));
    const size_t lds = sizeof(float) * 4 * G;
    if (res && hold)
        hipLaunchKernelGGL((gn_fwd_fused_kernel<8, true, true>), dim3(N * split), dim3(threads), lds, (hipStream_t)stream, (const bf16*)x, gamma, beta,
                           (const bf16*)res, (bf16*)y, stats, (unsigned*)ws, N, HW, C, G, relu, split, inv_cnt, eps, gn_static_items());
    else if (res)
        hipLaunchKernelGGL((gn_fwd_fused_kernel<16, true>), dim3(N * split), dim3(threads), lds, (hipStream_t)stream, (const bf16*)x, gamma, beta,
                           (const bf16*)res, (bf16*)y, stats, (unsigned*)ws, N, HW, C, G, relu, split, inv_cnt, eps, gn_static_items());
    else
        hipLaunchKernelGGL((gn_fwd_fused_kernel<16, false>), dim3(N * split), dim3(threads), lds, (hipStream_t)stream, (const bf16*)x, gamma, beta,
                           (const bf16*)nullptr, (bf16*)y, stats, (unsigned*)ws, N, HW, C, G, relu, split, inv_cnt, eps, gn_static_items());
    return merlot_launch_status("merlot_groupnorm_fwd_fused");
}

extern "C" int merlot_groupnorm_bwd_fused(const void* dy, const void* y, const void* x, const float* stats, const float* gamma,
                                          const float* beta, float* dgamma, float* dbeta, float* gsum, void* dx, void* dres, int N, int H,
                                          int W, int C, int G, float eps, int relu, void* ws, int64_t ws_bytes, merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_groupnorm_bwd_fused");
    MERLOT_CHECK(dy && stats && gamma && dgamma && dbeta && gsum && dx && (!relu || y || beta) && G > 0 && C % G == 0 && C % 8 == 0 &&
                     C <= 2048 && G <= 1024, MERLOT_ESHAPE, "merlot_groupnorm_bwd_fused: bad arguments (relu needs y, or beta to recompute the mask from x)");
    MERLOT_CHECK(ws && ws_bytes >= merlot_groupnorm_bwd_fused_workspace_bytes(N, H, W, C, G) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, MERLOT_ESHAPE,
                 "merlot_groupnorm_bwd_fused: workspace of merlot_groupnorm_bwd_fused_workspace_bytes(N, H, W, C, G) = %lld bytes required (got %lld)",
                 (long long)merlot_groupnorm_bwd_fused_workspace_bytes(N, H, W, C, G), (long long)ws_bytes);
    MERLOT_CHECK(((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dres)) & 15) == 0,
                 MERLOT_EALIGN, "merlot_groupnorm_bwd_fused: operands must be 16-B aligned");
    const int HW = H * W;
    const int threads = gn_block_threads(C);
    const int pstep = threads / (C / 8);
    constexpr int ITER = GN_BWD_ITER;
    const int split = (HW + ITER * pstep - 1) / (ITER * pstep);
    MERLOT_CHECK((int64_t)N * split < (1LL << 31), MERLOT_ESHAPE, "merlot_groupnorm_bwd_fused: too many slices");
    // 66 slices per sample: calls that never return (profiles/r06_z6_gn_stress.txt; cause not found -- every shape up to 33 slices ran, thousands of calls forward)
    MERLOT_CHECK(split <= gn_fused_max_slices(), MERLOT_ESHAPE, "merlot_groupnorm_bwd_fused: %d slices per sample (H * W = %d positions of %d channels); this entry stalls above %d -- use merlot_groupnorm_bwd",
                 split, HW, C, GN_FUSED_MAX_SLICES);
    hipError_t e = hipMemsetAsync(ws, 0, 4 * (16 + gn_ws_arrive_words(N)), (hipStream_t)stream);      // the control words; every slot and gsum entry is written before it is read
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    const size_t lds = sizeof(float) * 2 * (C + G);
    const bool has_y = relu && y != nullptr;
    if (has_y)
        hipLaunchKernelGGL((gn_bwd_fused_kernel<ITER, true>), dim3(N * split), dim3(threads), lds, (hipStream_t)stream, (const bf16*)dy, (const bf16*)y,
                           (const bf16*)x, stats, gamma, beta, dgamma, dbeta, gsum, (bf16*)dx, (bf16*)dres, (unsigned*)ws, N, HW, C, G, relu, split, gn_static_items());
    else
        hipLaunchKernelGGL((gn_bwd_fused_kernel<ITER, false>), dim3(N * split), dim3(threads), lds, (hipStream_t)stream, (const bf16*)dy,
                           (const bf16*)nullptr, (const bf16*)x, stats, gamma, beta, dgamma, dbeta, gsum, (bf16*)dx, (bf16*)dres, (unsigned*)ws, N,
                           HW, C, G, relu, split, gn_static_items());
    return merlot_launch_status("merlot_groupnorm_bwd_fused");
}

extern "C" int merlot_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_avgpool2_fwd");
    MERLOT_CHECK(y && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, MERLOT_ESHAPE, "merlot_avgpool2_fwd: even H, W and C % 8 == 0 required");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y,
                       (int64_t)N, H, W, C);
    return merlot_launch_status("merlot_avgpool2_fwd");
}

extern "C" int merlot_avgpool2_bwd(const void* x, void* dx, int N, int H, int W, int C, merlot_stream_t stream) {
    CONV_CHECK_GEOM("merlot_avgpool2_bwd");                 // x = dy [N, H/2, W/2, C]; dx [N, H, W, C]
    MERLOT_CHECK(dx && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, MERLOT_ESHAPE, "merlot_avgpool2_bwd: even H, W and C % 8 == 0 required");
    const int64_t total = (int64_t)N * H * W * (C / 8);
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)dx,
                       (int64_t)N, H, W, C);
    return merlot_launch_status("merlot_avgpool2_bwd");
}

// ------------------------------------------------------------------------------------------------
// Weight standardisation of the ResNet-hybrid stem's kernels (utils/vision_transformer.py:52-56): per OUTPUT channel, over
// (kh, kw, ci): khat = (k - mean) * rsqrt(var + 1e-5), population variance.  k is the fp32 master in HWIO order = [K, Co]
// row-major (K = kh*kw*ci), so a channel is a strided COLUMN: one workgroup per 16 channels x 16 K-lanes (below).  Outputs: khat fp32 [K, Co] + rstd [Co] (for the backward), the
// NT operand wb bf16 [Co, Kp] and the dgrad operand wbT bf16 [Kp, Cop] (their padding is zeroed by the caller once).
// ------------------------------------------------------------------------------------------------
namespace {
// block = 16 channels x 16 K-lanes: thread (kl = tid >> 4, cl = tid & 15) walks rows kl, kl + 16, ... of its channel with 8
// independent loads in flight (the tensors are a few MB: this is a latency problem, not a bandwidth one); LDS reduction over
// the K-lanes.
constexpr int WS_C = 16, WS_K = 16;
__device__ __forceinline__ float ws_reduce(float v, float (&red)[WS_K][WS_C], int kl, int cl) {
    __syncthreads();
    red[kl][cl] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < WS_K; ++i) t += red[i][cl];
    return t;
}
// wdg (3x3 kernels only, else null): the operand of the layer's INPUT gradient as an implicit convolution of dY (csrc/conv_gemm.hip):
// wdg[ci][(2-ky, 2-kx, co)] = khat[(ky, kx, ci)][co], bf16 [Cin, 9 Co]
__device__ __forceinline__ void weight_std_fwd_body(const float* __restrict__ k, int K, int Co, float* __restrict__ khat,
                                                    float* __restrict__ rstd_out, bf16* __restrict__ wb, int Kp,
                                                    bf16* __restrict__ wbT, int Cop, bf16* __restrict__ wdg, int Cin, int blk,
                                                    float (&red)[WS_K][WS_C]) {
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blk * WS_C + cl;
    const bool live = c < Co;
    const int cc = live ? c : Co - 1;
    float s = 0.f;
#pragma unroll 8
    for (int r = kl; r < K; r += WS_K) s += k[(int64_t)r * Co + cc];
    const float mean = ws_reduce(s, red, kl, cl) / (float)K;
    float q = 0.f;
#pragma unroll 8
    for (int r = kl; r < K; r += WS_K) {
        const float d = k[(int64_t)r * Co + cc] - mean;
        q += d * d;
    }
    const float var = ws_reduce(q, red, kl, cl) / (float)K;
    const float rs = rsqrtf(var + 1e-5f);
    if (!live) return;
    if (kl == 0) rstd_out[c] = rs;
#pragma unroll 8
    for (int r = kl; r < K; r += WS_K) {
        const float v = (k[(int64_t)r * Co + c] - mean) * rs;
        khat[(int64_t)r * Co + c] = v;
        wb[(int64_t)c * Kp + r] = (bf16)v;
        wbT[(int64_t)r * Cop + c] = (bf16)v;
        if (wdg) {
            const int tap = r / Cin, ci = r - tap * Cin;
            wdg[(int64_t)ci * (9 * Co) + (8 - tap) * Co + c] = (bf16)v;
        }
    }
}
__global__ __launch_bounds__(256) void weight_std_fwd_kernel(const float* __restrict__ k, int K, int Co, float* __restrict__ khat,
                                                             float* __restrict__ rstd_out, bf16* __restrict__ wb, int Kp,
                                                             bf16* __restrict__ wbT, int Cop) {
    __shared__ float red[WS_K][WS_C];
    weight_std_fwd_body(k, K, Co, khat, rstd_out, wb, Kp, wbT, Cop, nullptr, 1, blockIdx.x, red);
}
// every kernel of the stem in ONE launch (52 kernels of a few KB to a few MB: 52 launches of 18 us each otherwise).  Job j = 12 int64:
// {k offset, K, Co, khat offset, rstd offset, wb offset, Kp, wbT offset, Cop, wdg offset (< 0: none), Cin, first block}; offsets in
// elements from the respective base; a block finds its job by binary search on the first-block column.
constexpr int WS_JOB = 12;
__device__ __forceinline__ const int64_t* ws_find_job(const int64_t* __restrict__ jobs, int njobs, int stride, int first_col, int64_t b) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                    // last job whose first block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid * stride + first_col] <= b) lo = mid; else hi = mid - 1;
    }
    return jobs + lo * stride;
}
__global__ __launch_bounds__(256) void weight_std_fwd_batched_kernel(const float* __restrict__ k_base, const int64_t* __restrict__ jobs,
                                                                     int njobs, float* __restrict__ khat_base,
                                                                     float* __restrict__ rstd_base, bf16* __restrict__ wb_base,
                                                                     bf16* __restrict__ wbT_base, bf16* __restrict__ wdg_base) {
    __shared__ float red[WS_K][WS_C];
    const int64_t* j = ws_find_job(jobs, njobs, WS_JOB, 11, blockIdx.x);
    weight_std_fwd_body(k_base + j[0], (int)j[1], (int)j[2], khat_base + j[3], rstd_base + j[4], wb_base + j[5], (int)j[6],
                        wbT_base + j[7], (int)j[8], j[9] >= 0 ? wdg_base + j[9] : nullptr, (int)j[10], (int)(blockIdx.x - j[11]), red);
}

// dk = rstd * (dkhat - mean_K(dkhat) - khat * mean_K(dkhat * khat)), accumulated into the gradient arena.
// dkhat is read TRANSPOSED from the wgrad GEMM's output [Co(+pad), ld] (row = channel, K contiguous).
__device__ __forceinline__ void weight_std_bwd_body(const float* __restrict__ dkt, int64_t ld, const float* __restrict__ khat,
                                                    const float* __restrict__ rstd, int K, int Co, float* __restrict__ gk, int blk,
                                                    float (&red)[WS_K][WS_C]) {
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blk * WS_C + cl;
    const bool live = c < Co;
    const int cc = live ? c : Co - 1;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int r = kl; r < K; r += WS_K) {
        const float d = dkt[(int64_t)cc * ld + r];
        s1 += d;
        s2 += d * khat[(int64_t)r * Co + cc];
    }
    const float m1 = ws_reduce(s1, red, kl, cl) / (float)K;
    const float m2 = ws_reduce(s2, red, kl, cl) / (float)K;
    if (!live) return;
    const float rs = rstd[c];
#pragma unroll 8
    for (int r = kl; r < K; r += WS_K)
        gk[(int64_t)r * Co + c] += rs * (dkt[(int64_t)c * ld + r] - m1 - khat[(int64_t)r * Co + c] * m2);
}
__global__ __launch_bounds__(256) void weight_std_bwd_kernel(const float* __restrict__ dkt, int64_t ld, const float* __restrict__ khat,
                                                             const float* __restrict__ rstd, int K, int Co, float* __restrict__ gk) {
    __shared__ float red[WS_K][WS_C];
    weight_std_bwd_body(dkt, ld, khat, rstd, K, Co, gk, blockIdx.x, red);
}
// job j = 8 int64: {dk offset, ld, khat offset, rstd offset, K, Co, gk offset, first block}
constexpr int WS_BJOB = 8;
__global__ __launch_bounds__(256) void weight_std_bwd_batched_kernel(const float* __restrict__ dk_base, const int64_t* __restrict__ jobs,
                                                                     int njobs, const float* __restrict__ khat_base,
                                                                     const float* __restrict__ rstd_base, float* __restrict__ gk_base) {
    __shared__ float red[WS_K][WS_C];
    const int64_t* j = ws_find_job(jobs, njobs, WS_BJOB, 7, blockIdx.x);
    weight_std_bwd_body(dk_base + j[0], j[1], khat_base + j[2], rstd_base + j[3], (int)j[4], (int)j[5], gk_base + j[6],
                        (int)(blockIdx.x - j[7]), red);
}
}  // namespace

extern "C" int merlot_weight_std_fwd(const float* k, int K, int Co, float* khat, float* rstd, void* wb, int Kp, void* wbT, int Cop,
                                     merlot_stream_t stream) {
    MERLOT_CHECK(k && khat && rstd && wb && wbT && K > 0 && Co > 0 && Kp >= K && Cop >= Co, MERLOT_ESHAPE, "merlot_weight_std_fwd: bad arguments");
    hipLaunchKernelGGL(weight_std_fwd_kernel, dim3((Co + WS_C - 1) / WS_C), dim3(256), 0, (hipStream_t)stream, k, K, Co, khat, rstd, (bf16*)wb, Kp,
                       (bf16*)wbT, Cop);
    return merlot_launch_status("merlot_weight_std_fwd");
}

extern "C" int merlot_weight_std_bwd(const float* dkhat_t, int64_t ld, const float* khat, const float* rstd, int K, int Co, float* gk,
                                     merlot_stream_t stream) {
    MERLOT_CHECK(dkhat_t && khat && rstd && gk && K > 0 && Co > 0 && ld >= K, MERLOT_ESHAPE, "merlot_weight_std_bwd: bad arguments");
    hipLaunchKernelGGL(weight_std_bwd_kernel, dim3((Co + WS_C - 1) / WS_C), dim3(256), 0, (hipStream_t)stream, dkhat_t, ld, khat, rstd, K, Co, gk);
    return merlot_launch_status("merlot_weight_std_bwd");
}

extern "C" int merlot_weight_std_fwd_batched(const float* k_base, const void* jobs, int njobs, int64_t total_blocks, float* khat_base,
                                             float* rstd_base, void* wb_base, void* wbT_base, void* wdg_base, merlot_stream_t stream) {
    MERLOT_CHECK(k_base && jobs && njobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31) && khat_base && rstd_base && wb_base && wbT_base,
                 MERLOT_ESHAPE, "merlot_weight_std_fwd_batched: bad arguments");
    hipLaunchKernelGGL(weight_std_fwd_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, k_base,
                       (const int64_t*)jobs, njobs, khat_base, rstd_base, (bf16*)wb_base, (bf16*)wbT_base, (bf16*)wdg_base);
    return merlot_launch_status("merlot_weight_std_fwd_batched");
}

extern "C" int merlot_weight_std_bwd_batched(const float* dk_base, const void* jobs, int njobs, int64_t total_blocks, const float* khat_base,
                                             const float* rstd_base, float* gk_base, merlot_stream_t stream) {
    MERLOT_CHECK(dk_base && jobs && njobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31) && khat_base && rstd_base && gk_base,
                 MERLOT_ESHAPE, "merlot_weight_std_bwd_batched: bad arguments");
    hipLaunchKernelGGL(weight_std_bwd_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, dk_base,
                       (const int64_t*)jobs, njobs, khat_base, rstd_base, gk_base);
    return merlot_launch_status("merlot_weight_std_bwd_batched");
}
