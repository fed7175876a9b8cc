// HBM-bound helpers of the MERLOT hot path: parameter casts, gathers / scatter-adds (embedding sums),
// the CLS + 2x2 average pool, bias gradients, row softmax-CE, l2-normalise, GELU, fused AdamW.
// All are vectorised (8-16 B per lane), grid-strided, and launched with >> 256 workgroups.
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int per_block, int cap = 4096) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- casts ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        reinterpret_cast<bf16x4*>(dst)[i] = o;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (bf16)src[i];
}

// dst[C][R] = src[R][C]^T, 64x64 tiles through LDS (padded), coalesced both ways.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                             int64_t R, int64_t C, int64_t ld_dst) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[c * ld_dst + r] = (bf16)tile[tx][i];
    }
}

// many transposes in ONE launch: job j = {src offset, dst offset, R, C, ld_dst, first tile}; a block finds its job in
// the (short) table by binary search on the first-tile column.  Used for the per-step refresh of every Linear's
// transposed bf16 copy (~130 matrices: one launch instead of 130 launch latencies).
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const float* __restrict__ src_base, bf16* __restrict__ dst_base,
                                                                     const int64_t* __restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    int lo = 0, hi = njobs - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {                                    // last job whose first tile <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid * 6 + 5] <= b) lo = mid; else hi = mid - 1;
    }
    const int64_t* j = jobs + lo * 6;
    const float* src = src_base + j[0];
    bf16* dst = dst_base + j[1];
    const int64_t R = j[2], C = j[3], ld_dst = j[4];
    const int64_t t = b - j[5];
    const int64_t tiles_c = (C + 63) / 64;
    const int64_t r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[c * ld_dst + r] = (bf16)tile[tx][i];
    }
}

// ---- bias gradient: out[n] (+)= sum_t x[t][n] ------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, int64_t ld, float* __restrict__ out,
                                                     int64_t T, int64_t N, int64_t rows_per_block) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 256 + cg * 4;
    const int64_t t0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t t1 = min(T, t0 + rows_per_block);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (n + 3 < N) {
        for (int64_t t = t0 + rl; t < t1; t += 4) {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(x + t * ld + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (float)v[e];
        }
    } else {
        for (int64_t t = t0 + rl; t < t1; t += 4)
            for (int e = 0; e < 4; ++e)
                if (n + e < N) acc[e] += (float)x[t * ld + n + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cg * 4 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    const int64_t nn = (int64_t)blockIdx.x * 256 + c;
    if (nn < N) atomicAdd(out + nn, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

// ---- gather-add of up to three f32/bf16 tables ---------------------------------------------------------
template <typename TA>
__global__ __launch_bounds__(256) void gather_add4_kernel(const TA* __restrict__ a, const int32_t* __restrict__ ia,
                                                          const float* __restrict__ b, const int32_t* __restrict__ ib,
                                                          const float* __restrict__ c, const int32_t* __restrict__ ic,
                                                          const float* __restrict__ d, const int32_t* __restrict__ id,
                                                          float* __restrict__ out, int64_t rows, int H) {
    const int h4 = H >> 2;
    const int64_t total = rows * h4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / h4;
        const int col = (int)(i - r * h4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (a) {
            const int64_t k = ia ? ia[r] : r;
            if (k >= 0) {
                if (sizeof(TA) == 2) {
                    const bf16x4 t = *reinterpret_cast<const bf16x4*>((const bf16*)a + k * H + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)t[e];
                } else {
                    const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)a + k * H + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += t[e];
                }
            }
        }
        if (b) {
            const int64_t k = ib ? ib[r] : r;
            if (k >= 0) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(b + k * H + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += t[e];
            }
        }
        if (c) {
            const int64_t k = ic ? ic[r] : r;
            if (k >= 0) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(c + k * H + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += t[e];
            }
        }
        if (d) {
            const int64_t k = id ? id[r] : r;
            if (k >= 0) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(d + k * H + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += t[e];
            }
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
        *reinterpret_cast<f32x4*>(out + r * H + col) = o;
    }
}

// y = keep(seed, linear index) ? x / (1-p) : 0 ; same counter-based mask as the GEMM residual epilogue
__global__ __launch_bounds__(256) void dropout_apply_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n,
                                                            uint32_t thresh, float scale, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (float)x[i];
        y[i] = (bf16)((thresh == 0 || dropout_keep(seed, (uint64_t)i, thresh)) ? v * scale : 0.f);
    }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src,
                                                               const int32_t* __restrict__ idx, float* __restrict__ table,
                                                               int64_t rows, int H) {
    const int64_t total = rows * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / H;
        const int col = (int)(i - r * H);
        const int64_t k = idx ? idx[r] : r;
        if (k >= 0) atomicAdd(table + k * H + col, src[i]);
    }
}

// table[sorted_idx[j], :] += src[perm[j], :] over j in SORTED index order (perm = a stable argsort of the row indices, made by the
// caller): a block owns a contiguous range of sorted positions, a thread 4 columns; rows of one index are summed in registers and
// written once -- with a plain read-modify-write when the index's run lies strictly inside the block's range (no other block can
// touch that table row), with atomics for the first and the last index of the range (their runs may continue in a neighbour).
// The word-embedding gradient of a step is 65 536 rows of 768 floats onto ~20 000 distinct tokens: 50 M fp32 atomics (708 us) before.
constexpr int SCS_ROWS = 32;
__global__ __launch_bounds__(256) void scatter_add_sorted_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm,
                                                                 const int32_t* __restrict__ sidx, float* __restrict__ table,
                                                                 int64_t rows, int H) {
    const int64_t p0 = (int64_t)blockIdx.x * SCS_ROWS, p1 = p0 + SCS_ROWS < rows ? p0 + SCS_ROWS : rows;
    const int first = sidx[p0], last = sidx[p1 - 1];
    const bool first_shared = p0 > 0 && sidx[p0 - 1] == first, last_shared = p1 < rows && sidx[p1] == last;
    for (int col = threadIdx.x * 4; col < H; col += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int cur = first;
        for (int64_t j = p0; j <= p1; ++j) {
            const int k = j < p1 ? sidx[j] : -2147483647;               // sentinel: flush the last run
            if (k != cur) {
                if (cur >= 0) {
                    float* t = table + (int64_t)cur * H + col;
                    if ((cur == first && first_shared) || (cur == last && last_shared)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(t + e, acc[e]);
                    } else {
                        f32x4 o = *reinterpret_cast<const f32x4*>(t);
                        o += acc;
                        *reinterpret_cast<f32x4*>(t) = o;
                    }
                }
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                cur = k;
            }
            if (j < p1) acc += *reinterpret_cast<const f32x4*>(src + (int64_t)perm[j] * H + col);
        }
    }
}

// ---- CLS + 2x2 average pool ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cls_avgpool_fwd_kernel(const bf16* __restrict__ x, float* __restrict__ out,
                                                              int n_img, int h1, int w1, int cls_skip, int pool, int H) {
    const int h2 = h1 / pool, w2 = w1 / pool;
    const int vl = 1 + h2 * w2;
    const int S = cls_skip + h1 * w1;
    const int h4 = H >> 2;
    const int64_t total = (int64_t)n_img * vl * h4;
    const float inv = 1.0f / (pool * pool);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % h4) * 4;
        const int64_t rr = i / h4;
        const int t = (int)(rr % vl);
        const int64_t n = rr / vl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bf16* xb = x + n * S * (int64_t)H;
        if (t == 0) {
            const bf16x4 q = *reinterpret_cast<const bf16x4*>(xb + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)q[e];
        } else {
            const int ph = (t - 1) / w2, pw = (t - 1) % w2;
            for (int dy = 0; dy < pool; ++dy)
                for (int dx = 0; dx < pool; ++dx) {
                    const int s = cls_skip + (ph * pool + dy) * w1 + pw * pool + dx;
                    const bf16x4 q = *reinterpret_cast<const bf16x4*>(xb + (int64_t)s * H + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)q[e];
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= inv;
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
        *reinterpret_cast<f32x4*>(out + rr * H + col) = o;
    }
}

// dx[n, s, :] (bf16) for every s in [0, S): slot 0 <- dout[n,0]; other CLS slots <- 0 ; grid <- dout/pool^2
__global__ __launch_bounds__(256) void cls_avgpool_bwd_kernel(const float* __restrict__ dout, bf16* __restrict__ dx,
                                                              int n_img, int h1, int w1, int cls_skip, int pool, int H) {
    const int h2 = h1 / pool, w2 = w1 / pool;
    const int vl = 1 + h2 * w2;
    const int S = cls_skip + h1 * w1;
    const int h4 = H >> 2;
    const int64_t total = (int64_t)n_img * S * h4;
    const float inv = 1.0f / (pool * pool);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % h4) * 4;
        const int64_t rr = i / h4;
        const int s = (int)(rr % S);
        const int64_t n = rr / S;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (s == 0) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(dout + n * vl * (int64_t)H + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = q[e];
        } else if (s >= cls_skip) {
            const int g = s - cls_skip;
            const int ph = (g / w1) / pool, pw = (g % w1) / pool;
            if (ph < h2 && pw < w2) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(dout + (n * vl + 1 + ph * w2 + pw) * (int64_t)H + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = q[e] * inv;
            }
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        *reinterpret_cast<bf16x4*>(dx + rr * H + col) = o;
    }
}

// ---- row softmax cross-entropy ---------------------------------------------------------------------
// one workgroup (256 threads) per row; three passes over the row (max+argmax, sum-exp, gradient).
template <typename TDL>
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, int64_t ld,
                                                         const int32_t* __restrict__ labels, float* __restrict__ loss,
                                                         int32_t* __restrict__ argmax, const float* __restrict__ rowscale,
                                                         TDL* __restrict__ dl, int64_t ld_dl, int C) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    __shared__ float ssum[4];
    const int64_t row = blockIdx.x;
    const float* lr = logits + row * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = lr[c];
        if (v > mx) { mx = v; mi = c; }   // strided ascending per thread => first max per thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    if (lane == 0) { sval[wave] = mx; sidx[wave] = mi; }
    __syncthreads();
    mx = sval[0]; mi = sidx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sval[w] > mx || (sval[w] == mx && sidx[w] < mi)) { mx = sval[w]; mi = sidx[w]; }
    float sum = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) sum += __expf(lr[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) ssum[wave] = sum;
    __syncthreads();
    sum = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    const float lse = mx + __logf(sum);
    const int lab = labels[row];
    if (threadIdx.x == 0) {
        loss[row] = lse - lr[lab];
        if (argmax) argmax[row] = mi;
    }
    if (dl) {
        const float rs = rowscale ? rowscale[row] : 1.0f;
        TDL* dr = dl + row * ld_dl;
        for (int64_t c = threadIdx.x; c < ld_dl; c += 256) {
            float g = 0.f;
            if (c < C) g = rs * (__expf(lr[c] - lse) - (c == lab ? 1.0f : 0.0f));
            dr[c] = (TDL)g;
        }
    }
}

// Wide rows (the vocabulary: C = 50 370 fp32 logits = 197 KB): one 1024-thread workgroup per row keeps the WHOLE row in registers
// (up to 16 x 16 B per thread: C <= 65 536) -- the logits are read from memory once instead of three times; 16-B loads, 8 / 16-B
// stores.  Same arithmetic per element as softmax_ce_kernel (first maximum wins the argmax; exp(x - max) summed; gradient from
// exp(x - lse)); the sum runs in a different order, so loss and gradient agree with it to fp32 rounding, not bit for bit.
template <typename TDL>
__global__ __launch_bounds__(1024) void softmax_ce_wide_kernel(const float* __restrict__ logits, int64_t ld,
                                                               const int32_t* __restrict__ labels, float* __restrict__ loss,
                                                               int32_t* __restrict__ argmax, const float* __restrict__ rowscale,
                                                               TDL* __restrict__ dl, int64_t ld_dl, int C) {
    constexpr int MAXV = 16;
    __shared__ float sval[16];
    __shared__ int sidx[16];
    __shared__ float ssum[16];
    const int64_t row = blockIdx.x;
    const float* lr = logits + row * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = (C + 3) >> 2;                          // 4-column groups that hold logits
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int g = i * 1024 + tid;
        if (g < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(lr + 4 * (int64_t)g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e >= C) v[i][e] = -INFINITY;
        } else {
            v[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    float mx = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[i][e] > mx) { mx = v[i][e]; mi = 4 * (i * 1024 + tid) + e; }     // ascending per thread => first max per thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    if (lane == 0) { sval[wave] = mx; sidx[wave] = mi; }
    __syncthreads();
    mx = sval[0]; mi = sidx[0];
#pragma unroll
    for (int w = 1; w < 16; ++w)
        if (sval[w] > mx || (sval[w] == mx && sidx[w] < mi)) { mx = sval[w]; mi = sidx[w]; }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += __expf(v[i][e] - mx);          // exp(-inf) = 0 beyond C
    sum = wave_sum(sum);
    if (lane == 0) ssum[wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) sum += ssum[w];
    const float lse = mx + __logf(sum);
    const int lab = labels[row];
    if (tid == 0) {
        loss[row] = lse - lr[lab];
        if (argmax) argmax[row] = mi;
    }
    if (dl) {
        const float rs = rowscale ? rowscale[row] : 1.0f;
        TDL* dr = dl + row * ld_dl;
        const int no = (int)(ld_dl >> 2);                 // 4-column groups of the output row (columns C .. ld_dl: zeros)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int g = i * 1024 + tid;
            if (g < no) {
                TDL o4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * g + e;
                    o4[e] = (TDL)(c < C ? rs * (__expf(v[i][e] - lse) - (c == lab ? 1.0f : 0.0f)) : 0.f);
                }
                *reinterpret_cast<decltype(o4)*>(dr + 4 * (int64_t)g) = o4;
            }
        }
    }
}

// ---- l2 normalise ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv_norm, int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ss = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float v = x[row * H + c];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float inv = rsqrtf(fmaxf(ss, 1e-12f));
    for (int c = lane; c < H; c += 64) y[row * H + c] = x[row * H + c] * inv;
    if (lane == 0 && inv_norm) inv_norm[row] = inv;
}
// y = x*inv ; dx = inv * (dy - y * <dy, y>)   (clamp branch ignored: norm^2 >= 1e-12 in practice)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                         int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < H; c += 64) dot += dy[row * H + c] * y[row * H + c];
    dot = wave_sum(dot);
    const float inv = inv_norm[row];
    for (int c = lane; c < H; c += 64) dx[row * H + c] = inv * (dy[row * H + c] - y[row * H + c] * dot);
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = gelu_f(x[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * gelu_grad_f(x[i]);
}

// ---- AdamW (utils/optimization.py:267-288, 339-416) -----------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {  // round-to-nearest-even (tf.cast semantics)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
constexpr float MISSING_PRECISION = 1.00390625f;  // optimization.py:268

template <bool STATE_BF16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                    void* __restrict__ m_, void* __restrict__ v_, int64_t n, float lr,
                                                    float beta1, float beta2, float omb1, float omb2, float eps,
                                                    float wd_rate, float gscale, const uint8_t* __restrict__ wd_flags) {
#pragma clang fp contract(off)   // one IEEE op per reference op: the bf16 state encoding is sensitive to the last ulp
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = grad[i] * gscale;
        const float wd = (wd_flags == nullptr || wd_flags[i >> 6]) ? wd_rate : 0.f;
        float m, v;
        if (STATE_BF16) {
            m = bf16_bits_to_f32(((const uint16_t*)m_)[i]);
            const uint16_t sv = ((const uint16_t*)v_)[i];
            const float va = bf16_bits_to_f32(sv & 0x7fffu);
            // _decode_v: sign > 0 -> |v| ; otherwise (negative OR zero) -> |v| * 1.00390625   (:278-280)
            const bool positive = ((sv & 0x8000u) == 0) && ((sv & 0x7fffu) != 0);
            v = positive ? va : va * MISSING_PRECISION;
        } else {
            m = ((const float*)m_)[i];
            v = ((const float*)v_)[i];
        }
        const float g2 = g * g + 1e-30f;                         // :360
        const float nm = beta1 * m + omb1 * g;                   // :390  (1 - beta evaluated in double, as python does)
        const float nv = beta2 * v + omb2 * g2;                  // :391
        float upd = nm / (sqrtf(nv) + eps);                      // :393
        const float pw = param[i];
        if (wd > 0.f) upd += wd * pw;                            // :402-403
        param[i] = pw - lr * upd;                                // :405-407
        if (STATE_BF16) {
            ((uint16_t*)m_)[i] = f32_to_bf16_bits(nm);           // :410
            const uint16_t enc = f32_to_bf16_bits(nv);           // _encode_v :283-288
            const float ef = bf16_bits_to_f32(enc);
            const float err0 = fabsf(ef - nv);
            const float err1 = fabsf(ef * MISSING_PRECISION - nv);
            ((uint16_t*)v_)[i] = (err0 <= err1) ? enc : (uint16_t)(enc ^ 0x8000u);
        } else {
            ((float*)m_)[i] = nm;
            ((float*)v_)[i] = nv;
        }
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int merlot_cast_f32_bf16(const float* src, void* dst, int64_t n, merlot_stream_t stream) {
    MERLOT_CHECK(src && dst && n > 0, MERLOT_ESHAPE, "merlot_cast_f32_bf16: bad args");
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, STREAM, src, (bf16*)dst, n);
    return merlot_launch_status("merlot_cast_f32_bf16");
}

extern "C" int merlot_cast_transpose_f32_bf16(const float* src, void* dst, int64_t R, int64_t C, int64_t ld_dst,
                                              merlot_stream_t stream) {
    MERLOT_CHECK(src && dst && R > 0 && C > 0 && ld_dst >= R, MERLOT_ESHAPE, "merlot_cast_transpose_f32_bf16: bad args");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, STREAM, src, (bf16*)dst, R, C,
                       ld_dst);
    return merlot_launch_status("merlot_cast_transpose_f32_bf16");
}

extern "C" int merlot_cast_transpose_batched(const float* src_base, void* dst_base, const void* jobs, int njobs,
                                             int64_t total_tiles, merlot_stream_t stream) {
    MERLOT_CHECK(src_base && dst_base && jobs && njobs > 0 && total_tiles > 0 && total_tiles < (1ll << 31), MERLOT_ESHAPE,
                 "merlot_cast_transpose_batched: bad args");
    hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, STREAM, src_base, (bf16*)dst_base,
                       (const int64_t*)jobs, njobs);
    return merlot_launch_status("merlot_cast_transpose_batched");
}

extern "C" int merlot_colsum_bf16(const void* x, int64_t ld, float* out, int64_t T, int64_t N, int accumulate,
                                  merlot_stream_t stream) {
    MERLOT_CHECK(x && out && T > 0 && N > 0, MERLOT_ESHAPE, "merlot_colsum_bf16: bad args");
    MERLOT_CHECK(ld % 4 == 0, MERLOT_EALIGN, "merlot_colsum_bf16: ld must be a multiple of 4");
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, STREAM);
        MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "merlot_colsum_bf16: memset failed");
    }
    const int ncol = cdiv(N, 256);
    int nrb = 2048 / ncol;
    if (nrb < 1) nrb = 1;
    int64_t rpb = (T + nrb - 1) / nrb;
    if (rpb < 32) rpb = 32;
    nrb = cdiv(T, rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(ncol, nrb), dim3(256), 0, STREAM, (const bf16*)x, ld, out, T, N, rpb);
    return merlot_launch_status("merlot_colsum_bf16");
}

extern "C" int merlot_gather_add4(const void* a, int a_bf16, const int32_t* ia, const float* b, const int32_t* ib,
                                  const float* c, const int32_t* ic, const float* d, const int32_t* id, float* out,
                                  int64_t rows, int H, merlot_stream_t stream) {
    MERLOT_CHECK(out && rows > 0 && H > 0 && H % 4 == 0, MERLOT_ESHAPE, "merlot_gather_add4: bad args");
    const int g = grid_for(rows * (H / 4), 256);
    if (a_bf16)
        hipLaunchKernelGGL((gather_add4_kernel<bf16>), dim3(g), dim3(256), 0, STREAM, (const bf16*)a, ia, b, ib, c, ic, d,
                           id, out, rows, H);
    else
        hipLaunchKernelGGL((gather_add4_kernel<float>), dim3(g), dim3(256), 0, STREAM, (const float*)a, ia, b, ib, c, ic,
                           d, id, out, rows, H);
    return merlot_launch_status("merlot_gather_add4");
}

extern "C" int merlot_dropout_apply(const void* x, void* y, int64_t rows, int64_t N, float p, uint64_t seed,
                                    merlot_stream_t stream) {
    MERLOT_CHECK(x && y && rows > 0 && N > 0 && p >= 0.f && p < 1.f, MERLOT_ESHAPE, "merlot_dropout_apply: bad args");
    const uint32_t thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for(rows * N, 256)), dim3(256), 0, STREAM, (const bf16*)x, (bf16*)y,
                       rows * N, thresh, 1.0f / (1.0f - p), seed);
    return merlot_launch_status("merlot_dropout_apply");
}

extern "C" int merlot_scatter_add_rows(const float* src, const int32_t* idx, float* table, int64_t rows, int H,
                                       merlot_stream_t stream) {
    MERLOT_CHECK(src && table && rows > 0 && H > 0, MERLOT_ESHAPE, "merlot_scatter_add_rows: bad args");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(rows * H, 256)), dim3(256), 0, STREAM, src, idx, table, rows,
                       H);
    return merlot_launch_status("merlot_scatter_add_rows");
}

extern "C" int merlot_scatter_add_sorted(const float* src, const int32_t* perm, const int32_t* sorted_idx, float* table, int64_t rows,
                                         int H, merlot_stream_t stream) {
    MERLOT_CHECK(src && perm && sorted_idx && table && rows > 0 && H > 0 && H % 4 == 0, MERLOT_ESHAPE,
                 "merlot_scatter_add_sorted: bad args (H must be a multiple of 4)");
    MERLOT_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)table & 15) == 0, MERLOT_EALIGN, "merlot_scatter_add_sorted: 16-byte aligned rows");
    const int64_t blocks = (rows + SCS_ROWS - 1) / SCS_ROWS;
    MERLOT_CHECK(blocks < (1LL << 31), MERLOT_ESHAPE, "merlot_scatter_add_sorted: too many rows");
    hipLaunchKernelGGL(scatter_add_sorted_kernel, dim3((unsigned)blocks), dim3(H >= 1024 ? 256 : (H / 4 + 63) / 64 * 64), 0, STREAM, src, perm,
                       sorted_idx, table, rows, H);
    return merlot_launch_status("merlot_scatter_add_sorted");
}

extern "C" int merlot_cls_avgpool_fwd(const void* x, float* out, int n_img, int h1, int w1, int cls_skip, int pool, int H,
                                      merlot_stream_t stream) {
    MERLOT_CHECK(x && out && n_img > 0 && pool >= 1 && h1 % pool == 0 && w1 % pool == 0 && H % 4 == 0 && cls_skip >= 1,
                 MERLOT_ESHAPE, "merlot_cls_avgpool_fwd: bad args");
    const int64_t total = (int64_t)n_img * (1 + (h1 / pool) * (w1 / pool)) * (H / 4);
    hipLaunchKernelGGL(cls_avgpool_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, (const bf16*)x, out, n_img,
                       h1, w1, cls_skip, pool, H);
    return merlot_launch_status("merlot_cls_avgpool_fwd");
}

extern "C" int merlot_cls_avgpool_bwd(const float* dout, void* dx, int n_img, int h1, int w1, int cls_skip, int pool, int H,
                                      merlot_stream_t stream) {
    MERLOT_CHECK(dout && dx && n_img > 0 && pool >= 1 && h1 % pool == 0 && w1 % pool == 0 && H % 4 == 0 && cls_skip >= 1,
                 MERLOT_ESHAPE, "merlot_cls_avgpool_bwd: bad args");
    const int64_t total = (int64_t)n_img * (cls_skip + h1 * w1) * (H / 4);
    hipLaunchKernelGGL(cls_avgpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, dout, (bf16*)dx, n_img, h1,
                       w1, cls_skip, pool, H);
    return merlot_launch_status("merlot_cls_avgpool_bwd");
}

extern "C" int merlot_softmax_ce(const float* logits, int64_t ld, const int32_t* labels, float* loss, int32_t* argmax,
                                 const float* rowscale, void* dlogits, int dl_bf16, int64_t ld_dl, int64_t rows, int C,
                                 merlot_stream_t stream) {
    MERLOT_CHECK(logits && labels && loss && rows > 0 && C > 0 && ld >= C, MERLOT_ESHAPE, "merlot_softmax_ce: bad args");
    MERLOT_CHECK(!dlogits || ld_dl >= C, MERLOT_ESHAPE, "merlot_softmax_ce: ld_dl < C");
    // wide rows: the row-in-registers kernel (its vector accesses need 16-B aligned rows; columns C .. ld_dl of the gradient are
    // zero-filled by both kernels)
    const bool wide = C > 4096 && C <= 65536 && ld % 4 == 0 && (uintptr_t)logits % 16 == 0 &&
                      (!dlogits || (ld_dl % 4 == 0 && ld_dl <= 65536 && (uintptr_t)dlogits % 16 == 0));
    if (wide) {
        if (dl_bf16)
            hipLaunchKernelGGL((softmax_ce_wide_kernel<bf16>), dim3((unsigned)rows), dim3(1024), 0, STREAM, logits, ld, labels, loss,
                               argmax, rowscale, (bf16*)dlogits, ld_dl, C);
        else
            hipLaunchKernelGGL((softmax_ce_wide_kernel<float>), dim3((unsigned)rows), dim3(1024), 0, STREAM, logits, ld, labels, loss,
                               argmax, rowscale, (float*)dlogits, ld_dl, C);
        return merlot_launch_status("merlot_softmax_ce");
    }
    if (dl_bf16)
        hipLaunchKernelGGL((softmax_ce_kernel<bf16>), dim3((unsigned)rows), dim3(256), 0, STREAM, logits, ld, labels, loss,
                           argmax, rowscale, (bf16*)dlogits, ld_dl, C);
    else
        hipLaunchKernelGGL((softmax_ce_kernel<float>), dim3((unsigned)rows), dim3(256), 0, STREAM, logits, ld, labels, loss,
                           argmax, rowscale, (float*)dlogits, ld_dl, C);
    return merlot_launch_status("merlot_softmax_ce");
}

extern "C" int merlot_l2norm_fwd(const float* x, float* y, float* inv_norm, int64_t rows, int H, merlot_stream_t stream) {
    MERLOT_CHECK(x && y && rows > 0 && H > 0, MERLOT_ESHAPE, "merlot_l2norm_fwd: bad args");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, STREAM, x, y, inv_norm, rows, H);
    return merlot_launch_status("merlot_l2norm_fwd");
}
extern "C" int merlot_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t rows, int H,
                                 merlot_stream_t stream) {
    MERLOT_CHECK(dy && y && inv_norm && dx && rows > 0 && H > 0, MERLOT_ESHAPE, "merlot_l2norm_bwd: bad args");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, STREAM, dy, y, inv_norm, dx, rows, H);
    return merlot_launch_status("merlot_l2norm_bwd");
}
extern "C" int merlot_gelu_fwd(const float* x, float* y, int64_t n, merlot_stream_t stream) {
    MERLOT_CHECK(x && y && n > 0, MERLOT_ESHAPE, "merlot_gelu_fwd: bad args");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, x, y, n);
    return merlot_launch_status("merlot_gelu_fwd");
}
extern "C" int merlot_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, merlot_stream_t stream) {
    MERLOT_CHECK(dy && x && dx && n > 0, MERLOT_ESHAPE, "merlot_gelu_bwd: bad args");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, dy, x, dx, n);
    return merlot_launch_status("merlot_gelu_bwd");
}

extern "C" int merlot_adamw_step(float* param, const float* grad, void* m, void* v, int64_t n, float lr, double beta1d,
                                 double beta2d, float eps, float weight_decay, float grad_scale, const uint8_t* wd_flags,
                                 int state_bf16, merlot_stream_t stream) {
    const float beta1 = (float)beta1d, beta2 = (float)beta2d;
    const float omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);
    MERLOT_CHECK(param && grad && m && v && n > 0, MERLOT_ESHAPE, "merlot_adamw_step: bad args");
    const int g = grid_for(n, 256, 8192);
    if (state_bf16)
        hipLaunchKernelGGL((adamw_kernel<true>), dim3(g), dim3(256), 0, STREAM, param, grad, m, v, n, lr, beta1, beta2, omb1,
                           omb2, eps, weight_decay, grad_scale, wd_flags);
    else
        hipLaunchKernelGGL((adamw_kernel<false>), dim3(g), dim3(256), 0, STREAM, param, grad, m, v, n, lr, beta1, beta2, omb1,
                           omb2, eps, weight_decay, grad_scale, wd_flags);
    return merlot_launch_status("merlot_adamw_step");
}
