// bf16 MFMA GEMMs for gfx950 (CDNA4): the NT form (activations x weights, dgrad) and the TN form
// (weight gradients), plus the implicit-im2col patch-embed variants.
//
// Design (see DESIGN.md "GEMM"):
//   * 128x128 block tile, BK = 64, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA 32x32x16 bf16
//     accumulators (64 fp32 regs/lane).
//   * LDS tile image: [128 rows][128 B] with the 16-byte chunk index XOR-swizzled by (row>>1)&7 so a
//     ds_read_b128 lane group (16 lanes = 16 distinct rows mod 16, same chunk) hits 16 distinct 16-B slots.
//   * NT operands are staged HBM->LDS by `global_load_lds_dwordx4` (no VGPR round trip); the swizzle is
//     applied on the per-lane SOURCE address, the LDS destination stays lane-linear.  Two LDS buffers,
//     tile t+1 streams in while tile t is multiplied; one barrier per K step.
//   * the MFMA is issued with operands swapped (D[n][m] = sum_k Bt[n][k] A[m][k]) so every lane owns one
//     output row m and 4 consecutive columns n per accumulator quad -> 8/16-byte epilogue stores.
//   * TN (wgrad) stages through registers: each lane loads dwords (2 adjacent columns) of 8 consecutive
//     reduction rows, transposes them in registers and writes the same swizzled LDS image.  Split-R over
//     the grid with fp32 atomics fills the 256 CUs when M*N is only a few dozen tiles.
//   * block -> tile map is XCD-aware: the 8 XCDs each walk a contiguous band of row tiles, all column
//     tiles of a row tile adjacent, so an A row band is fetched from HBM once per XCD L2.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 64;
constexpr int TILE_BYTES = 128 * 128;  // 128 rows x 64 bf16

struct PatchGeom {
    int H, W, P, h1, w1;  // image height/width, patch, grid
};

struct GemmNTArgs {
    const bf16* A;
    const bf16* B;
    void* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    float alpha;
    const float* bias;
    const bf16* aux_in;
    int64_t ld_aux_in;
    bf16* aux_out;
    int64_t ld_aux_out;
    uint32_t drop_thresh;
    float drop_scale;
    uint64_t drop_seed;
    int accumulate;
    int ntm, ntn;
    PatchGeom pg;
};

struct GemmTNArgs {
    const bf16* A;
    const bf16* B;
    float* C;
    int64_t lda, ldb, ldc;
    int M, N, R;
    float alpha;
    int use_atomics;
    int ntm, ntn, splits, rchunk;
    PatchGeom pg;
};

// XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous id range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// element offset of patch-matrix entry (row, k) inside the NHWC bf16 image; k = (py, px, c), P == 16.
__device__ __forceinline__ int64_t patch_row_base(const PatchGeom& g, int row) {
    const int per_img = g.h1 * g.w1;
    const int n = row / per_img;
    const int rem = row - n * per_img;
    const int ph = rem / g.w1;
    const int pw = rem - ph * g.w1;
    return ((int64_t)(n * g.H + ph * g.P) * g.W + pw * g.P) * 3;
}
__device__ __forceinline__ int patch_k_off(const PatchGeom& g, int k) {
    const int run = g.P * 3;  // 48 contiguous elements per patch row
    const int py = k / run;
    return py * g.W * 3 + (k - py * run);
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

__device__ __forceinline__ void compute_tile(const char* la, const char* lb, const int (&a_row)[2], const int (&b_row)[2],
                                             int hi, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            af[f] = *reinterpret_cast<const bf16x8*>(la + lds_off(a_row[f], 2 * kk + hi));
            bfr[f] = *reinterpret_cast<const bf16x8*>(lb + lds_off(b_row[f], 2 * kk + hi));
        }
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
                acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[fj], af[fi], acc[fi][fj], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F32, bool PATCH>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmNTArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = wgid / p.ntn;
    const int tile_n = wgid - tile_m * p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane staging sources: 4 x 1 KiB pieces of the A tile and of the B tile per wave
    const bf16* a_src[4];
    const bf16* b_src[4];
    int a_chunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ (row >> 1)) & 7;  // logical chunk stored at physical slot lane&7
        const int ga = min(m0 + row, p.M - 1);
        const int gb = min(n0 + row, p.N - 1);
        if (PATCH) {
            a_src[i] = p.A + patch_row_base(p.pg, ga);
            a_chunk[i] = chunk * 8;
        } else {
            a_src[i] = p.A + (int64_t)ga * p.lda + chunk * 8;
            a_chunk[i] = 0;
        }
        b_src[i] = p.B + (int64_t)gb * p.ldb + chunk * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * 2 * TILE_BYTES + wave * 4096;
        char* lb = la + TILE_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16* sa = PATCH ? a_src[i] + patch_k_off(p.pg, k0 + a_chunk[i]) : a_src[i] + k0;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sa), LDS_PTR(la + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(b_src[i] + k0), LDS_PTR(lb + i * 1024), 16, 0, 0);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    int a_row[2], b_row[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        a_row[f] = wm * 64 + f * 32 + (lane & 31);
        b_row[f] = wn * 64 + f * 32 + (lane & 31);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* la = smem + (kt & 1) * 2 * TILE_BYTES;
        compute_tile(la, la + TILE_BYTES, a_row, b_row, hi, acc);
    }

    // ---- epilogue: lane owns row m, quads of 4 consecutive n
    const bool vec_ok = ((p.ldc & 3) == 0) && ((p.ld_aux_in & 3) == 0) && ((p.ld_aux_out & 3) == 0);
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const int m = m0 + wm * 64 + fi * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + fj * 32 + 8 * q + 4 * hi;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[fi][fj][4 * q + e] * p.alpha;
                const bool full = vec_ok && (n + 3 < p.N);
                const int ne = full ? 4 : min(4, p.N - n);
                if (p.bias) {
                    if (full) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b4[e];
                    } else {
                        for (int e = 0; e < ne; ++e) v[e] += p.bias[n + e];
                    }
                }
                if (EPI == MERLOT_EPI_GELU) {
                    if (p.aux_out) {
                        bf16* ao = p.aux_out + (int64_t)m * p.ld_aux_out + n;
                        if (full) {
                            bf16x4 u4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) u4[e] = (bf16)v[e];
                            *reinterpret_cast<bf16x4*>(ao) = u4;
                        } else {
                            for (int e = 0; e < ne; ++e) ao[e] = (bf16)v[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
                } else if (EPI == MERLOT_EPI_DGELU) {
                    const bf16* ai = p.aux_in + (int64_t)m * p.ld_aux_in + n;
                    float u[4] = {0.f, 0.f, 0.f, 0.f};
                    if (full) {
                        const bf16x4 u4 = *reinterpret_cast<const bf16x4*>(ai);
#pragma unroll
                        for (int e = 0; e < 4; ++e) u[e] = (float)u4[e];
                    } else {
                        for (int e = 0; e < ne; ++e) u[e] = (float)ai[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(u[e]);
                } else if (EPI == MERLOT_EPI_RESIDUAL) {
                    if (p.drop_thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint64_t idx = (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e);
                            v[e] = dropout_keep(p.drop_seed, idx, p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
                        }
                    }
                    const bf16* ai = p.aux_in + (int64_t)m * p.ld_aux_in + n;
                    if (full) {
                        const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(ai);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
                    } else {
                        for (int e = 0; e < ne; ++e) v[e] += (float)ai[e];
                    }
                }
                if (OUT_F32) {
                    float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
                    if (full) {
                        f32x4 o;
                        if (p.accumulate) {
                            o = *reinterpret_cast<const f32x4*>(c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += v[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = v[e];
                        }
                        *reinterpret_cast<f32x4*>(c) = o;
                    } else {
                        for (int e = 0; e < ne; ++e) c[e] = p.accumulate ? c[e] + v[e] : v[e];
                    }
                } else {
                    bf16* c = reinterpret_cast<bf16*>(p.C) + (int64_t)m * p.ldc + n;
                    if (full) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                        *reinterpret_cast<bf16x4*>(c) = o;
                    } else {
                        for (int e = 0; e < ne; ++e) c[e] = (bf16)v[e];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN kernel (weight gradients): C[m][n] += alpha * sum_r A[r][m] B[r][n]
// ------------------------------------------------------------------------------------------------
template <bool PATCH_B>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmTNArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.ntm * p.ntn;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wgid / ntiles;
    const int tile = wgid - split * ntiles;
    const int tile_m = tile / p.ntn;
    const int tile_n = tile - tile_m * p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int r_begin = split * p.rchunk;
    const int r_end = min(p.R, r_begin + p.rchunk);
    const int nsteps = (r_end - r_begin + BK - 1) / BK;

    // lane owns columns (2*lane, 2*lane+1) of the tile for both operands
    const int am = min(m0 + 2 * lane, p.M - 2);
    const int bn = min(n0 + 2 * lane, p.N - 2);
    const bf16* a_col = p.A + am;
    const bf16* b_col = PATCH_B ? p.B + patch_k_off(p.pg, bn) : p.B + bn;

    uint32_t xa[2][8], xb[2][8];
    auto load = [&](int st) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int r0 = r_begin + st * BK + (2 * wave + g) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i;
                if (r < r_end) {
                    xa[g][i] = *reinterpret_cast<const uint32_t*>(a_col + (int64_t)r * p.lda);
                    const bf16* bp = PATCH_B ? b_col + patch_row_base(p.pg, r) : b_col + (int64_t)r * p.ldb;
                    xb[g][i] = *reinterpret_cast<const uint32_t*>(bp);
                } else {
                    xa[g][i] = 0u;
                    xb[g][i] = 0u;
                }
            }
        }
    };
    auto write = [&](int buf) {
        char* la = smem + buf * 2 * TILE_BYTES;
        char* lb = la + TILE_BYTES;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int chunk = 2 * wave + g;
            u32x4 a0, a1, b0, b1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = (xa[g][2 * e] & 0xffffu) | (xa[g][2 * e + 1] << 16);
                a1[e] = (xa[g][2 * e] >> 16) | (xa[g][2 * e + 1] & 0xffff0000u);
                b0[e] = (xb[g][2 * e] & 0xffffu) | (xb[g][2 * e + 1] << 16);
                b1[e] = (xb[g][2 * e] >> 16) | (xb[g][2 * e + 1] & 0xffff0000u);
            }
            *reinterpret_cast<u32x4*>(la + lds_off(2 * lane, chunk)) = a0;
            *reinterpret_cast<u32x4*>(la + lds_off(2 * lane + 1, chunk)) = a1;
            *reinterpret_cast<u32x4*>(lb + lds_off(2 * lane, chunk)) = b0;
            *reinterpret_cast<u32x4*>(lb + lds_off(2 * lane + 1, chunk)) = b1;
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    int a_row[2], b_row[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        a_row[f] = wm * 64 + f * 32 + (lane & 31);
        b_row[f] = wn * 64 + f * 32 + (lane & 31);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nsteps > 0) {
        load(0);
        write(0);
        __syncthreads();
        for (int st = 0; st < nsteps; ++st) {
            if (st + 1 < nsteps) load(st + 1);
            const char* la = smem + (st & 1) * 2 * TILE_BYTES;
            compute_tile(la, la + TILE_BYTES, a_row, b_row, hi, acc);
            if (st + 1 < nsteps) write((st + 1) & 1);
            __syncthreads();
        }
    }

#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const int m = m0 + wm * 64 + fi * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + fj * 32 + 8 * q + 4 * hi;
                float* c = p.C + (int64_t)m * p.ldc + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < p.N) {
                        const float v = acc[fi][fj][4 * q + e] * p.alpha;
                        if (p.use_atomics)
                            atomicAdd(c + e, v);
                        else
                            c[e] = v;
                    }
                }
            }
        }
    }
}

template <int EPI, bool PATCH>
int launch_nt(const GemmNTArgs& a, int out_f32, hipStream_t s) {
    const int grid = a.ntm * a.ntn;
    if (out_f32)
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, true, PATCH>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, false, PATCH>), dim3(grid), dim3(256), 0, s, a);
    return merlot_launch_status("merlot_gemm_bf16_nt");
}

int gemm_nt_dispatch(GemmNTArgs& a, int epilogue, int out_f32, bool patch, hipStream_t s) {
    a.ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.N, BN);
    if (patch) {
        MERLOT_CHECK(epilogue == MERLOT_EPI_NONE, MERLOT_ESHAPE, "patch embed supports EPI_NONE only");
        return launch_nt<MERLOT_EPI_NONE, true>(a, out_f32, s);
    }
    switch (epilogue) {
        case MERLOT_EPI_NONE: return launch_nt<MERLOT_EPI_NONE, false>(a, out_f32, s);
        case MERLOT_EPI_GELU: return launch_nt<MERLOT_EPI_GELU, false>(a, out_f32, s);
        case MERLOT_EPI_RESIDUAL: return launch_nt<MERLOT_EPI_RESIDUAL, false>(a, out_f32, s);
        case MERLOT_EPI_DGELU: return launch_nt<MERLOT_EPI_DGELU, false>(a, out_f32, s);
    }
    merlot_set_error("merlot_gemm_bf16_nt: unknown epilogue %d", epilogue);
    return MERLOT_ESHAPE;
}

int gemm_tn_dispatch(GemmTNArgs& a, int accumulate, bool patch_b, hipStream_t s) {
    a.ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.N, BN);
    const int tiles = a.ntm * a.ntn;
    int splits = cdiv(768, tiles);
    const int max_splits = (a.R + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rchunk = (int)(((a.R + splits - 1) / splits + BK - 1) / BK) * BK;
    splits = (a.R + rchunk - 1) / rchunk;
    a.splits = splits;
    a.rchunk = rchunk;
    a.use_atomics = (splits > 1 || accumulate) ? 1 : 0;
    if (splits > 1 && !accumulate) {
        hipError_t e = hipMemset2DAsync(a.C, (size_t)a.ldc * 4, 0, (size_t)a.N * 4, (size_t)a.M, s);
        MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "merlot_gemm_bf16_tn: memset failed: %s", hipGetErrorString(e));
    }
    const int grid = tiles * splits;
    if (patch_b)
        hipLaunchKernelGGL((gemm_tn_kernel<true>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<false>), dim3(grid), dim3(256), 0, s, a);
    return merlot_launch_status("merlot_gemm_bf16_tn");
}

}  // namespace

extern "C" int merlot_gemm_bf16_nt(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc,
                                   int64_t M, int64_t N, int64_t K, float alpha, int epilogue, int out_f32,
                                   int accumulate, const float* bias, const void* aux_in, int64_t ld_aux_in,
                                   void* aux_out, int64_t ld_aux_out, float dropout_p, uint64_t dropout_seed,
                                   merlot_stream_t stream) {
    MERLOT_CHECK(A && Bt && C, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: null operand");
    MERLOT_CHECK(M > 0 && N > 0 && K > 0 && M < (1LL << 31) && N < (1LL << 31), MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    MERLOT_CHECK(K % BK == 0, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: K=%lld must be a multiple of %d", (long long)K, BK);
    MERLOT_CHECK(lda % 8 == 0 && ldb % 8 == 0, MERLOT_EALIGN, "merlot_gemm_bf16_nt: lda/ldb must be multiples of 8");
    MERLOT_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)Bt & 15) == 0, MERLOT_EALIGN,
                 "merlot_gemm_bf16_nt: A/Bt must be 16-byte aligned");
    MERLOT_CHECK(!(accumulate && !out_f32), MERLOT_EDTYPE, "merlot_gemm_bf16_nt: accumulate needs f32 output");
    MERLOT_CHECK(dropout_p >= 0.f && dropout_p < 1.f, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: dropout_p out of range");
    if (epilogue == MERLOT_EPI_RESIDUAL || epilogue == MERLOT_EPI_DGELU)
        MERLOT_CHECK(aux_in != nullptr, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: epilogue %d needs aux_in", epilogue);
    GemmNTArgs a{};
    a.A = (const bf16*)A; a.B = (const bf16*)Bt; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.alpha = alpha; a.bias = bias;
    a.aux_in = (const bf16*)aux_in; a.ld_aux_in = aux_in ? ld_aux_in : 0;
    a.aux_out = (bf16*)aux_out; a.ld_aux_out = aux_out ? ld_aux_out : 0;
    a.drop_thresh = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    a.drop_scale = 1.0f / (1.0f - dropout_p);
    a.drop_seed = dropout_seed;
    a.accumulate = accumulate;
    return gemm_nt_dispatch(a, epilogue, out_f32, false, (hipStream_t)stream);
}

extern "C" int merlot_gemm_bf16_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                   int64_t M, int64_t N, int64_t R, float alpha, int accumulate,
                                   merlot_stream_t stream) {
    MERLOT_CHECK(A && B && C, MERLOT_ESHAPE, "merlot_gemm_bf16_tn: null operand");
    MERLOT_CHECK(M > 1 && N > 1 && R > 0 && R < (1LL << 31), MERLOT_ESHAPE, "merlot_gemm_bf16_tn: bad dims");
    MERLOT_CHECK(M % 2 == 0 && N % 2 == 0 && lda % 2 == 0 && ldb % 2 == 0, MERLOT_EALIGN,
                 "merlot_gemm_bf16_tn: M, N, lda, ldb must be even");
    GemmTNArgs a{};
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.R = (int)R; a.alpha = alpha;
    return gemm_tn_dispatch(a, accumulate, false, (hipStream_t)stream);
}

static int check_patch(int n_img, int H, int W, int P, int hidden) {
    MERLOT_CHECK(P == 16, MERLOT_ESHAPE, "patch embed: only patch_size 16 is supported (got %d)", P);
    MERLOT_CHECK(n_img > 0 && H % P == 0 && W % P == 0, MERLOT_ESHAPE, "patch embed: H, W must be multiples of P");
    MERLOT_CHECK((W * 3) % 8 == 0, MERLOT_EALIGN, "patch embed: W*3 must be a multiple of 8");
    MERLOT_CHECK(hidden > 1, MERLOT_ESHAPE, "patch embed: bad hidden");
    return MERLOT_OK;
}

extern "C" int merlot_patch_embed_fwd(const void* image, int n_img, int H, int W, int P, const void* Wt,
                                      const float* bias_folded, void* out, int hidden, merlot_stream_t stream) {
    int rc = check_patch(n_img, H, W, P, hidden);
    if (rc) return rc;
    GemmNTArgs a{};
    a.A = (const bf16*)image; a.B = (const bf16*)Wt; a.C = out;
    a.lda = 0; a.ldb = P * P * 3; a.ldc = hidden;
    a.M = n_img * (H / P) * (W / P); a.N = hidden; a.K = P * P * 3;
    a.alpha = 1.f; a.bias = bias_folded;
    a.pg = PatchGeom{H, W, P, H / P, W / P};
    return gemm_nt_dispatch(a, MERLOT_EPI_NONE, 0, true, (hipStream_t)stream);
}

extern "C" int merlot_patch_embed_wgrad(const void* image, int n_img, int H, int W, int P, const void* dY, float* dWt,
                                        int hidden, int accumulate, merlot_stream_t stream) {
    int rc = check_patch(n_img, H, W, P, hidden);
    if (rc) return rc;
    // dWt[hidden][k] = sum_rows dY[row][hidden] * patch[row][k]   (the -0.5 shift is applied by the caller
    // through the bias gradient: d/dW of (x-0.5)W = x^T dY - 0.5 * colsum(dY))
    GemmTNArgs a{};
    a.A = (const bf16*)dY; a.B = (const bf16*)image; a.C = dWt;
    a.lda = hidden; a.ldb = 0; a.ldc = P * P * 3;
    a.M = hidden; a.N = P * P * 3; a.R = n_img * (H / P) * (W / P); a.alpha = 1.f;
    a.pg = PatchGeom{H, W, P, H / P, W / P};
    return gemm_tn_dispatch(a, accumulate, true, (hipStream_t)stream);
}
