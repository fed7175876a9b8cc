// bf16 MFMA GEMMs for gfx950 (CDNA4): the NT form (activations x weights, dgrad) and the TN form (weight gradients).
// See DESIGN.md 3.1 for the measurements behind every choice.
//
// Kernels in this file
//   gemm_nt_p8_kernel (gemm_p8.inc) production NT for large problems: persistent 256x256 tiles, 8 waves in two ping-pong groups,
//                                BK 64, one continuous LDS-DMA stream across tiles, dynamic per-XCD tile claims, interior
//                                tiles through fast_tile_epilogue (bias once per tile, prefetched auxiliary operand).
//                                (Rounds 1-3 also kept the lock-step persistent kernels it replaced -- static striding for K < 128,
//                                64-bit addressing for operands of 4 GiB and more; retired in round 4: K = 64 runs the 128x256
//                                ring kernel, larger operands are cut into row ranges of the same kernel.)
//   gemm_nt_ring_kernel<Cfg>     non-persistent ring kernel: 128x256 (ragged tile counts, 2 WG/CU), 256x64 and
//                                256x128 (narrow outputs), 256x256 (experiments).
//   gemm_tn_ring_kernel<Cfg>     wgrad: operands as stored ([R][M], [R][N]) through ds_read_b64_tr_b16, split-R
//                                partials to a caller-owned workspace + tn_reduce_kernel (no atomics).
//   gemm_nt_kernel / gemm_tn_kernel   first-generation 128x128, 2-buffer kernels: tails and shapes the ring kernels refuse.
// Common to all: v_mfma_f32_32x32x16_bf16 issued with swapped operands (a lane owns one output row and 4 consecutive
// columns per accumulator quad), LDS tile images [rows][BK] with the 16-B chunk index XOR-swizzled (applied on the
// per-lane SOURCE address of `global_load_lds_dwordx4`; the LDS destination stays lane-linear), counted
// `s_waitcnt vmcnt(N)` + raw `s_barrier`, XCD-aware block -> tile maps, row-contiguous epilogues staged through LDS.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_ring.h"

extern "C" int merlot_colsum_bf16(const void* x, int64_t ld, float* out, int64_t T, int64_t N, int accumulate,
                                  merlot_stream_t stream);

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 64;
constexpr int TILE_BYTES = 128 * 128;  // 128 rows x 64 bf16

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
    float* colsum;        // optional f32 [N], ACCUMULATED: column sums of the stored (bf16-rounded) C -- the bias gradient of the
                          // layer whose output gradient this GEMM produces; only kernels that say so support it (else host fallback)
    const float* scale_a; // fp8 kernels only: device scalars, the per-tensor dequantisation factors of A and B (alpha *= both)
    const float* scale_b;
    const float* row_scale;  // fp8 kernels only, optional: device f32 [M], per-ROW dequantisation factors of A (merlot_ln_fwd_q8)
    unsigned int* ctr;    // persistent kernels with dynamic tile claims: the CALLER's counter block (merlot_gemm_nt_workspace_bytes():
                          // [0..7] claims per XCD, [8] departures), zero on entry, left zero by the last workgroup out
    int ntm, ntn;
    int cg;               // persistent kernel: tiles are enumerated in groups of `cg` tile columns (0: plain row-major)
    int dephase;          // experiments only: workgroup i of an XCD starts ((i * 5) & 7) * dephase shader cycles late (0: off)
    int dbg;              // experiments only: 1 = skip epilogue, 2 = skip main loop
    int64_t m_off;        // rows of C in front of this launch when a GEMM is cut into row ranges: the dropout mask is a function of the
                          // element's GLOBAL index (m_off + m) * N + n
    // Round 6: LayerNorm of the finished rows of C out of the SAME launch (ping-pong kernel, RESIDUAL epilogue, N = 768 = three
    // tile columns; gemm_p8.inc "LNF").  ln_out == nullptr: off.
    bf16* ln_out;         // [M, N] bf16: LN(C) of every row of a whole 256-row block (the tail rows are the host's: merlot_ln_fwd)
    int64_t ld_ln;
    const float* ln_gamma;
    const float* ln_beta;
    float* ln_mean;       // f32 [M] (may be nullptr)
    float* ln_rstd;
    float ln_eps;
    float* ln_part;       // caller workspace: [N / 64][ntm * 256] x (sum, M2) of each row's 64-column segments, written by the tile epilogues
    unsigned int* ln_ctr; // caller workspace: [ntm] arrivals per row block, zero on entry, left zero
    // Round 6: an 8-bit float copy of the (bf16-rounded) output from the SAME epilogue (ping-pong kernel, GELU / DGELU epilogues, interior tiles only:
    // the host refuses ragged shapes) -- the operand of the 8-bit weight gradient (gemm_q8.inc) without a pass of its own.  q8_out == nullptr: off.
    uint8_t* q8_out;      // [M, N] bytes: f8(clamp(bf16(C) * q8_scale[0])), format by the kernel's Q8 template value (e4m3 / e5m2)
    int64_t ld_q8;
    const float* q8_scale;   // device float[4], merlot_quantize_f8's block: [0] = s (DELAYED: made from the amax an earlier launch recorded)
    unsigned int* q8_amax;   // &block[3] as bits: max|bf16(C)| of this launch is atomically max-ed into it (one atomic per wave and LAUNCH)
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
    float* colsum_a;      // round 6, one-phase ping-pong kernel only (else host fallback): f32 [colsum_m], ACCUMULATED: sum_r A[r][m] for m < colsum_m -- the
    int colsum_m;         // bias gradient of the layer whose weight gradient this is, from the A fragments the kernel holds anyway (dot products with 1)
    int dbg;
};

// XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous id range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
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

// epilogue for 4 consecutive output columns n..n+3 of row m (v = alpha * accumulator; FOLD: v = the raw accumulator, alpha joins the bias as ONE
// fused multiply-add -- the rounding of the ping-pong kernel's interior-tile epilogue, so that an output's bits do not depend on whether an element
// lies in an interior or a boundary tile)
template <int EPI, bool OUT_F32, bool FOLD = false>
__device__ __forceinline__ void nt_epilogue_quad(const GemmNTArgs& p, int m, int n, float (&v)[4], bool vec_ok) {
        const bool full = vec_ok && (n + 3 < p.N);
        const int ne = full ? 4 : min(4, p.N - n);
        if (FOLD) {
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                for (int e = 0; e < ne; ++e) b[e] = p.bias[n + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], p.alpha, b[e]);
        } else if (p.bias) {
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
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
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
            for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast(u[e]);
        } else if (EPI == MERLOT_EPI_RESIDUAL) {
            if (p.drop_thresh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint64_t idx = (uint64_t)(m + p.m_off) * (uint64_t)p.N + (uint64_t)(n + e);
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

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F32>
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ (row >> 1)) & 7;  // logical chunk stored at physical slot lane&7
        const int ga = min(m0 + row, p.M - 1);
        const int gb = min(n0 + row, p.N - 1);
        a_src[i] = p.A + (int64_t)ga * p.lda + chunk * 8;
        b_src[i] = p.B + (int64_t)gb * p.ldb + chunk * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * 2 * TILE_BYTES + wave * 4096;
        char* lb = la + TILE_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[i] + k0), LDS_PTR(la + i * 1024), 16, 0, 0);
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
                nt_epilogue_quad<EPI, OUT_F32>(p, m, n, v, vec_ok);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN kernel (weight gradients): C[m][n] += alpha * sum_r A[r][m] B[r][n]
// ------------------------------------------------------------------------------------------------
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
    const bf16* b_col = p.B + bn;

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
                    xb[g][i] = *reinterpret_cast<const uint32_t*>(b_col + (int64_t)r * p.ldb);
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

// ------------------------------------------------------------------------------------------------
// Row-contiguous epilogue.  The MFMA accumulator layout gives a lane 4 consecutive columns of ONE row per quad, so
// direct stores touch 32 rows x 16 B per wave-instruction (measured: ~30 us to write a [50688, 768] bf16 output,
// a third of a K=768 GEMM).  Instead each wave transposes its 32 x (FN*32) fp32 slab through a private LDS buffer
// and then works on whole row segments: a lane owns 8 consecutive columns, so the output store, the residual /
// pre-activation loads and the bias loads are all 16-32 B per lane and 128-512 contiguous bytes per row.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F32, bool FOLD = false, int Q8 = 0>
__device__ __forceinline__ void epilogue_row8(const GemmNTArgs& p, int m, int n, float (&v)[8], float* cacc = nullptr, float* q8_m = nullptr) {
    // v = alpha * acc for columns n..n+7 of row m (FOLD: the raw accumulators, see nt_epilogue_quad); n + 7 < N guaranteed, 16-B alignment
    // guaranteed by the caller
    if DBG_BIT(p, 8) {                                     // experiments: epilogue arithmetic + staging, no memory ops
        if (v[0] == 12345.678f) *reinterpret_cast<float*>(p.C) = v[1];
        return;
    }
    if (FOLD) {
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = __builtin_fmaf(v[e], p.alpha, b0[e]);
            v[4 + e] = __builtin_fmaf(v[4 + e], p.alpha, b1[e]);
        }
    } else if (p.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
        }
    }
    if (EPI == MERLOT_EPI_GELU) {
        if (p.aux_out) {
            bf16x8 u8;
#pragma unroll
            for (int e = 0; e < 8; ++e) u8[e] = (bf16)v[e];
            *reinterpret_cast<bf16x8*>(p.aux_out + (int64_t)m * p.ld_aux_out + n) = u8;
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) gelu_fast2(v[e], v[e + 1]);
    } else if (EPI == MERLOT_EPI_DGELU) {
        const bf16x8 u8 = *reinterpret_cast<const bf16x8*>(p.aux_in + (int64_t)m * p.ld_aux_in + n);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float g0, g1;
            gelu_grad_fast2((float)u8[e], (float)u8[e + 1], g0, g1);
            v[e] *= g0;
            v[e + 1] *= g1;
        }
    } else if (EPI == MERLOT_EPI_RESIDUAL) {
        if (p.drop_thresh) {
            bool keep[8];
            const uint64_t idx0 = (uint64_t)(m + p.m_off) * (uint64_t)p.N + (uint64_t)n;      // n % 8 == 0 on this path
            if (!(p.N & 1)) {
                dropout_keep_n<8>(p.drop_seed, idx0, p.drop_thresh, keep);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) keep[e] = dropout_keep(p.drop_seed, idx0 + e, p.drop_thresh);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = keep[e] ? v[e] * p.drop_scale : 0.f;
        }
        const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(p.aux_in + (int64_t)m * p.ld_aux_in + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
    }
    if (OUT_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
        f32x4 o0, o1;
        if (p.accumulate) {
            o0 = *reinterpret_cast<const f32x4*>(c);
            o1 = *reinterpret_cast<const f32x4*>(c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] += v[e];
                o1[e] += v[4 + e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = v[e];
                o1[e] = v[4 + e];
            }
        }
        *reinterpret_cast<f32x4*>(c) = o0;
        *reinterpret_cast<f32x4*>(c + 4) = o1;
    } else {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        if constexpr (!(Q8 & 4)) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + (int64_t)m * p.ldc + n) = o;
        if (cacc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cacc[e] += (float)o[e];
        }
        if constexpr (Q8 != 0) {                         // the 8-bit copy in a row block's ragged last tile: the block of fast_tile_epilogue, per row
            constexpr float FMAX8 = (Q8 & 3) == 1 ? 448.f : 57344.f;
            const float q8_s = p.q8_scale[0];
            float mx = *q8_m, f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mx = fmaxf(mx, fabsf(v[e]));
                f[e] = __builtin_amdgcn_fmed3f(v[e] * q8_s, -FMAX8, FMAX8);
            }
            *q8_m = mx;
            int w0 = 0, w1 = 0;
            if constexpr ((Q8 & 3) == 1) {
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            } else {
                w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w0, false);
                w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], w1, false);
                w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], w1, true);
            }
            *reinterpret_cast<u32x2*>(p.q8_out + (int64_t)m * p.ld_q8 + n) = u32x2{(uint32_t)w0, (uint32_t)w1};
        }
    }
}

// column sums of a 64-column slab: lane (lane & 7) owns 8 columns, the 8 lanes lane >> 3 own different rows.  Fold the rows,
// then one atomic per column from lanes 0..7.
__device__ __forceinline__ void colsum_flush(float* colsum, int n0, int lane, float (&cacc)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = cacc[e];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lane < 8) atomicAdd(colsum + n0 + lane * 8 + e, t);
    }
}

// whole-workgroup epilogue: `stage` = this wave's private LDS buffer of 32 * (FN*32*4 + 16) bytes
template <typename C, int EPI, bool OUT_F32>
__device__ __forceinline__ void staged_epilogue(const GemmNTArgs& p, f32x16 (&acc)[C::FM][C::FN], char* stage, int m_base,
                                                int n_base, int lane) {
    constexpr int COLS = C::FN * 32;
    constexpr int RSTRIDE = COLS * 4 + 16;           // +16 B: the 8 rows of a ds_write_b128 lane group tile all 32 banks
    constexpr int LPR = COLS / 8;                    // lanes per row in the row-contiguous phase
    constexpr int RPP = 64 / LPR;                    // rows per pass
    constexpr int PPB = 32 / RPP;                    // passes per 32-row block
    constexpr int NPASS = C::FM * PPB;
    constexpr bool HAS_AUX = (EPI == MERLOT_EPI_RESIDUAL) || (EPI == MERLOT_EPI_DGELU);
    const int hi = lane >> 5;
    const int rr = lane / LPR, c0 = (lane % LPR) * 8;
    const int n = n_base + c0;                       // this lane's 8 columns, the same in every pass
    const bool aligned = ((p.ldc & 7) == 0) && ((p.ld_aux_in & 7) == 0) && ((p.ld_aux_out & 7) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.aux_in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.aux_out) & 15) == 0);
    // fast path (wave-uniform): every row and column of this wave's slab exists and every operand is 16-B aligned.
    // Then nothing below waits on a fresh global load: bias is read once, the auxiliary operand of ALL passes is
    // requested up front (the old code re-loaded bias after every store -- it may alias C -- and loaded each auxiliary
    // row segment where it was consumed: one dependent L2/HBM round trip per pass).
    const bool fast = aligned && !(p.N & 1) && !(OUT_F32 && p.accumulate) && m_base + C::FM * 32 <= p.M &&
                      n_base + COLS <= p.N && NPASS <= 16;
    f32x4 b0, b1;
    bf16x8 aux[NPASS <= 16 ? NPASS : 1];
    if (fast) {
        if (p.bias) {
            b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b0[e] = b1[e] = 0.f;
        }
        if (HAS_AUX) {
#pragma unroll
            for (int q = 0; q < NPASS; ++q)
                aux[q] = *reinterpret_cast<const bf16x8*>(p.aux_in + (int64_t)(m_base + (q / PPB) * 32 + (q % PPB) * RPP + rr) * p.ld_aux_in + n);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int fi = 0; fi < C::FM; ++fi) {
        // phase 1: accumulators -> LDS slab [32 rows][COLS] fp32
#pragma unroll
        for (int fj = 0; fj < C::FN; ++fj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = acc[fi][fj][4 * q + e] * p.alpha;
                *reinterpret_cast<f32x4*>(stage + (lane & 31) * RSTRIDE + (fj * 32 + 8 * q + 4 * hi) * 4) = t;
            }
        // same wave writes and reads: only the LDS queue has to drain (no workgroup barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // phase 2: row segments
#pragma unroll
        for (int ps = 0; ps < PPB; ++ps) {
            const int r = ps * RPP + rr;
            const int m = m_base + fi * 32 + r;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(stage + r * RSTRIDE + c0 * 4);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(stage + r * RSTRIDE + c0 * 4 + 16);
            if (fast) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = x0[e] + b0[e];
                    v[4 + e] = x1[e] + b1[e];
                }
                if (EPI == MERLOT_EPI_GELU) {
                    if (p.aux_out) {
                        bf16x8 u8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) u8[e] = (bf16)v[e];
                        *reinterpret_cast<bf16x8*>(p.aux_out + (int64_t)m * p.ld_aux_out + n) = u8;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) gelu_fast2(v[e], v[e + 1]);
                } else if (EPI == MERLOT_EPI_DGELU) {
                    const bf16x8 u8 = aux[fi * PPB + ps];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        float g0, g1;
                        gelu_grad_fast2((float)u8[e], (float)u8[e + 1], g0, g1);
                        v[e] *= g0;
                        v[e + 1] *= g1;
                    }
                } else if (EPI == MERLOT_EPI_RESIDUAL) {
                    if (p.drop_thresh) {
                        bool keep[8];
                        dropout_keep_n<8>(p.drop_seed, (uint64_t)(m + p.m_off) * (uint64_t)p.N + (uint64_t)n, p.drop_thresh, keep);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = keep[e] ? v[e] * p.drop_scale : 0.f;
                    }
                    const bf16x8 r8 = aux[fi * PPB + ps];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                }
                if (OUT_F32) {
                    float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
                    f32x4 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = v[e];
                        o1[e] = v[4 + e];
                    }
                    *reinterpret_cast<f32x4*>(c) = o0;
                    *reinterpret_cast<f32x4*>(c + 4) = o1;
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + (int64_t)m * p.ldc + n) = o;
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if (m >= p.M || n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = x0[e];
                v[4 + e] = x1[e];
            }
            if (aligned && n + 7 < p.N) {
                epilogue_row8<EPI, OUT_F32>(p, m, n, v);
            } else {                                  // ragged right edge / unaligned leading dims: per-quad path
                float q0[4] = {x0[0], x0[1], x0[2], x0[3]}, q1[4] = {x1[0], x1[1], x1[2], x1[3]};
                const bool vec_ok = ((p.ldc & 3) == 0) && ((p.ld_aux_in & 3) == 0) && ((p.ld_aux_out & 3) == 0);
                nt_epilogue_quad<EPI, OUT_F32>(p, m, n, q0, vec_ok);
                if (n + 4 < p.N) nt_epilogue_quad<EPI, OUT_F32>(p, m, n + 4, q1, vec_ok);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// NT kernel, ring-pipelined (production path; the 2-buffer kernel above is kept as the small-shape fallback)
// ------------------------------------------------------------------------------------------------
// waves per SIMD a configuration is compiled for (register budget 512 / OCC): 1 unless the configuration says otherwise
template <typename C, typename = void>
struct ring_occ { static constexpr int value = 1; };
template <typename C>
struct ring_occ<C, std::void_t<decltype(C::OCC)>> { static constexpr int value = C::OCC; };

template <typename C, int EPI, bool OUT_F32>
__global__ __launch_bounds__(C::NT, ring_occ<C>::value) void gemm_nt_ring_kernel(const GemmNTArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    constexpr int BK = C::BK, S = C::STAGES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = wgid / p.ntn;
    const int tile_n = wgid - tile_m * p.ntn;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

    // ---- per-lane LDS-DMA sources
    constexpr int CH = BK / 8;                       // 16-B chunks per row
    const int prow = lane / CH;                      // row inside a piece
    const int pch = lane % CH;                       // physical chunk slot
    const bf16* a_src[C::A_PIECES];
    const bf16* b_src[C::B_PIECES];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int row = (wave * C::A_PIECES + i) * C::RP + prow;
        const int chunk = BK == 64 ? ((pch ^ (row >> 1)) & 7) : ((pch ^ (row >> 2)) & 3);
        a_src[i] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < C::B_PIECES; ++i) {
        const int row = (wave * C::B_PIECES + i) * C::RP + prow;
        const int chunk = BK == 64 ? ((pch ^ (row >> 1)) & 7) : ((pch ^ (row >> 2)) & 3);
        b_src[i] = p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + chunk * 8;
    }
    auto stage = [&](int kt) {
        if DBG_BIT(p, 4) return;                       // experiments: no global->LDS traffic at all
        char* la = dsm + (kt % S) * C::STAGE_BYTES + wave * C::A_PIECES * 1024;
        char* lb = dsm + (kt % S) * C::STAGE_BYTES + C::A_BYTES + wave * C::B_PIECES * 1024;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[i] + k0), LDS_PTR(la + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(b_src[i] + k0), LDS_PTR(lb + i * 1024), 16, 0, 0);
    };

    const int wm = wave / C::WN, wn = wave % C::WN;
    const int hi = lane >> 5;
    int a_row[C::FM], b_row[C::FN];
#pragma unroll
    for (int f = 0; f < C::FM; ++f) a_row[f] = (wm * C::FM + f) * 32 + (lane & 31);
#pragma unroll
    for (int f = 0; f < C::FN; ++f) b_row[f] = (wn * C::FN + f) * 32 + (lane & 31);

    f32x16 acc[C::FM][C::FN];
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int kt) {
        const char* la = dsm + (kt % S) * C::STAGE_BYTES;
        const char* lb = la + C::A_BYTES;
        // all fragment reads of the step first, into DISTINCT registers: otherwise the compiler re-uses one register
        // set per k-substep and every MFMA group waits (lgkmcnt(0)) for reads issued after the previous group.
        constexpr int KK = BK / 16;
        bf16x8 af[KK][C::FM], bfr[KK][C::FN];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int f = 0; f < C::FM; ++f)
                af[kk][f] = *reinterpret_cast<const bf16x8*>(la + ring::kc_off<BK>(a_row[f], 2 * kk + hi));
#pragma unroll
            for (int f = 0; f < C::FN; ++f)
                bfr[kk][f] = *reinterpret_cast<const bf16x8*>(lb + ring::kc_off<BK>(b_row[f], 2 * kk + hi));
        }
        __builtin_amdgcn_sched_barrier(0);           // keep the reads above the MFMAs (the scheduler sinks them back)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int fi = 0; fi < C::FM; ++fi)
#pragma unroll
                for (int fj = 0; fj < C::FN; ++fj)
                    acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[kk][fj], af[kk][fi], acc[fi][fj], 0, 0, 0);
    };

    const int nk = DBG_BIT(p, 2) ? 0 : p.K / BK;
    // prologue: S-1 stages in flight
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) stage(t);
    const int steady = nk - (S - 1);                 // steps whose prefetch target exists
    int kt = 0;
    for (; kt < steady; ++kt) {
        ring::wait_vmcnt<(S - 2) * C::LOADS>();      // stage kt landed; the newer S-2 stages stay in flight
        __builtin_amdgcn_s_barrier();
        stage(kt + S - 1);
        compute(kt);
    }
    for (; kt < nk; ++kt) {                          // drain
        ring::wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(kt);
    }

    // ---- epilogue (row-contiguous, staged through this wave's slice of the now idle ring)
    if (DBG_BIT(p, 1) && acc[0][0][0] != 12345.678f) return;
    __builtin_amdgcn_s_barrier();                    // every wave is done reading the last ring stage
    staged_epilogue<C, EPI, OUT_F32>(p, acc, dsm + wave * C::EPI_BYTES, m0 + wm * C::FM * 32, n0 + wn * C::FN * 32, lane);
}

// ------------------------------------------------------------------------------------------------
// NT kernel, PERSISTENT variant for large problems: one workgroup per CU walks its tiles and the LDS-DMA ring never
// drains between tiles -- while a tile's epilogue runs, the first STAGES-1 K-steps of the next tile are already in
// flight, so the ~20 us prologue + epilogue of a short-K (768) tile overlaps with memory traffic instead of adding
// to it.  Epilogue staging has its own LDS region (XOR-swizzled 32x64 fp32 slab per wave), ring + staging = 160 KiB.
// Counted vmcnt stays valid across the epilogue's own loads/stores: loads complete in order, so "<= (S-2)*LOADS
// outstanding" still implies stage t has landed; stores only make the wait more conservative.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F32, bool RS = false, int Q8 = 0>
__device__ __forceinline__ void slab_epilogue(const GemmNTArgs& p, const f32x16& acc0, const f32x16& acc1, char* slab,
                                              int m_base, int n_base, int lane, float* q8_m = nullptr) {
    // slab: [32 rows][64 cols] fp32, 16-B chunk index XOR (row & 15); acc0 = columns 0..31, acc1 = columns 32..63
    const int hi = lane >> 5;
    const int row = lane & 31;
    float ra = p.alpha;                                  // RS (fp8 operands): times this row's dequantisation factor
    if (RS) {
        if (p.row_scale && m_base + row < p.M) ra *= p.row_scale[m_base + row];
    }
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = RS ? (fj ? acc1[4 * q + e] : acc0[4 * q + e]) * ra : (fj ? acc1[4 * q + e] : acc0[4 * q + e]);
            const int chunk = fj * 8 + 2 * q + hi;      // (!RS: raw accumulators; alpha joins the bias below as one FMA, as in fast_tile_epilogue)
            *reinterpret_cast<f32x4*>(slab + row * 256 + ((chunk ^ (row & 15)) << 4)) = t;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const bool aligned = ((p.ldc & 7) == 0) && ((p.ld_aux_in & 7) == 0) && ((p.ld_aux_out & 7) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
    float cacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (lane >> 3);
        const int c0 = (lane & 7) * 8;
        const int ch = 2 * (lane & 7);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(slab + r * 256 + ((ch ^ (r & 15)) << 4));
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(slab + r * 256 + (((ch + 1) ^ (r & 15)) << 4));
        const int m = m_base + r, n = n_base + c0;
        if (m >= p.M || n >= p.N) continue;
        if (aligned && n + 7 < p.N) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = x0[e];
                v[4 + e] = x1[e];
            }
            epilogue_row8<EPI, OUT_F32, !RS, Q8>(p, m, n, v, (!OUT_F32 && p.colsum) ? cacc : nullptr, q8_m);
        } else {
            float q0[4] = {x0[0], x0[1], x0[2], x0[3]}, q1[4] = {x1[0], x1[1], x1[2], x1[3]};
            const bool vec_ok = ((p.ldc & 3) == 0) && ((p.ld_aux_in & 3) == 0) && ((p.ld_aux_out & 3) == 0);
            nt_epilogue_quad<EPI, OUT_F32, !RS>(p, m, n, q0, vec_ok);
            if (n + 4 < p.N) nt_epilogue_quad<EPI, OUT_F32, !RS>(p, m, n + 4, q1, vec_ok);
        }
    }
    // (the host only passes `colsum` to this path when N % 8 == 0 and everything is 16-B aligned: row8 branch above)
    if (!OUT_F32 && p.colsum && n_base + 64 <= p.N) colsum_flush(p.colsum, n_base, lane, cacc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Tile epilogue for INTERIOR tiles of the persistent kernels (every row/column valid, 16-B aligned operands): the
// same LDS slab staging as slab_epilogue, but nothing in it waits on a fresh global load.  Measured on the old path
// (profiles/r01_f_epilogue_latency.txt): the epilogue's instructions are ~free, its HBM traffic is modest -- what
// cost as much as the whole K=768 main loop was LATENCY: `bias` was re-loaded after every store (may alias C) and
// each residual / pre-activation row segment was loaded right where it was consumed, 16 dependent round trips per
// wave per tile.  Here bias is read once per tile and the auxiliary operand runs PF passes (1 KiB each) ahead.
// FLAG: the run-time option of the GELU epilogue (the pre-activation output exists) and of the RESIDUAL epilogue (dropout is on)
// resolved at compile time, so that their 16 passes are straight-line code (round 4, profiles/r04_l_epilogue_flags.txt: proj -4 %,
// fc2 -2.5 %; doing the same for the column sums of the NONE / DGELU epilogues made THOSE kernels 5-19 % slower -- two copies of
// the epilogue in a kernel whose main loop is register-tight -- and was taken back).
// sum over the 8 lanes that share a row of a 64-column slab (lane & 7 = column octet): two quad permutes and the half-row mirror, no LDS crossbar
__device__ __forceinline__ float octet_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

// LNF (RESIDUAL epilogue of the ping-pong kernel only): besides storing its rows the pass leaves, per row and 64-column segment, (sum, M2 =
// sum of squared deviations from the segment's own mean) of the STORED bf16 values in p.ln_part -- the inputs of the row block's LayerNorm, which
// the last of the block's three tile columns to finish performs (gemm_p8.inc).  Segment statistics combine exactly (Chan et al.), so the variance
// does not suffer the cancellation of a sum-of-squares formula when a row's mean is large against its spread.
// Q8 (round 6): 0 = off; bits 0-1: 1 = e4m3, 2 = e5m2 copy of the bf16-rounded output to p.q8_out (scaled by q8_s); bit 2: the bf16 output itself is NOT stored
// (its only consumers read the 8-bit copy).  q8_amx: this lane's running max|output| (the caller reduces it).
template <int EPI, bool OUT_F32, int FM = 2, int FN = 4, int PF = 8, bool RS = false, bool FLAG = false, bool LNF = false, int Q8 = 0>
__device__ __forceinline__ void fast_tile_epilogue(const GemmNTArgs& p, f32x16 (&acc)[FM][FN], char* slab, int m_base,
                                                   int n_base, int lane, float q8_s = 1.f, float* q8_amx = nullptr) {
    static_assert(FM * FN == 8, "a wave owns 8 accumulators = 4 slabs of 32 x 64");
    constexpr int NFP = FN / 2;                          // 64-column slabs per 32-row block
    constexpr bool HAS_AUX = (EPI == MERLOT_EPI_RESIDUAL) || (EPI == MERLOT_EPI_DGELU);
    // PF = aux prefetch distance in passes (4 VGPRs each)
    const int hi = lane >> 5, row = lane & 31;
    const int rr = lane >> 3, c0 = (lane & 7) * 8, ch = 2 * (lane & 7);
    f32x4 bias_r[NFP][2];
#pragma unroll
    for (int fp = 0; fp < NFP; ++fp) {
        if (p.bias) {
            bias_r[fp][0] = *reinterpret_cast<const f32x4*>(p.bias + n_base + fp * 64 + c0);
            bias_r[fp][1] = *reinterpret_cast<const f32x4*>(p.bias + n_base + fp * 64 + c0 + 4);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_r[fp][0][e] = bias_r[fp][1][e] = 0.f;
        }
    }
    // pass q = slab * 4 + ps, slab = fi * NFP + fp: rows m_base + fi*32 + ps*8 + rr, columns n_base + fp*64 + c0 .. +7.
    // (Round 5: every stream's address = ONE per-lane pointer formed here + a wave-uniform offset per pass.  As `base + (int64_t) m * ld + n` per pass
    // the compiler rebuilt the 64-bit product in vector registers each time: a v_mul_lo, a v_mad_u64_u32 and three 64-bit adds per stream and pass.)
    const int64_t lane_row = (int64_t)(m_base + rr);
    const bf16* const aux_lane = HAS_AUX ? p.aux_in + lane_row * p.ld_aux_in + (n_base + c0) : nullptr;
    bf16* const auxo_lane = (EPI == MERLOT_EPI_GELU && FLAG) ? p.aux_out + lane_row * p.ld_aux_out + (n_base + c0) : nullptr;
    char* const c_lane = reinterpret_cast<char*>(p.C) + (lane_row * p.ldc + (n_base + c0)) * (OUT_F32 ? 4 : 2);
    // (one 32-bit lane offset against the uniform base: a 64-bit lane pointer more tipped this epilogue, already at 250 registers beside the 128 accumulators, into
    // 100 dwords of scratch per lane -- the first version's GELU' launch ran 39 % slower for it, profiles/r06_n_f8_producers.txt)
    const uint32_t q8_off = Q8 ? (uint32_t)(lane_row * p.ld_q8 + (n_base + c0)) : 0u;
    float q8_m = 0.f;
    auto aux_addr = [&](int q) {
        const int sl = q >> 2, ps = q & 3;
        return aux_lane + ((int64_t)((sl / NFP) * 32 + ps * 8) * p.ld_aux_in + (sl % NFP) * 64);
    };
    bf16x8 aux[PF];
    if (HAS_AUX) {
#pragma unroll
        for (int q = 0; q < PF; ++q) aux[q] = *reinterpret_cast<const bf16x8*>(aux_addr(q));
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch up here (the scheduler sinks loads)
    }
    const bool want_cs = !OUT_F32 && p.colsum != nullptr;       // wave-uniform (NONE / DGELU: stays a run-time test, see FLAG)
    float cacc[NFP][8];
#pragma unroll
    for (int fp = 0; fp < NFP; ++fp)
#pragma unroll
        for (int e = 0; e < 8; ++e) cacc[fp][e] = 0.f;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const int fi = sl / NFP, fp = sl % NFP;
        float ra = p.alpha;                              // RS (fp8 operands): times this row's dequantisation factor
        if (RS) {
            if (p.row_scale) ra *= p.row_scale[m_base + fi * 32 + row];
        }
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = RS ? acc[fi][2 * fp + fj][4 * q4 + e] * ra : acc[fi][2 * fp + fj][4 * q4 + e];   // (!RS: alpha joins the bias below, one FMA)
                const int chunk = fj * 8 + 2 * q4 + hi;
                *reinterpret_cast<f32x4*>(slab + row * 256 + ((chunk ^ (row & 15)) << 4)) = t;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int q = sl * 4 + ps;
            const int r = ps * 8 + rr;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(slab + r * 256 + ((ch ^ (r & 15)) << 4));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(slab + r * 256 + (((ch + 1) ^ (r & 15)) << 4));
            const int m = m_base + fi * 32 + r, n = n_base + fp * 64 + c0;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = RS ? x0[e] + bias_r[fp][0][e] : __builtin_fmaf(x0[e], p.alpha, bias_r[fp][0][e]);
                v[4 + e] = RS ? x1[e] + bias_r[fp][1][e] : __builtin_fmaf(x1[e], p.alpha, bias_r[fp][1][e]);
            }
            const bool no_store = DBG_BIT(p, 8), no_math = DBG_BIT(p, 128);       // experiments only
            if (EPI == MERLOT_EPI_GELU) {
                if (FLAG && !no_store) {
                    bf16x8 u8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) u8[e] = (bf16)v[e];
                    *reinterpret_cast<bf16x8*>(auxo_lane + ((int64_t)(fi * 32 + ps * 8) * p.ld_aux_out + fp * 64)) = u8;
                }
                if (!no_math) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) gelu_fast2(v[e], v[e + 1]);
                }
            } else if (EPI == MERLOT_EPI_DGELU) {
                const bf16x8 u8 = aux[q % PF];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    float g0 = (float)u8[e], g1 = (float)u8[e + 1];
                    if (!no_math) gelu_grad_fast2((float)u8[e], (float)u8[e + 1], g0, g1);
                    v[e] *= g0;
                    v[e + 1] *= g1;
                }
            } else if (EPI == MERLOT_EPI_RESIDUAL) {
                if (FLAG && !no_math) {
                    bool keep[8];                        // interior tiles: N % 256 == 0, the index is even
                    dropout_keep_n<8>(p.drop_seed, (uint64_t)(m + p.m_off) * (uint64_t)p.N + (uint64_t)n, p.drop_thresh, keep);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = keep[e] ? v[e] * p.drop_scale : 0.f;
                }
                const bf16x8 r8 = aux[q % PF];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
            }
            if (HAS_AUX && q + PF < 16) {
                aux[q % PF] = *reinterpret_cast<const bf16x8*>(aux_addr(q + PF));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (no_store) {
                if (v[0] == 12345.678f) *reinterpret_cast<float*>(p.C) = v[1];
            } else if (OUT_F32) {
                float* c = reinterpret_cast<float*>(c_lane + ((int64_t)(fi * 32 + ps * 8) * p.ldc + fp * 64) * 4);
                f32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = v[e];
                    o1[e] = v[4 + e];
                }
                *reinterpret_cast<f32x4*>(c) = o0;
                *reinterpret_cast<f32x4*>(c + 4) = o1;
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                if constexpr (!(Q8 & 4)) *reinterpret_cast<bf16x8*>(c_lane + ((int64_t)(fi * 32 + ps * 8) * p.ldc + fp * 64) * 2) = o;
                if constexpr (Q8 != 0) {
                    // (the copy is taken from the fp32 results, not from their bf16 rounding: 8 conversions less per pass, and one rounding instead of two;
                    // max3 / med3 / packed conversions: the epilogue is vector-ALU bound -- the first version of this block, 36 instructions per pass on
                    // the bf16-rounded values, made the GELU' launch 42 % slower, profiles/r06_m_f8_fuse_kernels.txt)
                    constexpr float FMAX8 = (Q8 & 3) == 1 ? 448.f : 57344.f;
                    q8_m = fmaxf(fmaxf(q8_m, fabsf(v[0])), fabsf(v[1]));
                    q8_m = fmaxf(fmaxf(q8_m, fabsf(v[2])), fabsf(v[3]));
                    q8_m = fmaxf(fmaxf(q8_m, fabsf(v[4])), fabsf(v[5]));
                    q8_m = fmaxf(fmaxf(q8_m, fabsf(v[6])), fabsf(v[7]));
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(v[e] * q8_s, -FMAX8, FMAX8);
                    int w0 = 0, w1 = 0;
                    if constexpr ((Q8 & 3) == 1) {
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
                    } else {
                        w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w0, false);
                        w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w0, true);
                        w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], w1, false);
                        w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], w1, true);
                    }
                    *reinterpret_cast<u32x2*>(p.q8_out + ((int64_t)(fi * 32 + ps * 8) * p.ld_q8 + fp * 64) + q8_off) = u32x2{(uint32_t)w0, (uint32_t)w1};
                }
                if (want_cs) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) cacc[fp][e] += (float)o[e];
                }
                if constexpr (LNF) {
                    float f[8], s1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = (float)o[e];
                        s1 += f[e];
                    }
                    s1 = octet_sum(s1);
                    const float mu = s1 * (1.0f / 64.0f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) m2 = __builtin_fmaf(f[e] - mu, f[e] - mu, m2);
                    m2 = octet_sum(m2);
                    if ((lane & 7) == 0)
                        *reinterpret_cast<f32x2*>(p.ln_part + ((int64_t)((n_base >> 6) + fp) * ((int64_t)p.ntm * 256) + m) * 2) = f32x2{s1, m2};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (want_cs) {
#pragma unroll
        for (int fp = 0; fp < NFP; ++fp) colsum_flush(p.colsum, n_base + fp * 64, lane, cacc[fp]);
    }
    if constexpr (Q8 != 0) *q8_amx = fmaxf(*q8_amx, q8_m);
}

// Tile enumeration of the persistent ping-pong kernel (gemm_p8.inc).  Row-major order makes a 32-tile XCD band 2-3 tile rows x ALL tile
// columns: with 12 columns the weight panel (12 x 393 KB at K = 768) exceeds the XCD's 4 MB L2 and is re-fetched through
// the fabric in every round of tiles.  Grouped order walks the grid column group by column group (cg columns wide, all
// rows inside a group, row-major within it): a band is ~32/cg rows x cg columns, successive bands of an XCD stay in the
// same group, and its weight slice (cg x 393 KB) stays L2-resident; activations are read once per group instead of once.
__device__ __forceinline__ void tile_coords(const GemmNTArgs& p, int tile, int& tm, int& tn) {
    if (p.cg <= 0 || p.ntn <= p.cg) {
        tm = tile / p.ntn;
        tn = tile - tm * p.ntn;
        return;
    }
    const int per_group = p.ntm * p.cg;
    const int g = tile / per_group;
    const int r = tile - g * per_group;
    const int w = min(p.cg, p.ntn - g * p.cg);
    tm = r / w;
    tn = g * p.cg + (r - tm * w);
}

// experiments (dbg & 512): per-workgroup timeline, [wg][tile-slot][0..2] = s_memtime at tile start / loop end / epilogue end
constexpr int TRACE_TILES = 32;
#ifdef MERLOT_EXPERIMENTS
__device__ long long g_persist_trace[256 * TRACE_TILES * 8];
#define PERSIST_TRACE(i, j, v) g_persist_trace[((i) * TRACE_TILES + trace_i) * 8 + (j)] = (v)
#else
#define PERSIST_TRACE(i, j, v) ((void)0)
#endif
// Tile-claim counters of the persistent ping-pong kernel live in CALLER-owned workspace (GemmNTArgs::ctr): the library
// holds no device or host state, so launches on different streams are independent as long as each uses its own block.
constexpr int64_t NT_WORKSPACE_BYTES = 64;

// LDS transpose-read of one 8-deep MFMA operand fragment (rows r..r+3 and r+4..r+7 of a 64-B-stride panel)
__device__ __forceinline__ bf16x8 tr_pair(const char* p) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * 64));
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = lo[e];
        r[4 + e] = hi4[e];
    }
    return r;
}
// the same from an absolute LDS address held in a register + a constant displacement (becomes the ds_read immediate)
__device__ __forceinline__ bf16x8 tr_pair_lds(unsigned addr, int disp) {
    typedef __attribute__((address_space(3))) bf16x4* lp_t;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(uintptr_t)(addr + (unsigned)disp));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(uintptr_t)(addr + (unsigned)(disp + 4 * 64)));
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = lo[e];
        r[4 + e] = hi4[e];
    }
    return r;
}

// ------------------------------------------------------------------------------------------------
// TN ring kernel (production weight-gradient path): ring-pipelined, big tiles, NO atomics.
// Split-R partial tiles go to a caller-owned fp32 workspace [splits][M][N] with row-contiguous stores and a small
// reduce kernel folds them into C (fp32 atomics measured ~41 G/s on this chip: 175-200 us per wgrad launch).
// ------------------------------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void gemm_tn_ring_kernel(const GemmTNArgs p, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    constexpr int BK = C::BK, S = C::STAGES;
    constexpr int PPP = BK / 16;                         // 1 KiB pieces per 32-column panel
    constexpr int PANEL_BYTES = BK * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.ntm * p.ntn;
    const int ksteps = p.R / BK;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);  // split-major: an XCD owns (nearly) all tiles of one R chunk
    const int split = wgid / ntiles;
    const int tile = wgid - split * ntiles;
    const int tile_m = tile / p.ntn;
    const int tile_n = tile - tile_m * p.ntn;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int ks = split * p.rchunk;                     // rchunk in K-steps
    const int ke = DBG_BIT(p, 2) ? ks : min(ksteps, ks + p.rchunk);
    const int nk = ke - ks;

    const bf16* a_src[C::A_PIECES];
    const bf16* b_src[C::B_PIECES];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int j = wave * C::A_PIECES + i;
        const int panel = j / PPP, rb = j % PPP;
        const int col = min(m0 + panel * 32 + (lane & 3) * 8, (int)p.lda - 8);
        a_src[i] = p.A + (int64_t)(ks * BK + rb * 16 + (lane >> 2)) * p.lda + col;
    }
#pragma unroll
    for (int i = 0; i < C::B_PIECES; ++i) {
        const int j = wave * C::B_PIECES + i;
        const int panel = j / PPP, rb = j % PPP;
        const int col = min(n0 + panel * 32 + (lane & 3) * 8, (int)p.ldb - 8);
        b_src[i] = p.B + (int64_t)(ks * BK + rb * 16 + (lane >> 2)) * p.ldb + col;
    }
    auto stage = [&](int t) {
        if DBG_BIT(p, 4) return;
        char* la = dsm + (t % S) * C::STAGE_BYTES + wave * C::A_PIECES * 1024;
        char* lb = dsm + (t % S) * C::STAGE_BYTES + C::A_BYTES + wave * C::B_PIECES * 1024;
        const int64_t r0 = (int64_t)t * BK;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[i] + r0 * p.lda), LDS_PTR(la + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(b_src[i] + r0 * p.ldb), LDS_PTR(lb + i * 1024), 16, 0, 0);
    };

    const int wm = wave / C::WN, wn = wave % C::WN;
    const int hi = lane >> 5;
    const int i16 = lane & 15;
    const int frag_off = (8 * hi + (i16 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
    const int a_off = wm * C::FM * PANEL_BYTES + frag_off;
    const int b_off = C::A_BYTES + wn * C::FN * PANEL_BYTES + frag_off;

    f32x16 acc[C::FM][C::FN];
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int t) {
        const char* base = dsm + (t % S) * C::STAGE_BYTES;
        constexpr int KK = BK / 16;
        bf16x8 af[KK][C::FM], bfr[KK][C::FN];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int f = 0; f < C::FM; ++f) af[kk][f] = tr_pair(base + a_off + f * PANEL_BYTES + kk * 16 * 64);
#pragma unroll
            for (int f = 0; f < C::FN; ++f) bfr[kk][f] = tr_pair(base + b_off + f * PANEL_BYTES + kk * 16 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int fi = 0; fi < C::FM; ++fi)
#pragma unroll
                for (int fj = 0; fj < C::FN; ++fj)
                    acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[kk][fj], af[kk][fi], acc[fi][fj], 0, 0, 0);
    };

#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) stage(t);
    const int steady = nk - (S - 1);
    int t = 0;
    for (; t < steady; ++t) {
        ring::wait_vmcnt<(S - 2) * C::LOADS>();
        __builtin_amdgcn_s_barrier();
        stage(t + S - 1);
        compute(t);
    }
    for (; t < nk; ++t) {
        ring::wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(t);
    }

    if (DBG_BIT(p, 1) && acc[0][0][0] != 12345.678f) return;
    __builtin_amdgcn_s_barrier();
    GemmNTArgs e{};                                      // reuse the row-contiguous NT epilogue (EPI_NONE, fp32 out)
    e.M = p.M; e.N = p.N; e.alpha = p.alpha;
    if (p.splits > 1) {
        e.C = ws + (int64_t)split * p.M * p.N;
        e.ldc = p.N;
        e.accumulate = 0;
    } else {
        e.C = p.C;
        e.ldc = p.ldc;
        e.accumulate = p.use_atomics;                    // == caller's accumulate flag when splits == 1
    }
    staged_epilogue<C, MERLOT_EPI_NONE, true>(e, acc, dsm + wave * C::EPI_BYTES, m0 + wm * C::FM * 32, n0 + wn * C::FN * 32,
                                              lane);
}

// C[m][n] = (accumulate ? C : 0) + sum_s ws[s][m][n]
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ c,
                                                        int64_t ldc, int M, int N, int accumulate) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4;
    const int64_t plane = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + m * N + n);
        for (int s = 1; s < splits; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ws + s * plane + m * N + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        float* cp = c + m * ldc + n;
        if (accumulate) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += o[e];
        }
        *reinterpret_cast<f32x4*>(cp) = acc;
    }
}

template <int EPI>
int launch_nt(const GemmNTArgs& a, int out_f32, hipStream_t s) {
    const int grid = a.ntm * a.ntn;
    if (out_f32)
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), dim3(grid), dim3(256), 0, s, a);
    return merlot_launch_status("merlot_gemm_bf16_nt");
}

template <typename C, int EPI, bool OUT_F32>
int launch_ring_one(GemmNTArgs& a, hipStream_t s) {
    auto kern = gemm_nt_ring_kernel<C, EPI, OUT_F32>;
    MERLOT_ENSURE_LDS(kern, C::LDS_BYTES, "merlot_gemm_bf16_nt(ring)");
    a.ntm = cdiv(a.M, C::BM);
    a.ntn = cdiv(a.N, C::BN);
    hipLaunchKernelGGL(kern, dim3(a.ntm * a.ntn), dim3(C::NT), C::LDS_BYTES, s, a);
    return merlot_launch_status("merlot_gemm_bf16_nt(ring)");
}

template <typename C>
int launch_ring(GemmNTArgs& a, int epilogue, int out_f32, hipStream_t s) {
#define RING_CASE(E)                                                            \
    case E:                                                                     \
        return out_f32 ? launch_ring_one<C, E, true>(a, s) : launch_ring_one<C, E, false>(a, s);
    switch (epilogue) {
        RING_CASE(MERLOT_EPI_NONE)
        RING_CASE(MERLOT_EPI_GELU)
        RING_CASE(MERLOT_EPI_RESIDUAL)
        RING_CASE(MERLOT_EPI_DGELU)
    }
#undef RING_CASE
    merlot_set_error("merlot_gemm_bf16_nt: unknown epilogue %d", epilogue);
    return MERLOT_ESHAPE;
}

using RingC = ring::Cfg<4, 2, 2, 4, 32, 4>;   // 256x256, BK 32, 4 stages, 128 KB, 8 waves (experiments build only)
using RingK = ring::Cfg<2, 4, 2, 2, 32, 3>;   // 128x256, BK 32, 3 stages, 72 KB (2 blocks/CU)
using RingN64 = ring::Cfg<4, 1, 2, 2, 32, 3>;   // 256x64,  4 waves, 60 KB: narrow outputs (ResNet-stem 1x1 / 3x3 with 32..64 filters)
using RingN128 = ring::Cfg<4, 1, 2, 4, 32, 3>;  // 256x128, 4 waves, 72 KB
#ifdef MERLOT_EXPERIMENTS
// Round 6 (VERDICT r5 #1b): TWO co-resident 4-wave workgroups per CU on 128 x 256 tiles -- the wave tile of the ping-pong kernel (128 x 64), BK 32,
// three stages = 72 KiB, 256 registers per wave.  One workgroup's epilogue then runs beside the other's main loop; what is given up is the
// barrier-locked pairing of the two waves of a SIMD and B's sharing between the row halves (+50 % LDS-DMA bytes per MFMA).  id 23.
struct RingQ : ring::Cfg<1, 4, 4, 2, 32, 3> { static constexpr int OCC = 2; };
#endif

// ---- LNF: the LayerNorm of one finished 256-row block of C (N = 768), run by the workgroup whose tile was the last of the block's three to arrive.
// The other two tiles' rows were stored by other CUs of the SAME XCD (the LNF tile order keeps a row block inside one XCD), so they sit in this XCD's L2:
// the reads carry sc1 (agent scope: not served from this CU's vector L1).  lds: 2 KiB for the rows' (mean, rstd).
// The LayerNorm operands of an LNF launch, read from the kernel-argument segment WHERE THEY ARE USED (gemm_p8.inc): as ordinary by-value arguments they were
// loaded at kernel entry and held in scalar registers through the main loop -- the loop then restored spilled scalars with 21 v_readlane per K-tile, +8.6 % on
// the K = 3 072 main loop (profiles/r06_e_ln_fold_trace.txt).
struct LnArgs {
    bf16* out;
    int64_t ld;
    const float* gamma;
    const float* beta;
    float* mean;
    float* rstd;
    float eps;
    float* part;
    unsigned int* ctr;
};
__device__ __forceinline__ LnArgs ln_args_from_kernarg() {
    typedef const GemmNTArgs __attribute__((address_space(4)))* kargp_t;
    kargp_t kp = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();      // the kernel's one by-value argument sits at offset 0
    asm volatile("" : "+s"(kp));                                        // (a fresh scalar load per use: nothing to keep live across the tile loop)
    LnArgs a;
    a.out = kp->ln_out; a.ld = kp->ld_ln; a.gamma = kp->ln_gamma; a.beta = kp->ln_beta; a.mean = kp->ln_mean; a.rstd = kp->ln_rstd;
    a.eps = kp->ln_eps; a.part = kp->ln_part; a.ctr = kp->ln_ctr;
    return a;
}

__device__ __forceinline__ void ln_row_block(const GemmNTArgs& p, const LnArgs& ln, int tm, char* lds, int tid_) {
    constexpr int NSEG = 12, H = 768;                      // 64-column segments of a row
    int tid = tid_;
    asm volatile("" : "+v"(tid));                          // laundered: the per-lane offsets below are formed here, not hoisted above the tile loop
    constexpr int SC1 = 16;                                // cache policy bit 4 on gfx94x / gfx950: sc1 = agent scope (compiler-visible loads: it places the waits)
    f32x2* stat = reinterpret_cast<f32x2*>(lds);
    const int64_t mpad = (int64_t)p.ntm * 256;
    const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(ln.part, 0, (int)(NSEG * mpad * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
    const int rr = tid >> 5, cc = (tid & 31) * 8;          // 16 rows x 32 column octets per pass; 16 passes per tile column
    const unsigned row_step = (unsigned)(16 * p.ldc * 2);
    u32x4 xa[16], xb[16];
    auto load_round = [&](u32x4 (&x)[16], const int tnc) {
        const unsigned src = (unsigned)((((int64_t)tm * 256 + rr) * p.ldc + tnc * 256 + cc) * 2);     // byte offset into C (< 4 GiB: p8_lnf_ok)
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rs_c, (int)(src + (unsigned)ps * row_step), 0, SC1);
            static_assert(sizeof(raw) == 16, "four dwords");
            x[ps] = __builtin_bit_cast(u32x4, raw);
        }
    };
    load_round(xa, 0);
    if (tid < 256) {
        const int64_t row = (int64_t)tm * 256 + tid;
        float s1[NSEG], q[NSEG];
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            // (the builtin's result type is a GCC-style vector: converted IMPLICITLY to an ext_vector type hipcc kept element 0 only -- every row's M2 read back
            // as its sum, rstd NaN, profiles/r06_d_ln_fold_debug.txt; hence auto + bit_cast)
            const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rs_part, (int)(((int64_t)g * mpad + row) * 8), 0, SC1);
            static_assert(sizeof(raw) == 8, "two dwords");
            const f32x2 sq = __builtin_bit_cast(f32x2, raw);
            s1[g] = sq[0];
            q[g] = sq[1];
        }
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) tot += s1[g];
        const float mean = tot * (1.0f / H);
        float m2 = 0.f;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            const float d = s1[g] * (1.0f / 64.0f) - mean;
            m2 += q[g] + 64.0f * d * d;
        }
        const float rstd = rsqrtf(m2 * (1.0f / H) + ln.eps);
        stat[tid] = f32x2{mean, rstd};
        if (ln.mean) ln.mean[row] = mean;
        if (ln.rstd) ln.rstd[row] = rstd;
    }
    // (phase B's first reads do not depend on the statistics: they were issued above, in front of this barrier)
    __syncthreads();
    // Three rounds (tile columns) of 16 passes; round r + 1's sixteen 16-byte reads are in flight while round r is normalised and stored (the first version
    // waited for each round's reads, then stored, then read again: 80 k cycles per row block, profiles/r06_d_ln_fold_trace.txt)
    auto process = [&](const u32x4 (&x)[16], const int tnc) {
        const int col = tnc * 256 + cc;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln.gamma + col), g1 = *reinterpret_cast<const f32x4*>(ln.gamma + col + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln.beta + col), b1 = *reinterpret_cast<const f32x4*>(ln.beta + col + 4);
        bf16* dst = ln.out + ((int64_t)tm * 256 + rr) * ln.ld + col;
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const f32x2 st = stat[ps * 16 + rr];             // (mean, rstd)
            const float mr = -st[0] * st[1];
            const bf16x8 xv = __builtin_bit_cast(bf16x8, x[ps]);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = __builtin_fmaf((float)xv[e], st[1], mr);                                   // (x - mean) rstd
                o[e] = (bf16)__builtin_fmaf(t, e < 4 ? g0[e & 3] : g1[e & 3], e < 4 ? b0[e & 3] : b1[e & 3]);
            }
            *reinterpret_cast<bf16x8*>(dst + (int64_t)ps * 16 * ln.ld) = o;
        }
    };
    load_round(xb, 1);
    process(xa, 0);
    load_round(xa, 2);
    process(xb, 1);
    process(xa, 2);
}

#include "gemm_p8.inc"

// Which kernel a shape runs on (also exported as merlot_gemm_bf16_nt_plan so tests can assert it).  From the sweeps in
// profiles/r01_gemm_tile_sweep.txt: the persistent 256x256 kernel wins whenever its rounds of 256 workgroups are well
// filled (>= 85 % of the slots of its last round included); otherwise 128x256 tiles with two co-resident workgroups
// per CU absorb the ragged tail better.
int nt_plan(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = (int64_t)cdiv(M, 256) * cdiv(N, 256);
    const int64_t rounds = (tiles + 255) / 256;
    // well-filled rounds of 256 workgroups: the ping-pong persistent kernel (K-tiles of 64, at least 2 per tile; round 2:
    // +5..13 % over the lock-step persistent kernel on every shape of the step, profiles/r02_p8_ab.txt)
    int cfg = (tiles * 100 >= rounds * 256 * 85) ? MERLOT_NT_KERNEL_P8 : MERLOT_NT_KERNEL_RING_128x256;
    if (cfg == MERLOT_NT_KERNEL_P8 && K < 128) cfg = MERLOT_NT_KERNEL_RING_128x256;   // one K-tile of 64: the ring kernel (BK 32)
    // (round 1 sent every [T, 768] x [768, 768] launch to the 128x256 ring kernel; the ping-pong kernel wins those too once
    // its rounds are filled: 166 vs 200 us at T = 101376, 70 vs 78 at 41984, neutral at 16384 = 75 % of one round,
    // which the fill rule above already routes to the ring kernel -- profiles/r02_a_p8_vs_ring_768.txt)
    // narrow outputs (the ResNet-stem convolutions with 32..128 filters): a 256-wide tile would compute 2-8x the
    // columns that exist; these launches are HBM-bound and reach ~5 TB/s on 256x64 / 256x128 tiles
    if (N <= 64 || (N <= 128 && K <= 512)) cfg = MERLOT_NT_KERNEL_RING_256x64;
    else if (N <= 128) cfg = MERLOT_NT_KERNEL_RING_256x128;
    return cfg;
}

int gemm_nt_dispatch(GemmNTArgs& a, int epilogue, int out_f32, hipStream_t s) {
    int cfg = nt_plan(a.M, a.N, a.K);
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_DBG")) a.dbg = atoi(e);
    if (const char* e = getenv("MERLOT_NT_CFG_DYN")) cfg = atoi(e);
#endif
    if (cfg == MERLOT_NT_KERNEL_P8 && !p8_ok(a)) {
        // The ping-pong kernel addresses its operands with unsigned 32-bit byte offsets.  An A operand of 4 GiB or more is cut into
        // row ranges of whole tiles that fit (round 4; rounds 1-3 kept a second persistent kernel with 64-bit addressing for this):
        // the same kernel, the same tiles, launched back to back on the stream -- the claim counters are zero again when a launch
        // ends --, `m_off` keeps the dropout mask's element index global.
        MERLOT_CHECK(a.K % 64 == 0 && a.K >= 128 && ((int64_t)a.N + 256) * a.ldb * 2 < (1LL << 32), MERLOT_ESHAPE,
                     "merlot_gemm_bf16_nt: no kernel for M=%d N=%d K=%d ldb=%lld", a.M, a.N, a.K, (long long)a.ldb);
        MERLOT_CHECK(a.ctr != nullptr, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: this shape needs the caller's zeroed workspace");
        const int64_t rows_max = ((((1LL << 32) - 1) / (a.lda * 2)) - 256) / 256 * 256;
        MERLOT_CHECK(rows_max >= 256, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: lda=%lld is too large for the tiled kernels", (long long)a.lda);
        const int csz = out_f32 ? 4 : 2;
        for (int64_t r0 = 0; r0 < a.M; r0 += rows_max) {
            GemmNTArgs c = a;
            c.M = (int)(a.M - r0 < rows_max ? a.M - r0 : rows_max);
            c.m_off = a.m_off + r0;
            c.A = a.A + r0 * a.lda;
            c.C = (char*)a.C + r0 * a.ldc * csz;
            if (a.aux_in) c.aux_in = a.aux_in + r0 * a.ld_aux_in;
            if (a.aux_out) c.aux_out = a.aux_out + r0 * a.ld_aux_out;
            if (a.row_scale) c.row_scale = a.row_scale + r0;
            const int rc = gemm_nt_dispatch(c, epilogue, out_f32, s);
            if (rc != MERLOT_OK) return rc;
        }
        return MERLOT_OK;
    }
    if (cfg == MERLOT_NT_KERNEL_P8)
        MERLOT_CHECK(a.ctr != nullptr, MERLOT_ESHAPE,
                     "merlot_gemm_bf16_nt: this shape runs a persistent kernel with dynamic tile claims and needs the caller's "
                     "zeroed workspace (merlot_gemm_nt_workspace_bytes() bytes, one block per concurrently used stream)");
    // fused column sums of C (bias gradient): in the ping-pong kernel's epilogue when its row-contiguous path applies,
    // otherwise the stand-alone column-sum kernel right behind the GEMM (same stream, same result up to summation order)
    float* const colsum = a.colsum;
    // (N % 64: the epilogue flushes its per-lane column accumulators per whole 64-column slab; a ragged last slab would be dropped)
    const bool cs_fused = colsum && cfg == MERLOT_NT_KERNEL_P8 && !out_f32 && (a.N % 64 == 0) && (a.ldc % 8 == 0) &&
                          (a.ld_aux_in % 8 == 0) && (a.ld_aux_out % 8 == 0) && (((uintptr_t)a.C | (uintptr_t)colsum) & 15) == 0;
    if (!cs_fused) a.colsum = nullptr;
    if (colsum && !cs_fused) {
        const int rc = gemm_nt_dispatch(a, epilogue, out_f32, s);
        if (rc != MERLOT_OK) return rc;
        return merlot_colsum_bf16(a.C, a.ldc, colsum, a.M, a.N, 1, s);
    }
    switch (cfg) {
        case MERLOT_NT_KERNEL_RING_128x256: return launch_ring<RingK>(a, epilogue, out_f32, s);
        case MERLOT_NT_KERNEL_RING_256x64: return launch_ring<RingN64>(a, epilogue, out_f32, s);
        case MERLOT_NT_KERNEL_RING_256x128: return launch_ring<RingN128>(a, epilogue, out_f32, s);
        case MERLOT_NT_KERNEL_P8: return launch_p8(a, epilogue, out_f32, s);
#ifdef MERLOT_EXPERIMENTS
        case 3: return launch_ring<RingC>(a, epilogue, out_f32, s);
        case 23: return launch_ring<RingQ>(a, epilogue, out_f32, s);
#endif
        default: break;
    }
#ifndef MERLOT_EXPERIMENTS
    merlot_set_error("merlot_gemm_bf16_nt: no kernel for plan %d", cfg);
    return MERLOT_ESHAPE;
#else
    a.ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.N, BN);
    switch (epilogue) {
        case MERLOT_EPI_NONE: return launch_nt<MERLOT_EPI_NONE>(a, out_f32, s);
        case MERLOT_EPI_GELU: return launch_nt<MERLOT_EPI_GELU>(a, out_f32, s);
        case MERLOT_EPI_RESIDUAL: return launch_nt<MERLOT_EPI_RESIDUAL>(a, out_f32, s);
        case MERLOT_EPI_DGELU: return launch_nt<MERLOT_EPI_DGELU>(a, out_f32, s);
    }
    merlot_set_error("merlot_gemm_bf16_nt: unknown epilogue %d", epilogue);
    return MERLOT_ESHAPE;
#endif
}

int tn_launch(GemmTNArgs& a, int accumulate, hipStream_t s) {
    a.ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.N, BN);
    const int tiles = a.ntm * a.ntn;
    int splits = cdiv(512, tiles);
    const int max_splits = (a.R + 511) / 512;
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
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(256), 0, s, a);
    return merlot_launch_status("merlot_gemm_bf16_tn");
}

using TnRingC = ring::Cfg<4, 2, 2, 4, 32, 4>;   // 256x256, BK 32, 4 stages, 8 waves, 1 workgroup / CU
using TnRingK = ring::Cfg<2, 4, 2, 2, 32, 3>;   // 128x256, BK 32, 3 stages, 8 waves, 2 workgroups / CU

int tn_config() {
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_TN_CFG")) return atoi(e);
#endif
    return 1;                                            // 128x256, 2 WG/CU: +5-7 % over 256x256 on the wgrad shapes
}

// split plan of the ring TN kernel (shared with the workspace-size query)
struct TnPlan {
    int ntm, ntn, splits, chunk;
};
TnPlan tn_plan(int64_t M, int64_t N, int64_t R) {
    const bool k = tn_config() == 1;
    const int bm = k ? TnRingK::BM : TnRingC::BM, bn = k ? TnRingK::BN : TnRingC::BN;
    const int slots = k ? 512 : 256;
    TnPlan pl;
    pl.ntm = cdiv(M, bm);
    pl.ntn = cdiv(N, bn);
    const int tiles = pl.ntm * pl.ntn;
    const int ksteps = (int)(R / TnRingC::BK);
    int splits = slots / tiles;                          // one round of resident workgroups
    if (splits < 1) splits = 1;
    if (splits > ksteps / 8) splits = ksteps / 8 > 0 ? ksteps / 8 : 1;
#ifdef MERLOT_EXPERIMENTS
    if (const char* env = getenv("MERLOT_TN_SPLITS")) {
        const int v = atoi(env);
        if (v > 0) splits = v > ksteps ? ksteps : v;
    }
#endif
    pl.chunk = (ksteps + splits - 1) / splits;
    pl.splits = (ksteps + pl.chunk - 1) / pl.chunk;
    return pl;
}

template <typename C>
int tn_ring_launch_cfg(GemmTNArgs& a, const TnPlan& pl, float* ws, hipStream_t s) {
    auto kern = gemm_tn_ring_kernel<C>;
    MERLOT_ENSURE_LDS(kern, C::LDS_BYTES, "merlot_gemm_bf16_tn(ring)");
    hipLaunchKernelGGL(kern, dim3(pl.ntm * pl.ntn * pl.splits), dim3(C::NT), C::LDS_BYTES, s, a, ws);
    return MERLOT_OK;
}

int tn_ring_launch(GemmTNArgs& a, int accumulate, float* ws, int64_t ws_bytes, hipStream_t s) {
    const TnPlan pl = tn_plan(a.M, a.N, a.R);
    a.ntm = pl.ntm; a.ntn = pl.ntn; a.splits = pl.splits; a.rchunk = pl.chunk;
    a.use_atomics = accumulate;                          // meaning here: accumulate into C when splits == 1
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_DBG")) a.dbg = atoi(e);
#endif
    if (pl.splits > 1) {
        const int64_t need = (int64_t)pl.splits * a.M * a.N * 4;
        MERLOT_CHECK(ws != nullptr && ws_bytes >= need, MERLOT_ESHAPE,
                     "merlot_gemm_bf16_tn: workspace too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)need);
    }
    int rc = tn_config() == 1 ? tn_ring_launch_cfg<TnRingK>(a, pl, ws, s) : tn_ring_launch_cfg<TnRingC>(a, pl, ws, s);
    if (rc != MERLOT_OK) return rc;
    if (pl.splits > 1) {
        int64_t total = (int64_t)a.M * (a.N / 4);
        int grid = (int)((total + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(grid), dim3(256), 0, s, ws, pl.splits, a.C, a.ldc, a.M, a.N, accumulate);
    }
    return merlot_launch_status("merlot_gemm_bf16_tn(ring)");
}

bool tn_ring_ok(const GemmTNArgs& a) {
    const auto pad8 = [](int64_t x) { return (x + 7) / 8 * 8; };
    return a.R >= 8 * TnRingC::BK && (a.lda % 8 == 0) && (a.ldb % 8 == 0) && a.lda >= pad8(a.M) && a.ldb >= pad8(a.N) &&
           (a.N % 8 == 0) && (a.ldc % 4 == 0) && (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C) & 15) == 0;
}

// ---- TN ping-pong kernel (gemm_tn_p8_kernel): 256x256 tiles x reduction chunks = one round of workgroups
struct TnP8Plan {
    int ntm, ntn, splits, chunk;                         // chunk in K-tiles of 64 reduction rows
};
bool tn_p8_shape(int64_t M, int64_t N, int64_t R) {      // pure function of the shape (the workspace query uses it too)
    const int64_t tiles = (int64_t)cdiv(M, 256) * cdiv(N, 256);
    return R >= 4096 && tiles <= 256 && M >= 128 && N >= 128;
}
TnP8Plan tn_p8_plan(int64_t M, int64_t N, int64_t R) {
    TnP8Plan pl;
    pl.ntm = cdiv(M, 256);
    pl.ntn = cdiv(N, 256);
    const int tiles = pl.ntm * pl.ntn;
    const int nkt = (int)(R / 64);
    int splits = 256 / tiles;
    if (splits > nkt / 16) splits = nkt / 16;            // >= 16 K-tiles per chunk: prologue + epilogue stay small
    if (splits < 1) splits = 1;
    pl.chunk = (nkt + splits - 1) / splits;
    pl.splits = (nkt + pl.chunk - 1) / pl.chunk;
    return pl;
}
int tn_p8_launch(GemmTNArgs& a, int accumulate, float* ws, int64_t ws_bytes, hipStream_t s) {
    const TnP8Plan pl = tn_p8_plan(a.M, a.N, a.R);
    a.ntm = pl.ntm; a.ntn = pl.ntn; a.splits = pl.splits; a.rchunk = pl.chunk;
    a.use_atomics = accumulate;                          // meaning here: accumulate into C when splits == 1
    if (pl.splits > 1) {
        const int64_t need = (int64_t)pl.splits * a.M * a.N * 4;
        MERLOT_CHECK(ws != nullptr && ws_bytes >= need, MERLOT_ESHAPE,
                     "merlot_gemm_bf16_tn: workspace too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)need);
    }
    // ONE phase per K-tile (gemm_tn_p1_kernel, round 3).  profiles/r03_a_ph2.txt: four -> two phases +20-25 % on every
    // weight-gradient shape of the step, two -> one another +3-5 % (1 230 TFLOP/s = 0.49 of peak on dW1 / dW2); the two- and
    // four-phase bodies of gemm_tn_p8_kernel remain in the experiments build (MERLOT_TN_PH2 = 1 / 0).
    int ph = 1;
    MERLOT_ENSURE_LDS(gemm_tn_p1_kernel<false>, P8_LDS, "merlot_gemm_bf16_tn(p1)");
    MERLOT_ENSURE_LDS(gemm_tn_p1_kernel<true>, P8_LDS, "merlot_gemm_bf16_tn(p1)");
#ifdef MERLOT_EXPERIMENTS
    MERLOT_ENSURE_LDS(gemm_tn_p8_kernel<true>, P8_LDS, "merlot_gemm_bf16_tn(p8)");
    MERLOT_ENSURE_LDS(gemm_tn_p8_kernel<false>, P8_LDS, "merlot_gemm_bf16_tn(p8)");
    if (const char* e = getenv("MERLOT_TN_PH2")) ph = atoi(e) == 0 ? 4 : atoi(e) == 1 ? 2 : 1;
    if (ph == 4) hipLaunchKernelGGL(gemm_tn_p8_kernel<false>, dim3(pl.ntm * pl.ntn * pl.splits), dim3(512), P8_LDS, s, a, ws);
    if (ph == 2) hipLaunchKernelGGL(gemm_tn_p8_kernel<true>, dim3(pl.ntm * pl.ntn * pl.splits), dim3(512), P8_LDS, s, a, ws);
#endif
    if (ph == 1 && a.colsum_a) hipLaunchKernelGGL(gemm_tn_p1_kernel<true>, dim3(pl.ntm * pl.ntn * pl.splits), dim3(512), P8_LDS, s, a, ws);
    else if (ph == 1) hipLaunchKernelGGL(gemm_tn_p1_kernel<false>, dim3(pl.ntm * pl.ntn * pl.splits), dim3(512), P8_LDS, s, a, ws);
    if (pl.splits > 1) {
        int64_t total = (int64_t)a.M * (a.N / 4);
        int grid = (int)((total + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(grid), dim3(256), 0, s, ws, pl.splits, a.C, a.ldc, a.M, a.N, accumulate);
    }
    return merlot_launch_status("merlot_gemm_bf16_tn(p8)");
}

#include "gemm_q8.inc"

int gemm_tn_dispatch_(GemmTNArgs& a, int accumulate, float* ws, int64_t ws_bytes, hipStream_t s, int64_t& cs_rows_done);
// colsum_a (the column sums of A's first colsum_m columns, accumulated): from the one-phase ping-pong kernel's own fragments for the reduction rows that
// kernel covers, from merlot_colsum_bf16 (same stream, right behind) for the rest -- every row when another kernel takes the shape
int gemm_tn_dispatch(GemmTNArgs& a, int accumulate, float* ws, int64_t ws_bytes, hipStream_t s) {
    int64_t cs_rows_done = 0;
    const int rc = gemm_tn_dispatch_(a, accumulate, ws, ws_bytes, s, cs_rows_done);
    if (rc != MERLOT_OK || !a.colsum_a || cs_rows_done == a.R) return rc;
    return merlot_colsum_bf16(a.A + cs_rows_done * a.lda, a.lda, a.colsum_a, a.R - cs_rows_done, a.colsum_m, 1, s);
}
int gemm_tn_dispatch_(GemmTNArgs& a, int accumulate, float* ws, int64_t ws_bytes, hipStream_t s, int64_t& cs_rows_done) {
    if (!tn_ring_ok(a)) return tn_launch(a, accumulate, s);
    int tn_kernel = (tn_p8_shape(a.M, a.N, a.R / 64 * 64) && ((int64_t)a.R + 64) * a.lda * 2 < (1LL << 32) &&
                     ((int64_t)a.R + 64) * a.ldb * 2 < (1LL << 32)) ? 1 : 0;     // unsigned 32-bit byte offsets
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_TN_P8")) tn_kernel = atoi(e);
#endif
    if (tn_kernel) {
        const int R = a.R;
        const int r_main = R / 64 * 64;
        GemmTNArgs m = a;
        m.R = r_main;
        int rc = tn_p8_launch(m, accumulate, ws, ws_bytes, s);
#ifdef MERLOT_EXPERIMENTS
        if (!getenv("MERLOT_TN_PH2"))                    // (the two- and four-phase bodies of the experiments build do not sum A)
#endif
            if (a.colsum_a) cs_rows_done = r_main;
        if (rc != MERLOT_OK || r_main == R) return rc;
        GemmTNArgs t = a;                                // < 64 reduction rows left: the register-staged kernel adds them
        t.A = a.A + (int64_t)r_main * a.lda;
        t.B = a.B + (int64_t)r_main * a.ldb;
        t.R = R - r_main;
        return tn_launch(t, 1, s);
    }
    // main part: the first floor(R/32)*32 reduction rows through the ring kernel; the (< 32 row) tail, if any,
    // through the register-staged kernel accumulating on top.
    const int R = a.R;
    const int r_main = R / TnRingC::BK * TnRingC::BK;
    GemmTNArgs m = a;
    m.R = r_main;
    int rc = tn_ring_launch(m, accumulate, ws, ws_bytes, s);
    if (rc != MERLOT_OK || r_main == R) return rc;
    GemmTNArgs t = a;
    t.A = a.A + (int64_t)r_main * a.lda;
    t.B = a.B + (int64_t)r_main * a.ldb;
    t.R = R - r_main;
    return tn_launch(t, 1, s);
}

}  // namespace

// A persistent launch that FAILED may have left the caller's tile-claim counters non-zero ("zero on entry, left zero" is the kernels' whole protocol -- they carry no
// launch epoch): the block is cleared on the same stream before the failure is reported (the attention entries do the same, attention.hip)
static inline int nt_status(int rc, void* workspace, hipStream_t s) {
    if (rc == MERLOT_ELAUNCH && workspace) (void)hipMemsetAsync(workspace, 0, (size_t)NT_WORKSPACE_BYTES, s);
    return rc;
}

extern "C" int merlot_gemm_bf16_nt(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc,
                                   int64_t M, int64_t N, int64_t K, float alpha, int epilogue, int out_f32,
                                   int accumulate, const float* bias, const void* aux_in, int64_t ld_aux_in,
                                   void* aux_out, int64_t ld_aux_out, float dropout_p, uint64_t dropout_seed,
                                   float* colsum_out, void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(A && Bt && C, MERLOT_ESHAPE, "merlot_gemm_bf16_nt: null operand");
    MERLOT_CHECK(!workspace || (workspace_bytes >= NT_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0), MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt: workspace of %lld bytes, need %lld", (long long)workspace_bytes, (long long)NT_WORKSPACE_BYTES);
    MERLOT_CHECK(!(colsum_out && out_f32), MERLOT_EDTYPE, "merlot_gemm_bf16_nt: colsum_out needs a bf16 output");
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
    a.colsum = colsum_out;
    a.ctr = (unsigned int*)workspace;
    return nt_status(gemm_nt_dispatch(a, epilogue, out_f32, (hipStream_t)stream), workspace, (hipStream_t)stream);
}

// h' = aux_in + dropout(alpha * A Bt^T + bias) AND LayerNorm(h') from one launch (ABI v8; gemm_p8.inc "LNF").  Shapes the fused kernel does not take
// (N != 768, fewer than 96 row blocks, unaligned operands) and the rows of a ragged last row block run the same GEMM and the stand-alone LayerNorm kernel.
extern "C" int merlot_ln_fwd(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean, float* rstd,
                             int64_t rows, int H, float eps, merlot_stream_t stream);
static int64_t ln_ws_ctr_bytes(int64_t M) { return (cdiv(M, 256) * 4 + 255) / 256 * 256; }
extern "C" int64_t merlot_gemm_nt_ln_workspace_bytes(int64_t M, int64_t N) {
    return ln_ws_ctr_bytes(M) + (N / 64) * (int64_t)cdiv(M, 256) * 256 * 8;
}
extern "C" int merlot_gemm_bf16_nt_ln_plan(int64_t M, int64_t N, int64_t K) {     // 1: the fused kernel takes (aligned operands of) this shape
    return N == 768 && M >= 96 * 256 && nt_plan(M, N, K) == MERLOT_NT_KERNEL_P8 && K % 64 == 0 && K >= 128 ? 1 : 0;
}
extern "C" int merlot_gemm_bf16_nt_ln(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      float alpha, const float* bias, const void* aux_in, int64_t ld_aux_in, float dropout_p, uint64_t dropout_seed,
                                      const float* ln_gamma, const float* ln_beta, void* ln_out, int64_t ld_ln, float* ln_mean, float* ln_rstd,
                                      float ln_eps, void* ln_workspace, int64_t ln_workspace_bytes, void* workspace, int64_t workspace_bytes,
                                      merlot_stream_t stream) {
    MERLOT_CHECK(ln_gamma && ln_beta && ln_out, MERLOT_ESHAPE, "merlot_gemm_bf16_nt_ln: null LayerNorm operand");
    MERLOT_CHECK(N % 256 == 0 && N <= 2048 && ld_ln == N && ldc == N, MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt_ln: C and ln_out must be dense [M, N] with N a multiple of 256 (N=%lld ldc=%lld ld_ln=%lld)", (long long)N,
                 (long long)ldc, (long long)ld_ln);
    GemmNTArgs probe{};
    probe.A = (const bf16*)A; probe.B = (const bf16*)Bt; probe.C = C; probe.lda = lda; probe.ldb = ldb; probe.ldc = ldc;
    probe.M = (int)M; probe.N = (int)N; probe.K = (int)K; probe.aux_in = (const bf16*)aux_in; probe.ld_aux_in = ld_aux_in; probe.bias = bias;
    probe.ln_out = (bf16*)ln_out; probe.ld_ln = ld_ln; probe.ln_gamma = ln_gamma; probe.ln_beta = ln_beta;
    const bool fused = M > 0 && M < (1LL << 31) && K > 0 && merlot_gemm_bf16_nt_ln_plan(M, N, K) && p8_lnf_ok(probe) && ln_workspace != nullptr &&
                       ln_workspace_bytes >= merlot_gemm_nt_ln_workspace_bytes(M, N) && ((uintptr_t)ln_workspace & 15) == 0;
    if (!fused) {
        const int rc = merlot_gemm_bf16_nt(A, lda, Bt, ldb, C, ldc, M, N, K, alpha, MERLOT_EPI_RESIDUAL, 0, 0, bias, aux_in, ld_aux_in, nullptr, 0,
                                           dropout_p, dropout_seed, nullptr, workspace, workspace_bytes, stream);
        if (rc != MERLOT_OK) return rc;
        return merlot_ln_fwd(C, 0, ln_gamma, ln_beta, ln_out, nullptr, ln_mean, ln_rstd, M, (int)N, ln_eps, stream);
    }
    MERLOT_CHECK(A && Bt && C && aux_in, MERLOT_ESHAPE, "merlot_gemm_bf16_nt_ln: null operand");
    MERLOT_CHECK(workspace && workspace_bytes >= NT_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0, MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt_ln: needs the caller's zeroed workspace of merlot_gemm_nt_workspace_bytes() bytes");
    MERLOT_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)Bt & 15) == 0, MERLOT_EALIGN,
                 "merlot_gemm_bf16_nt_ln: A / Bt must be 16-byte aligned with leading dimensions in multiples of 8");
    MERLOT_CHECK(dropout_p >= 0.f && dropout_p < 1.f, MERLOT_ESHAPE, "merlot_gemm_bf16_nt_ln: dropout_p out of range");
    GemmNTArgs a = probe;
    a.alpha = alpha;
    a.drop_thresh = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    a.drop_scale = 1.0f / (1.0f - dropout_p);
    a.drop_seed = dropout_seed;
    a.ctr = (unsigned int*)workspace;
    a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_eps = ln_eps;
    a.ln_ctr = (unsigned int*)ln_workspace;
    a.ln_part = (float*)((char*)ln_workspace + ln_ws_ctr_bytes(M));
    int rc = nt_status(gemm_nt_dispatch(a, MERLOT_EPI_RESIDUAL, 0, (hipStream_t)stream), workspace, (hipStream_t)stream);
    if (rc != MERLOT_OK) {
        (void)hipMemsetAsync(ln_workspace, 0, (size_t)ln_ws_ctr_bytes(M), (hipStream_t)stream);      // arrival counters a failed launch may have left
        return rc;
    }
    const int64_t done = M / 256 * 256;                   // whole row blocks: normalised by the GEMM launch; the ragged rest by the LayerNorm kernel
    if (done < M)
        rc = merlot_ln_fwd((const bf16*)C + done * ldc, 0, ln_gamma, ln_beta, (bf16*)ln_out + done * ld_ln, nullptr, ln_mean ? ln_mean + done : nullptr,
                           ln_rstd ? ln_rstd + done : nullptr, M - done, (int)N, ln_eps, stream);
    return rc;
}

// C = epilogue(alpha * scale_a[0] * scale_b[0] * A8 * B8^T + bias): e4m3 operands (merlot_quantize_e4m3), fp32 accumulation on the
// MX-scaled MFMA with unit block scales, the bf16 kernel's epilogues.  ONE kernel (the 256 x 256 ping-pong kernel) -- shapes
// it cannot take are an error, not a fallback.
extern "C" int merlot_gemm_fp8_nt(const void* A8, int64_t lda, const float* scale_a, const float* row_scale_a, const void* B8t, int64_t ldb,
                                  const float* scale_b, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                                  int epilogue, int out_f32, const float* bias, const void* aux_in, int64_t ld_aux_in,
                                  void* aux_out, int64_t ld_aux_out, float dropout_p, uint64_t dropout_seed,
                                  void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(A8 && B8t && C && (scale_a || row_scale_a) && scale_b, MERLOT_ESHAPE, "merlot_gemm_fp8_nt: null operand");
    MERLOT_CHECK(workspace && workspace_bytes >= NT_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0, MERLOT_ESHAPE,
                 "merlot_gemm_fp8_nt: needs the caller's zeroed workspace of merlot_gemm_nt_workspace_bytes() bytes");
    MERLOT_CHECK(M > 0 && N > 0 && K > 0 && M < (1LL << 31) && N < (1LL << 31), MERLOT_ESHAPE,
                 "merlot_gemm_fp8_nt: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    MERLOT_CHECK(((uintptr_t)A8 & 15) == 0 && ((uintptr_t)B8t & 15) == 0, MERLOT_EALIGN, "merlot_gemm_fp8_nt: A8/B8t must be 16-byte aligned");
    MERLOT_CHECK(dropout_p >= 0.f && dropout_p < 1.f, MERLOT_ESHAPE, "merlot_gemm_fp8_nt: dropout_p out of range");
    if (epilogue == MERLOT_EPI_RESIDUAL || epilogue == MERLOT_EPI_DGELU)
        MERLOT_CHECK(aux_in != nullptr, MERLOT_ESHAPE, "merlot_gemm_fp8_nt: epilogue %d needs aux_in", epilogue);
    GemmNTArgs a{};
    a.A = (const bf16*)A8; a.B = (const bf16*)B8t; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.alpha = alpha; a.bias = bias;
    a.scale_a = scale_a; a.scale_b = scale_b; a.row_scale = row_scale_a;
    a.aux_in = (const bf16*)aux_in; a.ld_aux_in = aux_in ? ld_aux_in : 0;
    a.aux_out = (bf16*)aux_out; a.ld_aux_out = aux_out ? ld_aux_out : 0;
    a.drop_thresh = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    a.drop_scale = 1.0f / (1.0f - dropout_p);
    a.drop_seed = dropout_seed;
    a.ctr = (unsigned int*)workspace;
    MERLOT_CHECK(p8_fp8_ok(a), MERLOT_ESHAPE,
                 "merlot_gemm_fp8_nt: needs K %% 128 == 0, K >= 256, lda/ldb %% 16 == 0 and operands under 2 GiB (K=%lld lda=%lld ldb=%lld)",
                 (long long)K, (long long)lda, (long long)ldb);
    return nt_status(launch_p8(a, epilogue, out_f32, (hipStream_t)stream, true), workspace, (hipStream_t)stream);
}

// ---- ABI v9: the NT GEMMs that ALSO write the 8-bit float copy of their bf16-rounded output (gemm_p8.inc "Q8").  q8_scale = merlot_quantize_f8's block of the
// output tensor: [0] (s) is read, [3] receives max|output| of this launch (atomic max; the caller zeroes / rotates the block: merlot_f8_scale_rotate).
static int q8_mode(int q8_fmt, const void* C) { return (q8_fmt == 0 ? 1 : 2) + (C ? 0 : 4); }

extern "C" int merlot_gemm_bf16_nt_q8(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      float alpha, int epilogue, const float* bias, const void* aux_in, int64_t ld_aux_in, float* colsum_out,
                                      void* q8_out, int64_t ld_q8, int q8_fmt, float* q8_scale, void* workspace, int64_t workspace_bytes,
                                      merlot_stream_t stream) {
    MERLOT_CHECK(A && Bt && q8_out && q8_scale && aux_in, MERLOT_ESHAPE, "merlot_gemm_bf16_nt_q8: null operand");
    MERLOT_CHECK(epilogue == MERLOT_EPI_DGELU, MERLOT_ESHAPE, "merlot_gemm_bf16_nt_q8: the 8-bit copy exists behind the DGELU epilogue only");
    MERLOT_CHECK(q8_fmt == 0 || q8_fmt == 1, MERLOT_ESHAPE, "merlot_gemm_bf16_nt_q8: q8_fmt is 0 (e4m3) or 1 (e5m2)");
    MERLOT_CHECK(workspace && workspace_bytes >= NT_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0, MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt_q8: needs the caller's zeroed workspace of merlot_gemm_nt_workspace_bytes() bytes");
    MERLOT_CHECK(M > 0 && N > 0 && K > 0 && M < (1LL << 31) && N < (1LL << 31) && lda % 8 == 0 && ldb % 8 == 0, MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt_q8: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    MERLOT_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)Bt & 15) == 0 && (!colsum_out || ((uintptr_t)colsum_out & 15) == 0), MERLOT_EALIGN,
                 "merlot_gemm_bf16_nt_q8: A / Bt / colsum_out must be 16-byte aligned");
    GemmNTArgs a{};
    a.A = (const bf16*)A; a.B = (const bf16*)Bt; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = C ? ldc : 0;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.alpha = alpha; a.bias = bias;
    a.aux_in = (const bf16*)aux_in; a.ld_aux_in = ld_aux_in;
    a.drop_scale = 1.0f;
    a.colsum = colsum_out;
    a.ctr = (unsigned int*)workspace;
    a.q8_out = (uint8_t*)q8_out; a.ld_q8 = ld_q8; a.q8_scale = q8_scale; a.q8_amax = reinterpret_cast<unsigned int*>(q8_scale + 3);
    MERLOT_CHECK(p8_q8_ok(a, false), MERLOT_ESHAPE,
                 "merlot_gemm_bf16_nt_q8: needs N a multiple of 256, K %% 64 == 0, leading dimensions multiples of 8 and 16-byte aligned operands under 4 GiB "
                 "(M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    return nt_status(launch_p8_q8(a, epilogue, q8_mode(q8_fmt, C), (hipStream_t)stream, false), workspace, (hipStream_t)stream);
}

extern "C" int merlot_gemm_fp8_nt_q8(const void* A8, int64_t lda, const float* scale_a, const float* row_scale_a, const void* B8t, int64_t ldb,
                                     const float* scale_b, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, int epilogue,
                                     const float* bias, void* aux_out, int64_t ld_aux_out, void* q8_out, int64_t ld_q8, int q8_fmt, float* q8_scale,
                                     void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(A8 && B8t && q8_out && q8_scale && aux_out && (scale_a || row_scale_a) && scale_b, MERLOT_ESHAPE, "merlot_gemm_fp8_nt_q8: null operand");
    MERLOT_CHECK(epilogue == MERLOT_EPI_GELU && q8_fmt == 0, MERLOT_ESHAPE,
                 "merlot_gemm_fp8_nt_q8: the 8-bit copy exists behind the GELU epilogue (with its pre-activation output), as e4m3");
    MERLOT_CHECK(workspace && workspace_bytes >= NT_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0, MERLOT_ESHAPE,
                 "merlot_gemm_fp8_nt_q8: needs the caller's zeroed workspace of merlot_gemm_nt_workspace_bytes() bytes");
    MERLOT_CHECK(M > 0 && N > 0 && K > 0 && M < (1LL << 31) && N < (1LL << 31), MERLOT_ESHAPE, "merlot_gemm_fp8_nt_q8: bad dims");
    MERLOT_CHECK(((uintptr_t)A8 & 15) == 0 && ((uintptr_t)B8t & 15) == 0, MERLOT_EALIGN, "merlot_gemm_fp8_nt_q8: A8 / B8t must be 16-byte aligned");
    GemmNTArgs a{};
    a.A = (const bf16*)A8; a.B = (const bf16*)B8t; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = C ? ldc : 0;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.alpha = alpha; a.bias = bias;
    a.scale_a = scale_a; a.scale_b = scale_b; a.row_scale = row_scale_a;
    a.aux_out = (bf16*)aux_out; a.ld_aux_out = ld_aux_out;
    a.drop_scale = 1.0f;
    a.ctr = (unsigned int*)workspace;
    a.q8_out = (uint8_t*)q8_out; a.ld_q8 = ld_q8; a.q8_scale = q8_scale; a.q8_amax = reinterpret_cast<unsigned int*>(q8_scale + 3);
    MERLOT_CHECK(p8_q8_ok(a, true), MERLOT_ESHAPE,
                 "merlot_gemm_fp8_nt_q8: needs N a multiple of 256, K %% 128 == 0, lda / ldb multiples of 16, the other leading dimensions of 8 and 16-byte "
                 "aligned operands under 4 GiB (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    return nt_status(launch_p8_q8(a, epilogue, q8_mode(q8_fmt, C), (hipStream_t)stream, true), workspace, (hipStream_t)stream);
}

// ABI v9: the plain NT GEMM (no epilogue option, bf16 output) on 8-bit operands whose A may be e5m2 -- the input-gradient GEMM fed by the GELU' epilogue's copy
extern "C" int merlot_gemm_f8_nt(const void* A8, int64_t lda, int fmt_a, const float* scale_a, const void* B8t, int64_t ldb, const float* scale_b,
                                 void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, const float* bias, void* workspace,
                                 int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(A8 && B8t && C && scale_a && scale_b, MERLOT_ESHAPE, "merlot_gemm_f8_nt: null operand");
    MERLOT_CHECK(fmt_a == 0 || fmt_a == 1, MERLOT_ESHAPE, "merlot_gemm_f8_nt: fmt_a is 0 (e4m3) or 1 (e5m2)");
    MERLOT_CHECK(workspace && workspace_bytes >= NT_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0, MERLOT_ESHAPE,
                 "merlot_gemm_f8_nt: needs the caller's zeroed workspace of merlot_gemm_nt_workspace_bytes() bytes");
    MERLOT_CHECK(M > 0 && N > 0 && K > 0 && M < (1LL << 31) && N < (1LL << 31), MERLOT_ESHAPE, "merlot_gemm_f8_nt: bad dims");
    MERLOT_CHECK(((uintptr_t)A8 & 15) == 0 && ((uintptr_t)B8t & 15) == 0, MERLOT_EALIGN, "merlot_gemm_f8_nt: A8 / B8t must be 16-byte aligned");
    GemmNTArgs a{};
    a.A = (const bf16*)A8; a.B = (const bf16*)B8t; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.alpha = alpha; a.bias = bias;
    a.scale_a = scale_a; a.scale_b = scale_b;
    a.drop_scale = 1.0f;
    a.ctr = (unsigned int*)workspace;
    MERLOT_CHECK(p8_fp8_ok(a), MERLOT_ESHAPE, "merlot_gemm_f8_nt: needs K %% 128 == 0, K >= 256, lda / ldb %% 16 == 0 and operands under 4 GiB");
    const int rc = fmt_a == 0 ? launch_p8_two<MERLOT_EPI_NONE, false, true, true>(a, (hipStream_t)stream)
                              : launch_p8_two<MERLOT_EPI_NONE, false, true, true, false, 0, 1>(a, (hipStream_t)stream);
    return nt_status(rc, workspace, (hipStream_t)stream);
}

extern "C" int64_t merlot_gemm_nt_workspace_bytes(void) { return NT_WORKSPACE_BYTES; }

extern "C" int merlot_softmax_ce(const float* logits, int64_t ld, const int32_t* labels, float* loss, int32_t* argmax,
                                 const float* rowscale, void* dlogits, int dl_bf16, int64_t ld_dl, int64_t rows, int C,
                                 merlot_stream_t stream);

// scripts/exp_vocab_ce.py (profiles/r03_f_vocab_ce.txt): row chunks sized for the 256 MiB Infinity Cache LOSE -- 12 800 x 50 370:
// 3.72 ms in one chunk, 5.25 ms in 160 MiB chunks (831 rows = 3.2 rounds of tiles per GEMM launch, 16 launch pairs), 6.9 ms at 96 MiB.
// The recommended scratch is therefore the whole logits matrix; smaller scratch still works (memory-constrained callers).
extern "C" int64_t merlot_vocab_ce_scratch_bytes(int64_t T, int64_t V) {
    if (T <= 0 || V <= 0) return 0;
    return T * ((V + 63) / 64 * 64) * 4;
}

extern "C" int merlot_vocab_ce_fwd(const void* h, int64_t ldh, const void* table, int64_t ldt, const float* out_bias,
                                   const int32_t* targets, const float* rowscale, float* loss, int32_t* argmax, void* dlogits,
                                   int64_t ld_dl, int64_t T, int64_t V, int64_t K, void* scratch, int64_t scratch_bytes,
                                   void* nt_workspace, int64_t nt_workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(h && table && targets && loss && scratch, MERLOT_ESHAPE, "merlot_vocab_ce_fwd: null operand");
    MERLOT_CHECK(T > 0 && V > 1 && K > 0 && K % BK == 0 && V < (1LL << 31), MERLOT_ESHAPE, "merlot_vocab_ce_fwd: bad dims");
    const int64_t ldl = (V + 63) / 64 * 64;
    MERLOT_CHECK(!dlogits || ld_dl >= V, MERLOT_ESHAPE, "merlot_vocab_ce_fwd: ld_dl < V");
    const int64_t chunk = scratch_bytes / (ldl * 4);
    MERLOT_CHECK(chunk >= (T < 256 ? T : 256), MERLOT_ESHAPE, "merlot_vocab_ce_fwd: scratch of %lld bytes holds %lld rows; need >= %lld",
                 (long long)scratch_bytes, (long long)chunk, (long long)(T < 256 ? T : 256));
    for (int64_t r0 = 0; r0 < T; r0 += chunk) {
        const int64_t rows = T - r0 < chunk ? T - r0 : chunk;
        int rc = merlot_gemm_bf16_nt((const bf16*)h + r0 * ldh, ldh, table, ldt, scratch, ldl, rows, V, K, 1.0f, MERLOT_EPI_NONE, 1, 0,
                                     out_bias, nullptr, 0, nullptr, 0, 0.f, 0, nullptr, nt_workspace, nt_workspace_bytes, stream);
        if (rc) return rc;
        rc = merlot_softmax_ce((const float*)scratch, ldl, targets + r0, loss + r0, argmax ? argmax + r0 : nullptr,
                               rowscale ? rowscale + r0 : nullptr, dlogits ? (bf16*)dlogits + r0 * ld_dl : nullptr, 1, ld_dl, rows,
                               (int)V, stream);
        if (rc) return rc;
    }
    return MERLOT_OK;
}

extern "C" int merlot_gemm_bf16_nt_plan(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0) return -1;
    return nt_plan(M, N, K);
}

extern "C" int64_t merlot_gemm_bf16_tn_workspace_bytes(int64_t M, int64_t N, int64_t R) {
    if (M <= 0 || N <= 0 || R < 8 * TnRingC::BK) return 0;
    // the larger of what the two split plans need: which kernel runs also depends on the leading dimensions (32-bit offsets)
    const TnPlan pl = tn_plan(M, N, R / TnRingC::BK * TnRingC::BK);
    int64_t need = pl.splits > 1 ? (int64_t)pl.splits * M * N * 4 : 0;
    if (tn_p8_shape(M, N, R / 64 * 64)) {
        const TnP8Plan p8 = tn_p8_plan(M, N, R / 64 * 64);
        const int64_t n8 = p8.splits > 1 ? (int64_t)p8.splits * M * N * 4 : 0;
        if (n8 > need) need = n8;
    }
    return need;
}

extern "C" int merlot_gemm_bf16_tn_cs(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                      int64_t M, int64_t N, int64_t R, float alpha, int accumulate, float* colsum_a, int64_t colsum_m,
                                      void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(A && B && C, MERLOT_ESHAPE, "merlot_gemm_bf16_tn: null operand");
    MERLOT_CHECK(M > 1 && N > 1 && R > 0 && R < (1LL << 31), MERLOT_ESHAPE, "merlot_gemm_bf16_tn: bad dims");
    MERLOT_CHECK(M % 2 == 0 && N % 2 == 0 && lda % 2 == 0 && ldb % 2 == 0, MERLOT_EALIGN,
                 "merlot_gemm_bf16_tn: M, N, lda, ldb must be even");
    MERLOT_CHECK(!colsum_a || (colsum_m > 0 && colsum_m <= M), MERLOT_ESHAPE, "merlot_gemm_bf16_tn_cs: colsum_m=%lld outside (0, M=%lld]",
                 (long long)colsum_m, (long long)M);
    GemmTNArgs a{};
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.R = (int)R; a.alpha = alpha;
    a.colsum_a = colsum_a; a.colsum_m = colsum_a ? (int)colsum_m : 0;
    return gemm_tn_dispatch(a, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int merlot_gemm_bf16_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                   int64_t M, int64_t N, int64_t R, float alpha, int accumulate, void* workspace,
                                   int64_t workspace_bytes, merlot_stream_t stream) {
    return merlot_gemm_bf16_tn_cs(A, lda, B, ldb, C, ldc, M, N, R, alpha, accumulate, nullptr, 0, workspace, workspace_bytes, stream);
}

// Weight gradient on 8-bit float operands (ABI v9; gemm_q8.inc): C (+)= alpha * deq_a[0] * deq_b[0] * A8^T B8
extern "C" int64_t merlot_gemm_f8_tn_workspace_bytes(int64_t M, int64_t N, int64_t R) {
    if (M <= 0 || N <= 0 || !tn_q8_shape(M, N, R)) return 0;
    const TnP8Plan pl = tn_q8_plan(M, N, R);
    return pl.splits > 1 ? (int64_t)pl.splits * M * N * 4 : 0;
}
extern "C" int merlot_gemm_f8_tn(const void* A8, int64_t lda, int fmt_a, const float* deq_a, const void* B8, int64_t ldb, int fmt_b,
                                 const float* deq_b, float* C, int64_t ldc, int64_t M, int64_t N, int64_t R, float alpha, int accumulate,
                                 void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(A8 && B8 && C && deq_a && deq_b, MERLOT_ESHAPE, "merlot_gemm_f8_tn: null operand");
    MERLOT_CHECK((fmt_a == 0 || fmt_a == 1) && (fmt_b == 0 || fmt_b == 1), MERLOT_ESHAPE, "merlot_gemm_f8_tn: formats are 0 (e4m3) or 1 (e5m2)");
    MERLOT_CHECK(M > 1 && N > 1 && R > 0 && R < (1LL << 31) && M < (1LL << 31) && N < (1LL << 31), MERLOT_ESHAPE, "merlot_gemm_f8_tn: bad dims");
    MERLOT_CHECK(tn_q8_shape(M, N, R), MERLOT_ESHAPE,
                 "merlot_gemm_f8_tn: needs R %% 128 == 0, R >= 2048, M, N >= 128 and at most 256 tiles of 256 x 256 (M=%lld N=%lld R=%lld)",
                 (long long)M, (long long)N, (long long)R);
    const auto pad16 = [](int64_t x) { return (x + 15) / 16 * 16; };
    MERLOT_CHECK(lda % 16 == 0 && ldb % 16 == 0 && lda >= pad16(M) && ldb >= pad16(N) && N % 4 == 0 && ldc % 4 == 0 && ldc >= N, MERLOT_ESHAPE,
                 "merlot_gemm_f8_tn: lda / ldb multiples of 16 covering M / N rounded up to 16, N and ldc multiples of 4 (lda=%lld ldb=%lld ldc=%lld)",
                 (long long)lda, (long long)ldb, (long long)ldc);
    MERLOT_CHECK((R + 128) * lda < (1LL << 32) && (R + 128) * ldb < (1LL << 32), MERLOT_ESHAPE, "merlot_gemm_f8_tn: operands must stay under 4 GiB");
    MERLOT_CHECK((((uintptr_t)A8 | (uintptr_t)B8 | (uintptr_t)C) & 15) == 0, MERLOT_EALIGN, "merlot_gemm_f8_tn: A8, B8, C must be 16-byte aligned");
    GemmTNQ8Args q{};
    q.t.A = (const bf16*)A8; q.t.B = (const bf16*)B8; q.t.C = C;
    q.t.lda = lda; q.t.ldb = ldb; q.t.ldc = ldc;
    q.t.M = (int)M; q.t.N = (int)N; q.t.R = (int)R; q.t.alpha = alpha;
    q.deq_a = deq_a; q.deq_b = deq_b;
    return tn_q8_launch(q, fmt_a, fmt_b, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// Patch-embed 16x16/16 convolution = explicit im2col (merlot_im2col_patches, csrc/conv.hip: the `image - 0.5` of
// utils/vision_transformer.py:193 is applied in the gather) + the production GEMMs on the [rows, 768] patch matrix.
extern "C" int merlot_im2col_patches(const void* image, void* patches, int n_img, int H, int W, int P, float shift,
                                     merlot_stream_t stream);

extern "C" int merlot_patch_embed_fwd(const void* image, int n_img, int H, int W, int P, const void* Wt, const float* bias,
                                      void* patches, void* out, int hidden, void* workspace, int64_t workspace_bytes,
                                      merlot_stream_t stream) {
    MERLOT_CHECK(Wt && out && patches && hidden > 1, MERLOT_ESHAPE, "merlot_patch_embed_fwd: null operand / bad hidden");
    MERLOT_CHECK(!workspace || workspace_bytes >= NT_WORKSPACE_BYTES, MERLOT_ESHAPE, "merlot_patch_embed_fwd: workspace too small");
    int rc = merlot_im2col_patches(image, patches, n_img, H, W, P, -0.5f, stream);
    if (rc) return rc;
    GemmNTArgs a{};
    a.A = (const bf16*)patches; a.B = (const bf16*)Wt; a.C = out;
    a.lda = P * P * 3; a.ldb = P * P * 3; a.ldc = hidden;
    a.M = n_img * (H / P) * (W / P); a.N = hidden; a.K = P * P * 3;
    a.alpha = 1.f; a.bias = bias;
    a.ctr = (unsigned int*)workspace;
    return gemm_nt_dispatch(a, MERLOT_EPI_NONE, 0, (hipStream_t)stream);
}

extern "C" int merlot_patch_embed_wgrad(const void* patches, int64_t rows, int K, const void* dY, float* dWt, int hidden,
                                        int accumulate, void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    // dWt[hidden][k] = sum_rows dY[row][hidden] * patches[row][k]: the wgrad GEMM on the saved patch matrix
    MERLOT_CHECK(patches && dY && dWt && rows > 0 && K > 0 && hidden > 1, MERLOT_ESHAPE, "merlot_patch_embed_wgrad: bad arguments");
    return merlot_gemm_bf16_tn(dY, hidden, patches, K, dWt, K, hidden, K, rows, 1.f, accumulate, workspace, workspace_bytes, stream);
}

#ifdef MERLOT_EXPERIMENTS
extern "C" int merlot_probe_persist_trace(void* dst, int64_t bytes, merlot_stream_t stream) {
    MERLOT_CHECK(dst && bytes > 0 && bytes <= (int64_t)sizeof(long long) * 256 * TRACE_TILES * 8, MERLOT_ESHAPE,
                 "merlot_probe_persist_trace: bad size");
    hipError_t e = hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_persist_trace), (size_t)bytes, 0, hipMemcpyDeviceToDevice,
                                            (hipStream_t)stream);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipMemcpyFromSymbolAsync: %s", hipGetErrorString(e));
    return MERLOT_OK;
}
#endif
