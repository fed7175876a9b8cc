// 3x3 convolution (stride 1, SAME padding) as an IMPLICIT GEMM on the ring-pipelined bf16 MFMA core (gemm_ring.h): the im2col
// gather of merlot_im2col3x3 (csrc/conv.hip) is expressed in the LDS-DMA source addresses instead of being written to HBM and
// read back.  utils/vision_transformer.py:8-56 (conv2d_fixed_padding / StdConv) is the reference layer; every 3x3 convolution of
// the released ResNet-hybrid stem except the root (3 input channels, stride 2) has stride 1 and C % 32 == 0.
//
//   y[(n,yo,xo)][co] = sum over (ky,kx,c) of x[n][yo+ky-1][xo+kx-1][c] * w[co][(ky,kx,c)]          (zero outside the image)
//
// A K-step of the ring (BK = 32 channels) lies inside ONE tap because C % 32 == 0: a lane's 16-B source is its pixel's base
// address plus a wave-uniform offset ((ky-1) * W + (kx-1)) * C + c0, or a page of zeros when the tap leaves the image on the
// side(s) the pixel touches (four border bits per lane, computed once).  The K order (ky, kx, c) and the MFMA sequence are those
// of the explicit path (im2col + merlot_gemm_bf16_nt on the same tile configuration), so the forward is bit-identical to it.
// The input-gradient of the layer is the same kernel on dY with the taps flipped in the weight matrix (host side): one fp32
// accumulation over all nine taps instead of nine bf16-rounded partial products summed by col2im.
// Traffic per launch: x is read once from HBM and ~9x from L2 (a tile's taps overlap), against 9x written + 9x read for the
// explicit patches; profiles/r04_t_conv_implicit.txt.
#include "gemm_ring.h"

namespace {

struct Conv3x3Args {
    const bf16* X;                                       // [n_img, H, W, C] bf16
    const bf16* Wt;                                      // [Co, ldw] bf16, k = (ky, kx, c)
    bf16* Y;                                             // [n_img * H * W, ldy] bf16
    const bf16* zeros;                                   // >= 16 B of zeros (the source of a tap outside the image)
    int64_t ldw, ldy;
    int M, N, K;                                         // pixels, filters, 9 * C
    int H, W, C;
    int ntn;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {   // block b runs on XCD b % 8: give each XCD a contiguous range of tiles
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename C>
__global__ __launch_bounds__(C::NT) void conv3x3_ring_kernel(const Conv3x3Args p) {
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    constexpr int BK = C::BK, S = C::STAGES;
    static_assert(BK == 32, "one K-step = 32 channels of one tap");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = wgid / p.ntn;
    const int tile_n = wgid - tile_m * p.ntn;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

    // ---- per-lane LDS-DMA sources: a pixel (A) or a filter (B) row, 16 B = 8 channels per lane
    constexpr int CH = BK / 8;
    const int prow = lane / CH, pch = lane % CH;
    const bf16* a_src[C::A_PIECES];
    int a_edge[C::A_PIECES];                             // bit 0: y == 0, 1: y == H-1, 2: x == 0, 3: x == W-1
    const bf16* b_src[C::B_PIECES];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int row = (wave * C::A_PIECES + i) * C::RP + prow;
        const int chunk = (pch ^ (row >> 2)) & 3;
        const int pix = min(m0 + row, p.M - 1);
        const int xx = pix % p.W, yy = (pix / p.W) % p.H;
        a_src[i] = p.X + (int64_t)pix * p.C + chunk * 8;
        a_edge[i] = (yy == 0 ? 1 : 0) | (yy == p.H - 1 ? 2 : 0) | (xx == 0 ? 4 : 0) | (xx == p.W - 1 ? 8 : 0);
    }
#pragma unroll
    for (int i = 0; i < C::B_PIECES; ++i) {
        const int row = (wave * C::B_PIECES + i) * C::RP + prow;
        const int chunk = (pch ^ (row >> 2)) & 3;
        b_src[i] = p.Wt + (int64_t)min(n0 + row, p.N - 1) * p.ldw + chunk * 8;
    }
    // stage(kt) is called for kt = 0, 1, 2, ... in order: the tap and the channel offset advance with it
    int st_tap = 0, st_c = 0;
    auto stage = [&](int kt) {
        char* la = dsm + (kt % S) * C::STAGE_BYTES + wave * C::A_PIECES * 1024;
        char* lb = dsm + (kt % S) * C::STAGE_BYTES + C::A_BYTES + wave * C::B_PIECES * 1024;
        const int ky = (st_tap * 11) >> 5, kx = st_tap - 3 * ky;             // st_tap / 3 for 0..8
        const int delta = ((ky - 1) * p.W + (kx - 1)) * p.C + st_c;
        const int out = (ky == 0 ? 1 : 0) | (ky == 2 ? 2 : 0) | (kx == 0 ? 4 : 0) | (kx == 2 ? 8 : 0);
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) {
            const bf16* src = (a_edge[i] & out) ? p.zeros : a_src[i] + delta;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(la + i * 1024), 16, 0, 0);
        }
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < C::B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(b_src[i] + k0), LDS_PTR(lb + i * 1024), 16, 0, 0);
        st_c += BK;
        if (st_c >= p.C) {
            st_c = 0;
            ++st_tap;
        }
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
        constexpr int KK = BK / 16;
        bf16x8 af[KK][C::FM], bfr[KK][C::FN];            // every fragment of the step first, into distinct registers
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int f = 0; f < C::FM; ++f)
                af[kk][f] = *reinterpret_cast<const bf16x8*>(la + ring::kc_off<BK>(a_row[f], 2 * kk + hi));
#pragma unroll
            for (int f = 0; f < C::FN; ++f)
                bfr[kk][f] = *reinterpret_cast<const bf16x8*>(lb + ring::kc_off<BK>(b_row[f], 2 * kk + hi));
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

    const int nk = p.K / BK;
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) stage(t);
    const int steady = nk - (S - 1);
    int kt = 0;
    for (; kt < steady; ++kt) {
        ring::wait_vmcnt<(S - 2) * C::LOADS>();          // stage kt landed; the newer S-2 stages stay in flight
        __builtin_amdgcn_s_barrier();
        stage(kt + S - 1);
        compute(kt);
    }
    for (; kt < nk; ++kt) {
        ring::wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(kt);
    }

    // ---- epilogue: accumulators -> this wave's fp32 slab in the idle ring -> row-contiguous 16-B bf16 stores
    __builtin_amdgcn_s_barrier();
    constexpr int COLS = C::FN * 32;
    constexpr int RSTRIDE = COLS * 4 + 16;
    constexpr int LPR = COLS / 8, RPP = 64 / LPR, PPB = 32 / RPP;
    char* slab = dsm + wave * C::EPI_BYTES;
    const int m_base = m0 + wm * C::FM * 32, n_base = n0 + wn * COLS;
    const int rr = lane / LPR, c0 = (lane % LPR) * 8;
    const int n = n_base + c0;
#pragma unroll
    for (int fi = 0; fi < C::FM; ++fi) {
#pragma unroll
        for (int fj = 0; fj < C::FN; ++fj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = acc[fi][fj][4 * q + e];
                *reinterpret_cast<f32x4*>(slab + (lane & 31) * RSTRIDE + (fj * 32 + 8 * q + 4 * hi) * 4) = t;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PPB; ++ps) {
            const int r = ps * RPP + rr;
            const int m = m_base + fi * 32 + r;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(slab + r * RSTRIDE + c0 * 4);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(slab + r * RSTRIDE + c0 * 4 + 16);
            if (m < p.M && n + 8 <= p.N) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (bf16)x0[e];
                    o[4 + e] = (bf16)x1[e];
                }
                *reinterpret_cast<bf16x8*>(p.Y + (int64_t)m * p.ldy + n) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

using ConvN64 = ring::Cfg<4, 1, 2, 2, 32, 3>;     // 256 pixels x 64 filters, 4 waves
using ConvN128 = ring::Cfg<4, 1, 2, 4, 32, 3>;    // 256 x 128, 4 waves
using ConvN256 = ring::Cfg<2, 4, 2, 2, 32, 3>;    // 128 x 256, 8 waves, two workgroups per CU

template <typename C>
int conv3x3_launch(Conv3x3Args& a, hipStream_t s) {
    auto kern = conv3x3_ring_kernel<C>;
    MERLOT_ENSURE_LDS(kern, C::LDS_BYTES, "merlot_conv3x3_bf16");
    const int ntm = cdiv(a.M, C::BM);
    a.ntn = cdiv(a.N, C::BN);
    hipLaunchKernelGGL(kern, dim3(ntm * a.ntn), dim3(C::NT), C::LDS_BYTES, s, a);
    return merlot_launch_status("merlot_conv3x3_bf16");
}


// ---- weight gradient of the same layer, implicit as well -----------------------------------------------------------------------
//   dw[co][(ky,kx,c)] = sum over pixels of dy[pix][co] * x[pix shifted by the tap][c]                  (fp32, split over pixel ranges)
// The TN form of the ring core (both operands as stored, fragments through ds_read_b64_tr_b16): A = dY [pixels][Co], B = the patch
// matrix [pixels][9 C] that is never built -- a lane's 16-B source is 8 channels of ONE tap for the whole launch (a 32-column panel
// lies inside a tap because C % 32 == 0), its pixel advances by 32 per K-step: (y, x) are kept incrementally, the tap's border test is
// two compares against per-lane constants.  Partial tiles of the pixel ranges go to the caller's fp32 workspace and a small kernel
// folds them (no atomics), as merlot_gemm_bf16_tn does.
struct ConvWgradArgs {
    const bf16* DY;                                      // [T][lddy]
    const bf16* X;                                       // [n_img, H, W, C]
    float* DW;                                           // [Co][lddw] fp32
    const bf16* zeros;
    int64_t lddy, lddw;
    int T, M, N;                                         // pixels, Co, 9 * C
    int H, W, C;
    int ntm, ntn, splits, rchunk;                        // rchunk: K-steps (32 pixels) per split
    int accumulate;
};

__device__ __forceinline__ bf16x8 tr_pair(const char* p) {   // LDS transpose-read of one 8-deep MFMA fragment from a 64-B-stride panel
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

template <typename C>
__global__ __launch_bounds__(C::NT) void conv3x3_wgrad_ring_kernel(const ConvWgradArgs p, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    constexpr int BK = C::BK, S = C::STAGES;
    static_assert(BK == 32, "a K-step = 32 pixels");
    constexpr int PPP = BK / 16;                         // 1 KiB pieces (16 rows x 32 columns) per 32-column panel
    constexpr int PANEL_BYTES = BK * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.ntm * p.ntn;
    const int ksteps = (p.T + BK - 1) / BK;
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);  // split-major: an XCD owns the tiles of one pixel range (they share x and dy in L2)
    const int split = wgid / ntiles;
    const int tile = wgid - split * ntiles;
    const int tile_m = tile / p.ntn;
    const int tile_n = tile - tile_m * p.ntn;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int ks = split * p.rchunk;
    const int ke = min(ksteps, ks + p.rchunk);
    const int nk = ke - ks;

    const bf16* a_col[C::A_PIECES];
    int a_row[C::A_PIECES];
#pragma unroll
    for (int i = 0; i < C::A_PIECES; ++i) {
        const int j = wave * C::A_PIECES + i;
        const int panel = j / PPP, rb = j % PPP;
        a_col[i] = p.DY + min(m0 + panel * 32 + (lane & 3) * 8, (int)p.lddy - 8);
        a_row[i] = ks * BK + rb * 16 + (lane >> 2);
    }
    const bf16* b_base[C::B_PIECES];
    int b_pix[C::B_PIECES], b_x[C::B_PIECES], b_y[C::B_PIECES], b_xbad[C::B_PIECES], b_ybad[C::B_PIECES];
#pragma unroll
    for (int i = 0; i < C::B_PIECES; ++i) {
        const int j = wave * C::B_PIECES + i;
        const int panel = j / PPP, rb = j % PPP;
        const int f = min(n0 + panel * 32 + (lane & 3) * 8, p.N - 8);
        const int tap = f / p.C, c = f - tap * p.C;
        const int ky = tap / 3, kx = tap - 3 * ky;
        b_base[i] = p.X + ((ky - 1) * p.W + (kx - 1)) * p.C + c;
        const int pix = ks * BK + rb * 16 + (lane >> 2);
        b_pix[i] = pix;
        b_x[i] = pix % p.W;
        b_y[i] = (pix / p.W) % p.H;
        b_ybad[i] = ky == 0 ? 0 : (ky == 2 ? p.H - 1 : -1);
        b_xbad[i] = kx == 0 ? 0 : (kx == 2 ? p.W - 1 : -1);
    }
    const int qW = BK / p.W, rW = BK - qW * p.W;
    auto stage = [&](int t) {                            // called for t = 0, 1, 2, ... in order: the lanes' pixels advance with it
        char* la = dsm + (t % S) * C::STAGE_BYTES + wave * C::A_PIECES * 1024;
        char* lb = dsm + (t % S) * C::STAGE_BYTES + C::A_BYTES + wave * C::B_PIECES * 1024;
#pragma unroll
        for (int i = 0; i < C::A_PIECES; ++i) {
            const bf16* src = a_col[i] + (int64_t)min(a_row[i], p.T - 1) * p.lddy;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(la + i * 1024), 16, 0, 0);
            a_row[i] += BK;
        }
#pragma unroll
        for (int i = 0; i < C::B_PIECES; ++i) {
            const bool bad = (b_pix[i] >= p.T) | (b_y[i] == b_ybad[i]) | (b_x[i] == b_xbad[i]);
            const bf16* src = bad ? p.zeros : b_base[i] + (int64_t)b_pix[i] * p.C;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lb + i * 1024), 16, 0, 0);
            b_pix[i] += BK;
            b_x[i] += rW;
            b_y[i] += qW;
            if (b_x[i] >= p.W) {
                b_x[i] -= p.W;
                ++b_y[i];
            }
            while (b_y[i] >= p.H) b_y[i] -= p.H;
        }
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

    // ---- epilogue: this pixel range's partial tile, fp32, row-contiguous
    __builtin_amdgcn_s_barrier();
    constexpr int COLS = C::FN * 32;
    constexpr int RSTRIDE = COLS * 4 + 16;
    constexpr int LPR = COLS / 8, RPP = 64 / LPR, PPB = 32 / RPP;
    char* slab = dsm + wave * C::EPI_BYTES;
    const int m_base = m0 + wm * C::FM * 32, n_base = n0 + wn * COLS;
    const int rr = lane / LPR, c0 = (lane % LPR) * 8;
    const int n = n_base + c0;
    float* out = p.splits > 1 ? ws + (int64_t)split * p.M * p.N : p.DW;
    const int64_t ldo = p.splits > 1 ? p.N : p.lddw;
    const bool add = p.splits == 1 && p.accumulate;
#pragma unroll
    for (int fi = 0; fi < C::FM; ++fi) {
#pragma unroll
        for (int fj = 0; fj < C::FN; ++fj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[fi][fj][4 * q + e];
                *reinterpret_cast<f32x4*>(slab + (lane & 31) * RSTRIDE + (fj * 32 + 8 * q + 4 * hi) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PPB; ++ps) {
            const int r = ps * RPP + rr;
            const int m = m_base + fi * 32 + r;
            f32x4 x0 = *reinterpret_cast<const f32x4*>(slab + r * RSTRIDE + c0 * 4);
            f32x4 x1 = *reinterpret_cast<const f32x4*>(slab + r * RSTRIDE + c0 * 4 + 16);
            if (m < p.M && n + 8 <= p.N) {
                float* o = out + (int64_t)m * ldo + n;
                if (add) {
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(o), o1 = *reinterpret_cast<const f32x4*>(o + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[e] += o0[e];
                        x1[e] += o1[e];
                    }
                }
                *reinterpret_cast<f32x4*>(o) = x0;
                *reinterpret_cast<f32x4*>(o + 4) = x1;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// dw[m][n] = (accumulate ? dw : 0) + sum_s ws[s][m][n]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ dw,
                                                                int64_t lddw, int M, int N, int accumulate) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4, plane = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n4;
        const int n = (int)(i - m * n4) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + m * N + n);
        for (int s = 1; s < splits; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ws + s * plane + m * N + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        float* o = dw + m * lddw + n;
        if (accumulate) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        *reinterpret_cast<f32x4*>(o) = acc;
    }
}

using ConvW = ring::Cfg<2, 4, 2, 2, 32, 3>;       // 128 filters x 256 patch columns, 8 waves, two workgroups per CU
using ConvW64 = ring::Cfg<1, 4, 2, 2, 32, 3>;     // 64 filters x 256 patch columns, 4 waves (the 32 / 64-filter layers at 112^2 and 56^2)

struct ConvWgradPlan {
    int bm, ntm, ntn, splits, rchunk;
};
ConvWgradPlan conv_wgrad_plan(int64_t T, int C, int Co) {
    ConvWgradPlan pl;
    pl.bm = Co <= 64 ? ConvW64::BM : ConvW::BM;
    pl.ntm = cdiv(Co, pl.bm);
    pl.ntn = cdiv(9 * C, ConvW::BN);
    const int tiles = pl.ntm * pl.ntn;
    const int ksteps = cdiv(T, ConvW::BK);
    int splits = 512 / tiles;                            // one round of resident workgroups (two per CU)
    if (splits > ksteps / 8) splits = ksteps / 8;
    if (splits < 1) splits = 1;
    pl.rchunk = cdiv(ksteps, splits);
    pl.splits = cdiv(ksteps, pl.rchunk);
    return pl;
}

}  // namespace

extern "C" int merlot_conv3x3_bf16(const void* x, const void* w, int64_t ldw, void* y, int64_t ldy, int n_img, int H, int W,
                                   int C, int Co, const void* zeros, merlot_stream_t stream) {
    MERLOT_CHECK(x && w && y && zeros, MERLOT_ESHAPE, "merlot_conv3x3_bf16: null operand");
    MERLOT_CHECK(n_img > 0 && H > 0 && W > 0 && (int64_t)n_img * H * W < (1LL << 31), MERLOT_ESHAPE,
                 "merlot_conv3x3_bf16: bad geometry n=%d H=%d W=%d", n_img, H, W);
    MERLOT_CHECK(C >= 32 && C % 32 == 0, MERLOT_ESHAPE, "merlot_conv3x3_bf16: C=%d must be a multiple of 32 (a K-step is 32 channels of one tap)", C);
    MERLOT_CHECK(Co >= 8 && Co % 8 == 0 && ldy >= Co && ldy % 8 == 0 && ldw >= 9 * C && ldw % 8 == 0, MERLOT_ESHAPE,
                 "merlot_conv3x3_bf16: Co=%d must be a multiple of 8, ldy=%lld >= Co and ldw=%lld >= 9*C multiples of 8", Co,
                 (long long)ldy, (long long)ldw);
    MERLOT_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)zeros) & 15) == 0, MERLOT_EALIGN,
                 "merlot_conv3x3_bf16: operands must be 16-byte aligned");
    Conv3x3Args a{};
    a.X = (const bf16*)x; a.Wt = (const bf16*)w; a.Y = (bf16*)y; a.zeros = (const bf16*)zeros;
    a.ldw = ldw; a.ldy = ldy;
    a.M = n_img * H * W; a.N = Co; a.K = 9 * C;
    a.H = H; a.W = W; a.C = C;
    hipStream_t s = (hipStream_t)stream;
    if (Co <= 64) return conv3x3_launch<ConvN64>(a, s);
    if (Co <= 128) return conv3x3_launch<ConvN128>(a, s);
    return conv3x3_launch<ConvN256>(a, s);
}

extern "C" int64_t merlot_conv3x3_wgrad_workspace_bytes(int n_img, int H, int W, int C, int Co) {
    if (n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || Co <= 0) return 0;
    const ConvWgradPlan pl = conv_wgrad_plan((int64_t)n_img * H * W, C, Co);
    return pl.splits > 1 ? (int64_t)pl.splits * Co * 9 * C * 4 : 0;
}

extern "C" int merlot_conv3x3_wgrad_bf16(const void* dy, int64_t lddy, const void* x, float* dw, int64_t lddw, int n_img, int H,
                                         int W, int C, int Co, int accumulate, const void* zeros, void* workspace,
                                         int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(dy && x && dw && zeros, MERLOT_ESHAPE, "merlot_conv3x3_wgrad_bf16: null operand");
    MERLOT_CHECK(n_img > 0 && H > 0 && W > 0 && (int64_t)n_img * H * W < (1LL << 31) - 65536, MERLOT_ESHAPE,
                 "merlot_conv3x3_wgrad_bf16: bad geometry n=%d H=%d W=%d", n_img, H, W);
    MERLOT_CHECK(C >= 32 && C % 32 == 0, MERLOT_ESHAPE, "merlot_conv3x3_wgrad_bf16: C=%d must be a multiple of 32 (a 32-column panel is one tap)", C);
    MERLOT_CHECK(Co >= 8 && lddy >= Co && lddy % 8 == 0 && lddw >= 9 * C && lddw % 4 == 0, MERLOT_ESHAPE,
                 "merlot_conv3x3_wgrad_bf16: Co=%d >= 8, lddy=%lld >= Co a multiple of 8, lddw=%lld >= 9*C a multiple of 4", Co,
                 (long long)lddy, (long long)lddw);
    MERLOT_CHECK((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw | (uintptr_t)zeros | (uintptr_t)workspace) & 15) == 0, MERLOT_EALIGN,
                 "merlot_conv3x3_wgrad_bf16: operands must be 16-byte aligned");
    const int64_t T = (int64_t)n_img * H * W;
    const ConvWgradPlan pl = conv_wgrad_plan(T, C, Co);
    if (pl.splits > 1) {
        const int64_t need = (int64_t)pl.splits * Co * 9 * C * 4;
        MERLOT_CHECK(workspace && workspace_bytes >= need, MERLOT_ESHAPE,
                     "merlot_conv3x3_wgrad_bf16: workspace too small (%lld < %lld bytes, merlot_conv3x3_wgrad_workspace_bytes)",
                     (long long)workspace_bytes, (long long)need);
    }
    ConvWgradArgs a{};
    a.DY = (const bf16*)dy; a.X = (const bf16*)x; a.DW = dw; a.zeros = (const bf16*)zeros;
    a.lddy = lddy; a.lddw = lddw;
    a.T = (int)T; a.M = Co; a.N = 9 * C;
    a.H = H; a.W = W; a.C = C;
    a.ntm = pl.ntm; a.ntn = pl.ntn; a.splits = pl.splits; a.rchunk = pl.rchunk;
    a.accumulate = accumulate;
    hipStream_t s = (hipStream_t)stream;
    if (pl.bm == ConvW64::BM) {
        auto kern = conv3x3_wgrad_ring_kernel<ConvW64>;
        MERLOT_ENSURE_LDS(kern, ConvW64::LDS_BYTES, "merlot_conv3x3_wgrad_bf16");
        hipLaunchKernelGGL(kern, dim3(pl.ntm * pl.ntn * pl.splits), dim3(ConvW64::NT), ConvW64::LDS_BYTES, s, a, (float*)workspace);
    } else {
        auto kern = conv3x3_wgrad_ring_kernel<ConvW>;
        MERLOT_ENSURE_LDS(kern, ConvW::LDS_BYTES, "merlot_conv3x3_wgrad_bf16");
        hipLaunchKernelGGL(kern, dim3(pl.ntm * pl.ntn * pl.splits), dim3(ConvW::NT), ConvW::LDS_BYTES, s, a, (float*)workspace);
    }
    if (pl.splits > 1) {
        const int64_t total = (int64_t)a.M * (a.N / 4);
        int grid = (int)((total + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)workspace, pl.splits, dw, lddw, a.M,
                           a.N, accumulate);
    }
    return merlot_launch_status("merlot_conv3x3_wgrad_bf16");
}
