// Fused scaled-dot-product attention for gfx950, head_dim 64 (utils/transformer.py:98-127).
//
// Layout idea: every MFMA is issued "transposed" so that one lane owns one QUERY row of the score tile
// (S^T = K Q^T, 32x32x16 bf16 MFMA: lane = query column, 16 regs = 16 keys, partner lane+32 = other 16).
// The online softmax is then lane-local (one cross-lane exchange with lane^32 per tile), P converts to
// bf16 in registers and is fed straight back as the B operand of O^T = V^T P^T.  The V^T (and, in the
// backward, K^T / Q^T / dO^T) operand is produced from the row-major LDS tile by the gfx950 LDS
// transpose read `ds_read_b64_tr_b16` (4 rows x 16 columns per 16-lane group).
//
// LDS tile image ("H2"): R rows x 64 bf16 kept as two 32-column halves [2][R][64 B]; inside a half-row
// the four 16-B chunks are XOR-swizzled with (row>>2)&3.  Row stride 64 B puts the 4 rows of one
// transpose read in 4 disjoint bank quarters, and the swizzle makes the 16 rows of a ds_read_b128
// lane group hit 16 distinct slots.
//
// Masking semantics of the reference are kept: mask(b,i,j) = valid[b,i] & valid[b,j]; a masked score is the
// finite constant -1e10, so a padded QUERY row (every pair masked) attends uniformly over all S keys.  For such
// rows the kernels use the score 0 for every key instead of -1e10: the softmax is the same uniform 1/S, but the
// saved log-sum-exp is log(S) -- representable -- whereas -1e10 + log(S) rounds to -1e10 in fp32 and would make
// the recomputed probabilities of the backward / column-sum kernels 1 instead of 1/S.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr int64_t ATTN_WORKSPACE_BYTES = 64;             // item-claim counters of the persistent kernels (attention_pp.inc)

// a persistent launch that failed may have left the caller's claim counters non-zero ("zero on entry, left zero" is the kernels' whole
// protocol -- they carry no launch epoch): clear the block on the same stream before reporting the failure
static inline int pp_status(int rc, const char* what, void* workspace, hipStream_t s) {
    if (!rc) rc = merlot_launch_status(what);
    if (rc && workspace) (void)hipMemsetAsync(workspace, 0, (size_t)ATTN_WORKSPACE_BYTES, s);
    return rc;
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float MASKED_T = -1.0e10f * LOG2E;  // -1e10 in log2 units

struct AttnArgs {
    const bf16* qkv;
    int64_t ld;
    bf16* out;
    int64_t ldo;
    const bf16* dout;
    int64_t lddo;
    float* lse;          // [B, heads, S] natural log
    const float* delta;  // [B, heads, S]
    const uint8_t* valid;
    const int32_t* seg;  // optional [S] segment ids (model/modeling.py:160-168): a valid pair (q, k) is ALSO masked unless
                         // seg[q] == seg[k] or one of them is 0; same for every batch row; needs `valid`
    bf16* dqkv;
    int64_t lddqkv;
    int B, S, heads;
    float scale;
    // side outputs
    float* colsum_lo;
    float* colsum_hi;
    int qsplit;
    int valid_q_only;
    float weight;
    unsigned int* ctr;   // persistent kernels (attention_pp.inc): the caller's item-claim counters (zero on entry, left zero), or nullptr
    int dbg;             // experiments build only (MERLOT_ATTN_DBG): 1 = the streaming kernels move the data but skip the tile arithmetic
    // round 6 (ABI v9, merlot_attention_bwd_q8; the tiled dQ / dK dV pair only): an 8-bit float copy of dqkv from the launches that form it -- the operand of the
    // QKV weight gradient and input gradient on 8-bit operands (gemm_q8.inc) without a quantising pass.  dqkv8 == nullptr: off.
    uint8_t* dqkv8;      // [B * S, lddqkv8] bytes: f8(clamp(bf16(dqkv) * q8_scale[0])), q8_fmt 0 = e4m3, 1 = e5m2
    int64_t lddqkv8;
    const float* q8_scale;   // the tensor's merlot_quantize_f8 block (delayed scale in [0])
    unsigned int* q8_amax;   // &block[3] as bits: max|bf16(dqkv)| is max-ed into it
    int q8_fmt;
};

// the 8-bit copy of four consecutive bf16-rounded gradient values (one dword) + this lane's running amax
__device__ __forceinline__ void attn_store_q8(uint8_t* dst, const bf16x4& v4, float s, int fmt, float& amax) {
    float f[4];
    const float fm = fmt == 0 ? 448.f : 57344.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = (float)v4[e];
        amax = fmaxf(amax, fabsf(t));
        f[e] = __builtin_amdgcn_fmed3f(t * s, -fm, fm);
    }
    int w = 0;
    if (fmt == 0) {
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
    } else {
        w = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w, true);
    }
    *reinterpret_cast<int*>(dst) = w;
}
// one atomic per wave at most -- and none once the recorded maximum is at least this wave's (a plain read first: a stale value only costs a redundant atomic)
__device__ __forceinline__ void attn_amax_flush(unsigned int* dst, float amax, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const unsigned int bits = __float_as_uint(amax);
    if (lane == 0 && bits > __builtin_nontemporal_load(dst)) atomicMax(dst, bits);
}

__device__ __forceinline__ int h2_off(int R, int row, int chunk) {
    return (chunk >> 2) * R * 64 + row * 64 + ((((chunk & 3) ^ (row >> 2)) & 3) << 4);
}

// cooperative copy of a [rows<=R][64] bf16 tile (global row stride ld) into an H2 LDS image.
// 256 threads; R*8 16-B chunks.  Rows beyond `nrows_valid` are clamped to the last valid row.
template <int R>
__device__ __forceinline__ void tile_load_regs(const bf16* g, int64_t ld, int row0, int row_max, u32x4 (&regs)[R / 32],
                                               int tid) {
#pragma unroll
    for (int it = 0; it < R / 32; ++it) {
        const int idx = it * 256 + tid;
        const int half = idx / (R * 4);
        const int rem = idx - half * (R * 4);
        const int row = rem >> 2;
        const int chunk = half * 4 + (rem & 3);
        const int gr = min(row0 + row, row_max);
        regs[it] = *reinterpret_cast<const u32x4*>(g + (int64_t)gr * ld + chunk * 8);
    }
}
template <int R>
__device__ __forceinline__ void tile_store_lds(char* lds, const u32x4 (&regs)[R / 32], int tid) {
#pragma unroll
    for (int it = 0; it < R / 32; ++it) {
        const int idx = it * 256 + tid;
        const int half = idx / (R * 4);
        const int rem = idx - half * (R * 4);
        const int row = rem >> 2;
        const int chunk = half * 4 + (rem & 3);
        *reinterpret_cast<u32x4*>(lds + h2_off(R, row, chunk)) = regs[it];
    }
}

// transpose-read fragment: returns X^T fragment for a 32x32x16 MFMA operand whose MFMA-row/col index is
// the tile COLUMN (d) and whose 8 k-slots are tile ROWS  rb + {0..3} (slots 0-3) and rb + 8 + {0..3} (4-7),
// with rb already including the lane's 4*hi offset.  Column = d0 + (lane & 31).
template <int R>
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int rb, int d0, int lane) {
    const int i = lane & 15;
    const int col = d0 + ((lane >> 4) & 1) * 16 + 4 * (i & 3);
    const int row = rb + (i >> 2);
    const int o0 = h2_off(R, row, col >> 3) + (i & 1) * 8;
    const int o1 = h2_off(R, row + 8, col >> 3) + (i & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(lds + o0));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(lds + o1));
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = lo[e];
        r[4 + e] = hi4[e];
    }
    return r;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// row index inside a 32x32 accumulator for register r of a lane with half-index hi
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Mask handling without per-element selects: per key tile, 64 threads write two additive bias vectors (log2 units)
//   biasA[key] = 0 (valid key) | -1e10*log2e (masked key) | -inf (beyond S)      -- view of a VALID query row
//   biasB[key] = 0 (any key in range)                     | -inf (beyond S)      -- view of a PADDED query row
// and every lane reads its 32 key slots as 8 x 16 B from the array matching its own query row, with
// score' = fma(score, sc_lane, bias), sc_lane = scale*log2e for a valid row and 0 for a padded one (-> uniform).
constexpr float RESCALE_THR = 8.0f;   // defer-max: rescale O only when a row's running max grows by > 2^8

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 8192 + 768];
    char* ldsK = smem;
    char* ldsV = smem + 8192;
    float* biasA = reinterpret_cast<float*>(smem + 16384);
    float* biasB = biasA + 64;
    int* segK = reinterpret_cast<int*>(biasB + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int S = p.S;
    const int qw0 = blockIdx.x * 128 + wave * 32;  // first query row of this wave
    const bool wave_active = qw0 < S;
    const bf16* base = p.qkv + (int64_t)b * S * p.ld + h * 64;
    const bf16* kbase = base + p.heads * 64;
    const bf16* vbase = base + 2 * p.heads * 64;
    const uint8_t* vrow = MASKED ? p.valid + (int64_t)b * S : nullptr;

    const int q = min(qw0 + (lane & 31), S - 1);
    const bool qv = MASKED ? (vrow[q] != 0) : true;
    const bool segd = MASKED && p.seg != nullptr;
    const int sq = (segd && qv) ? p.seg[q] : 0;     // 0: every key allowed (viz / padded query rows stay uniform)
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        qf[kk] = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * p.ld + kk * 16 + hi * 8);

    f32x16 o[2] = {zero16(), zero16()};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = qv ? p.scale * LOG2E : 0.f;
    const float* bias = (qv ? biasA : biasB) + 4 * hi;

    const int nkt = (S + 63) / 64;
    u32x4 kreg[2], vreg[2];
    tile_load_regs<64>(kbase, p.ld, 0, S - 1, kreg, tid);
    tile_load_regs<64>(vbase, p.ld, 0, S - 1, vreg, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_store_lds<64>(ldsK, kreg, tid);
        tile_store_lds<64>(ldsV, vreg, tid);
        const bool ragged = MASKED || (kt * 64 + 64 > S);        // wave-uniform: this tile needs the bias vectors
        if (ragged && tid < 64) {
            const int kidx = kt * 64 + tid;
            const bool in = kidx < S;
            const bool kval = in && (MASKED ? vrow[min(kidx, S - 1)] != 0 : true);
            biasA[tid] = in ? (kval ? 0.f : MASKED_T) : -INFINITY;
            biasB[tid] = in ? 0.f : -INFINITY;
            if (segd) segK[tid] = p.seg[min(kidx, S - 1)];
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load_regs<64>(kbase, p.ld, (kt + 1) * 64, S - 1, kreg, tid);
            tile_load_regs<64>(vbase, p.ld, (kt + 1) * 64, S - 1, vreg, tid);
        }
        if (!wave_active) continue;
        // keys kt*64+32 .. kt*64+63 all beyond S (e.g. S = 198: the last tile holds 6 keys): skip that half entirely
        const bool half = kt * 64 + 32 >= S;

        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = zero16();
            if (kb == 1 && half) continue;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + h2_off(64, kb * 32 + (lane & 31), 2 * kk + hi));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kb], 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
        if (ragged) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && half) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + kb * 32 + 8 * g);
                    int sk[4] = {0, 0, 0, 0};
                    if (segd) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sk[e] = segK[4 * hi + kb * 32 + 8 * g + e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(st[kb][4 * g + e], sc, b4[e]);
                        if (segd) t = (sq == 0 || sk[e] == 0 || sk[e] == sq) ? t : MASKED_T;   // exactly -1e10, as a padded key
                        st[kb][4 * g + e] = t;
                        mloc = fmaxf(mloc, t);
                    }
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = st[kb][r] * sc;
                    st[kb][r] = t;
                    mloc = fmaxf(mloc, t);
                }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (__any(mloc > m_run + RESCALE_THR)) {       // wave-uniform; always taken on the first tile (m_run = -inf)
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = fast_exp2(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        float lsum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && half) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(st[kb][r] - m_run);
                lsum += pv;
                pf[kb][r >> 3][r & 7] = (bf16)pv;
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run += lsum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && half) continue;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const bf16x8 vf = tr_frag<64>(ldsV, kb * 32 + hf * 16 + 4 * hi, db * 32, lane);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][hf], o[db], 0, 0, 0);
                }
            }
    }

    if (wave_active && qw0 + (lane & 31) < S) {
        const float inv = 1.0f / l_run;
        bf16* orow = p.out + ((int64_t)b * S + q) * p.ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (bf16)(o[db][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * hi) = v4;
            }
        if (hi == 0 && p.lse) p.lse[((int64_t)b * p.heads + h) * S + q] = m_run * LN2 + logf(l_run);
    }
}

// ------------------------------------------------------------------------------------------------
// backward dQ: one wave per 32 query rows, loop over key tiles (same lane<->query layout and bias vectors as forward)
// ------------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 8192 + 768];
    char* ldsK = smem;
    char* ldsV = smem + 8192;
    float* biasA = reinterpret_cast<float*>(smem + 16384);
    float* biasB = biasA + 64;
    int* segK = reinterpret_cast<int*>(biasB + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int S = p.S;
    const int qw0 = blockIdx.x * 128 + wave * 32;
    const bool wave_active = qw0 < S;
    const bf16* base = p.qkv + (int64_t)b * S * p.ld + h * 64;
    const bf16* kbase = base + p.heads * 64;
    const bf16* vbase = base + 2 * p.heads * 64;
    const uint8_t* vrow = MASKED ? p.valid + (int64_t)b * S : nullptr;

    const int q = min(qw0 + (lane & 31), S - 1);
    const bool qv = MASKED ? (vrow[q] != 0) : true;
    const bool segd = MASKED && p.seg != nullptr;
    const int sq = (segd && qv) ? p.seg[q] : 0;
    bf16x8 qf[4], dof[4];
    const bf16* dorow = p.dout + ((int64_t)b * S + q) * p.lddo + h * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * p.ld + kk * 16 + hi * 8);
        dof[kk] = *reinterpret_cast<const bf16x8*>(dorow + kk * 16 + hi * 8);
    }
    const int64_t stat = ((int64_t)b * p.heads + h) * S + q;
    const float lse2 = p.lse[stat] * LOG2E;
    // delta_q = sum_d dO[q][d] * O[q][d], computed HERE from the dO fragments this lane holds anyway (+ one read of its O
    // rows) instead of by a separate pass over O and dO; lanes l and l ^ 32 hold the two halves of a row.  Published for the
    // dK / dV kernel, which runs after this one on the same stream.
    float dl = 0.f;
    {
        const bf16* orow = p.out + ((int64_t)b * S + q) * p.ldo + h * 64;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + kk * 16 + hi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)of[e] * (float)dof[kk][e];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (hi == 0 && qw0 + (lane & 31) < S) const_cast<float*>(p.delta)[stat] = dl;
    }
    const float sc = qv ? p.scale * LOG2E : 0.f;
    const float dsc = qv ? p.scale : 0.f;             // masked pairs have p == 0 exactly; padded query rows get ds = 0
    const float* bias = (qv ? biasA : biasB) + 4 * hi;

    f32x16 dq[2] = {zero16(), zero16()};
    const int nkt = (S + 63) / 64;
    u32x4 kreg[2], vreg[2];
    tile_load_regs<64>(kbase, p.ld, 0, S - 1, kreg, tid);
    tile_load_regs<64>(vbase, p.ld, 0, S - 1, vreg, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_store_lds<64>(ldsK, kreg, tid);
        tile_store_lds<64>(ldsV, vreg, tid);
        const bool ragged = MASKED || (kt * 64 + 64 > S);
        if (ragged && tid < 64) {
            const int kidx = kt * 64 + tid;
            const bool in = kidx < S;
            const bool kval = in && (MASKED ? vrow[min(kidx, S - 1)] != 0 : true);
            biasA[tid] = in ? (kval ? 0.f : MASKED_T) : -INFINITY;
            biasB[tid] = in ? 0.f : -INFINITY;
            if (segd) segK[tid] = p.seg[min(kidx, S - 1)];
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load_regs<64>(kbase, p.ld, (kt + 1) * 64, S - 1, kreg, tid);
            tile_load_regs<64>(vbase, p.ld, (kt + 1) * 64, S - 1, vreg, tid);
        }
        if (!wave_active) continue;
        const bool half = kt * 64 + 32 >= S;         // second 32 keys of this tile all beyond S: skip them

        bf16x8 dsf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && half) continue;
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = h2_off(64, kb * 32 + (lane & 31), 2 * kk + hi);
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + off);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(ldsV + off);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[kk], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 b4 = {-lse2, -lse2, -lse2, -lse2};
                if (ragged) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(bias + kb * 32 + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) b4[e] += t4[e];
                }
                if (segd) {                               // a pair the segment mask forbids has p == 0 exactly
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int sk = segK[4 * hi + kb * 32 + 8 * g + e];
                        if (!(sq == 0 || sk == 0 || sk == sq)) b4[e] = -INFINITY;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pv = fast_exp2(fmaf(st[r], sc, b4[e]));
                    dsf[kb][r >> 3][r & 7] = (bf16)(pv * (dp[r] - dl) * dsc);
                }
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && half) continue;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const bf16x8 ktf = tr_frag<64>(ldsK, kb * 32 + hf * 16 + 4 * hi, db * 32, lane);
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf[kb][hf], dq[db], 0, 0, 0);
                }
            }
    }
    float q8_m = 0.f;
    const float q8_s = p.dqkv8 ? p.q8_scale[0] : 0.f;
    if (wave_active && qw0 + (lane & 31) < S) {
        bf16* drow = p.dqkv + ((int64_t)b * S + q) * p.lddqkv + h * 64;
        uint8_t* drow8 = p.dqkv8 ? p.dqkv8 + ((int64_t)b * S + q) * p.lddqkv8 + h * 64 : nullptr;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (bf16)dq[db][4 * g + e];
                *reinterpret_cast<bf16x4*>(drow + db * 32 + 8 * g + 4 * hi) = v4;
                if (drow8) attn_store_q8(drow8 + db * 32 + 8 * g + 4 * hi, v4, q8_s, p.q8_fmt, q8_m);
            }
    }
    if (p.dqkv8 && wave_active) attn_amax_flush(p.q8_amax, q8_m, lane);
}

// ------------------------------------------------------------------------------------------------
// backward dK/dV: one wave per 32 keys, loop over 64-row query tiles (lane <-> key layout).  Per query tile the
// first wave publishes per-ROW vectors in LDS: lse2 (+inf beyond S => p = 0), delta, scq = scale*log2e (0 for a
// padded row) and mq = -1e10*log2e (0 for a padded row); a lane with a masked key adds mq, others add 0.
// ------------------------------------------------------------------------------------------------
// LOG (round 6; masked launches with the attention log, the sequence lengths of BASELINE configs[4]): the log's per-key sums of P over the VALID query rows below / from
// `qsplit` (a multiple of 4, checked by the host) are two lane accumulators of this pass, which forms P anyway -- until now the entry launched the tiled column-sum
// kernel in front (a second Q K^T walk: 2.85 ms per joint layer of config #5 beside this kernel's 3.1 ms).  The same construction as attn_bwd_fused_kernel<.., LOG>.
template <bool MASKED, bool LOG = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnArgs p) {
    static_assert(!LOG || MASKED, "the attention log counts valid pairs only: masked instantiations");
    __shared__ __attribute__((aligned(1024))) char smem[2 * 8192 + 5 * 256];
    char* ldsQ = smem;
    char* ldsO = smem + 8192;
    float* lds_lse = reinterpret_cast<float*>(smem + 16384);
    float* lds_dl = lds_lse + 64;
    float* lds_sc = lds_lse + 128;
    float* lds_mq = lds_lse + 192;
    int* lds_sq = reinterpret_cast<int*>(lds_lse + 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int S = p.S;
    const int kw0 = blockIdx.x * 128 + wave * 32;
    const bool wave_active = kw0 < S;
    const bf16* base = p.qkv + (int64_t)b * S * p.ld + h * 64;
    const bf16* kbase = base + p.heads * 64;
    const bf16* vbase = base + 2 * p.heads * 64;
    const bf16* dobase = p.dout + (int64_t)b * S * p.lddo + h * 64;
    const uint8_t* vrow = MASKED ? p.valid + (int64_t)b * S : nullptr;
    const float* lse_b = p.lse + ((int64_t)b * p.heads + h) * S;
    const float* dl_b = p.delta + ((int64_t)b * p.heads + h) * S;

    const int key = min(kw0 + (lane & 31), S - 1);
    const bool key_in = kw0 + (lane & 31) < S;
    const float kmul = (MASKED && !(vrow[key] != 0)) ? 1.f : 0.f;     // 1 for a masked key
    const bool segd = MASKED && p.seg != nullptr;
    const int sk = segd ? p.seg[key] : 0;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * p.ld + kk * 16 + hi * 8);
        vf[kk] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * p.ld + kk * 16 + hi * 8);
    }
    const float sc_const = p.scale * LOG2E;
    f32x16 dk[2] = {zero16(), zero16()};
    f32x16 dv[2] = {zero16(), zero16()};
    float log_lo1 = 0.f, log_hi1 = 0.f;                  // LOG: this key's sums over this lane's query rows

    const int nqt = (S + 63) / 64;
    u32x4 qreg[2], oreg[2];
    tile_load_regs<64>(base, p.ld, 0, S - 1, qreg, tid);
    tile_load_regs<64>(dobase, p.lddo, 0, S - 1, oreg, tid);
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        tile_store_lds<64>(ldsQ, qreg, tid);
        tile_store_lds<64>(ldsO, oreg, tid);
        if (tid < 64) {
            const int qi = qt * 64 + tid;
            const bool in = qi < S;
            lds_lse[tid] = in ? lse_b[qi] * LOG2E : INFINITY;  // +inf => p = 0 for rows beyond S
            lds_dl[tid] = in ? dl_b[qi] : 0.f;
            if (MASKED) {
                const bool qvl = in && (vrow[min(qi, S - 1)] != 0);
                lds_sc[tid] = qvl ? sc_const : 0.f;
                lds_mq[tid] = qvl ? MASKED_T : 0.f;
                if (segd) lds_sq[tid] = qvl ? p.seg[min(qi, S - 1)] : 0;     // padded rows stay uniform over all keys
            }
        }
        __syncthreads();
        if (qt + 1 < nqt) {
            tile_load_regs<64>(base, p.ld, (qt + 1) * 64, S - 1, qreg, tid);
            tile_load_regs<64>(dobase, p.lddo, (qt + 1) * 64, S - 1, oreg, tid);
        }
        if (!wave_active) continue;
        const bool half = qt * 64 + 32 >= S;         // second 32 queries of this tile all beyond S: skip them
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qb == 1 && half) continue;
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = h2_off(64, qb * 32 + (lane & 31), 2 * kk + hi);
                const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(ldsQ + off);
                const bf16x8 dofr = *reinterpret_cast<const bf16x8*>(ldsO + off);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[kk], st, 0, 0, 0);    // D[q][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[kk], dp, 0, 0, 0);
            }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ro = qb * 32 + 8 * g + 4 * hi;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lds_lse + ro);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(lds_dl + ro);
                f32x4 s4 = {sc_const, sc_const, sc_const, sc_const};
                f32x4 b4 = {-l4[0], -l4[1], -l4[2], -l4[3]};
                if (MASKED) {
                    s4 = *reinterpret_cast<const f32x4*>(lds_sc + ro);
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(lds_mq + ro);
#pragma unroll
                    for (int e = 0; e < 4; ++e) b4[e] = fmaf(m4[e], kmul, b4[e]);
                    if (segd) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int sq = lds_sq[ro + e];
                            if (!(sq == 0 || sk == 0 || sk == sq)) b4[e] = -INFINITY;
                        }
                    }
                }
                float tg = 0.f;                          // LOG: P summed over this lane's four rows of the group, padded query rows (uniform P) left out
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pv = fast_exp2(fmaf(st[r], s4[e], b4[e]));
                    const float ds = pv * (dp[r] - d4[e]) * (s4[e] * LN2);
                    pf[r >> 3][r & 7] = (bf16)pv;
                    dsf[r >> 3][r & 7] = (bf16)ds;
                    if (LOG) tg += s4[e] != 0.f ? pv : 0.f;
                }
                if (LOG) {                               // the four rows qt * 64 + ro .. + 3 lie on one side of qsplit (a multiple of 4)
                    const bool lo_side = qt * 64 + ro < p.qsplit;
                    log_lo1 += lo_side ? tg : 0.f;
                    log_hi1 += lo_side ? 0.f : tg;
                }
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int rb = qb * 32 + hf * 16 + 4 * hi;
                    const bf16x8 dot = tr_frag<64>(ldsO, rb, db * 32, lane);
                    const bf16x8 qtf = tr_frag<64>(ldsQ, rb, db * 32, lane);
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf[hf], dv[db], 0, 0, 0);   // D[d][key]
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[hf], dk[db], 0, 0, 0);
                }
        }
    }
    float q8_m = 0.f;
    const float q8_s = p.dqkv8 ? p.q8_scale[0] : 0.f;
    if (wave_active && key_in) {
        bf16* krow = p.dqkv + ((int64_t)b * S + key) * p.lddqkv + p.heads * 64 + h * 64;
        bf16* vrowp = krow + p.heads * 64;
        uint8_t* krow8 = p.dqkv8 ? p.dqkv8 + ((int64_t)b * S + key) * p.lddqkv8 + p.heads * 64 + h * 64 : nullptr;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 k4, v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    k4[e] = (bf16)dk[db][4 * g + e];
                    v4[e] = (bf16)dv[db][4 * g + e];
                }
                *reinterpret_cast<bf16x4*>(krow + db * 32 + 8 * g + 4 * hi) = k4;
                *reinterpret_cast<bf16x4*>(vrowp + db * 32 + 8 * g + 4 * hi) = v4;
                if (krow8) {
                    attn_store_q8(krow8 + db * 32 + 8 * g + 4 * hi, k4, q8_s, p.q8_fmt, q8_m);
                    attn_store_q8(krow8 + p.heads * 64 + db * 32 + 8 * g + 4 * hi, v4, q8_s, p.q8_fmt, q8_m);
                }
            }
    }
    if (p.dqkv8 && wave_active) attn_amax_flush(p.q8_amax, q8_m, lane);
    if (LOG) {                                           // the two row halves (lane, lane ^ 32) of this key, then one atomic per key and sum
        log_lo1 += __shfl_xor(log_lo1, 32, 64);
        log_hi1 += __shfl_xor(log_hi1, 32, 64);
        if (wave_active && lane < 32 && key_in) {
            if (p.colsum_lo) atomicAdd(p.colsum_lo + (int64_t)b * S + key, log_lo1 * p.weight);
            if (p.colsum_hi) atomicAdd(p.colsum_hi + (int64_t)b * S + key, log_hi1 * p.weight);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// side output: per-key column sums of the probabilities (one wave per (32 keys, head, batch))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_colsum_kernel(const AttnArgs p) {
    const int lane = threadIdx.x;
    const int hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int S = p.S;
    const int k0 = blockIdx.x * 32;
    const bf16* base = p.qkv + (int64_t)b * S * p.ld + h * 64;
    const bf16* kbase = base + p.heads * 64;
    const uint8_t* vrow = p.valid ? p.valid + (int64_t)b * S : nullptr;
    const float* lse_b = p.lse + ((int64_t)b * p.heads + h) * S;
    const int key = min(k0 + (lane & 31), S - 1);
    const bool key_in = k0 + (lane & 31) < S;
    const bool kv = key_in && (vrow ? vrow[key] != 0 : true);
    bf16x8 kf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        kf[kk] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * p.ld + kk * 16 + hi * 8);
    // per element:  p = exp2(s * a_r + c_r) * (row valid ? m_k : 1)
    //   a_r = valid query row ? scale*log2e : 0   (a padded query row scores 0 against every key: uniform 1/S)
    //   c_r = -lse_r*log2e if the row counts (inside S, and valid unless padded rows count too), else -inf => p = 0
    //   m_k = 1 for a valid key, 0 for a padded one (its probability under a valid query is exp2(-1e10...) = 0)
    const float sc = p.scale * LOG2E;
    const float m_k = kv ? 1.f : 0.f;
    const bool segd = p.seg != nullptr && vrow != nullptr;
    const int sk = segd ? p.seg[key] : 0;
    float acc_lo = 0.f, acc_hi = 0.f;
    const int nqb = (S + 31) / 32;
    for (int qb = 0; qb < nqb; ++qb) {
        const int qrow = min(qb * 32 + (lane & 31), S - 1);
        f32x16 st = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(base + (int64_t)qrow * p.ld + kk * 16 + hi * 8);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[kk], st, 0, 0, 0);  // D[q][key]
        }
        // this lane's query row of the block: validity and lse, handed to the 16 accumulator rows by lane shuffles
        const int ql = qb * 32 + (lane & 31);
        const bool q_in = ql < S;
        const bool qvl = q_in && (vrow ? vrow[qrow] != 0 : true);
        const bool counts = q_in && (p.valid_q_only ? qvl : true);
        const float my_a = qvl ? sc : 0.f;
        const float my_c = counts ? -lse_b[qrow] * LOG2E : -INFINITY;
        const int my_sq = (segd && qvl) ? p.seg[qrow] : 0;
        const bool all_lo = (qb + 1) * 32 <= p.qsplit, all_hi = qb * 32 >= p.qsplit;
        float part = 0.f, part_lo = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int src = acc_row(r) + 4 * hi;         // lane (0..31) that owns this accumulator row's query
            const float a_r = __shfl(my_a, src, 64);
            const float c_r = __shfl(my_c, src, 64);
            const float e = fast_exp2(fmaf(st[r], a_r, c_r));
            float pv = e * (a_r != 0.f ? m_k : 1.f);
            if (segd) {
                const int sq_r = __shfl(my_sq, src, 64);
                if (!(sq_r == 0 || sk == 0 || sk == sq_r)) pv = 0.f;
            }
            part += pv;
            if (!all_lo && !all_hi && qb * 32 + src < p.qsplit) part_lo += pv;
        }
        if (all_lo) acc_lo += part;
        else if (all_hi) acc_hi += part;
        else {
            acc_lo += part_lo;
            acc_hi += part - part_lo;
        }
    }
    acc_lo += __shfl_xor(acc_lo, 32, 64);
    acc_hi += __shfl_xor(acc_hi, 32, 64);
    if (lane < 32 && key_in) {
        if (p.colsum_lo) atomicAdd(p.colsum_lo + (int64_t)b * S + key, acc_lo * p.weight);
        if (p.colsum_hi) atomicAdd(p.colsum_hi + (int64_t)b * S + key, acc_hi * p.weight);
    }
}

#include "attention_res.inc"
#include "attention_fb.inc"
#include "attention_pp.inc"

int check_attn(const void* qkv, int64_t ld, int B, int S, int heads) {
    MERLOT_CHECK(qkv != nullptr, MERLOT_ESHAPE, "attention: null qkv");
    MERLOT_CHECK(B > 0 && S > 0 && heads > 0, MERLOT_ESHAPE, "attention: bad dims B=%d S=%d heads=%d", B, S, heads);
    MERLOT_CHECK(ld >= 3 * heads * 64 && ld % 8 == 0, MERLOT_EALIGN, "attention: ld=%lld too small / unaligned",
                 (long long)ld);
    MERLOT_CHECK(((uintptr_t)qkv & 15) == 0, MERLOT_EALIGN, "attention: qkv must be 16-byte aligned");
    MERLOT_CHECK(heads <= 65535 && B <= 65535, MERLOT_ESHAPE, "attention: grid too large");
    return MERLOT_OK;
}

}  // namespace

extern "C" int64_t merlot_attention_workspace_bytes(void) { return ATTN_WORKSPACE_BYTES; }

extern "C" int merlot_attention_fwd(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse,
                                    const uint8_t* valid, const int32_t* seg, int B, int S, int heads, float scale,
                                    float* colsum_lo, float* colsum_hi, int qsplit, int valid_q_only, float weight,
                                    void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    int rc = check_attn(qkv, ld, B, S, heads);
    if (rc) return rc;
    MERLOT_CHECK(out && ldo >= heads * 64 && ldo % 4 == 0, MERLOT_ESHAPE, "attention_fwd: bad out/ldo");
    MERLOT_CHECK(!seg || valid, MERLOT_ESHAPE, "attention: a segment mask needs the validity mask too");
    const bool want_cs = colsum_lo != nullptr || colsum_hi != nullptr;
    MERLOT_CHECK(!want_cs || lse, MERLOT_ESHAPE, "attention_fwd: the column sums need the lse output");
    AttnArgs a{};
    a.qkv = (const bf16*)qkv; a.ld = ld; a.out = (bf16*)out; a.ldo = ldo; a.lse = lse; a.valid = valid; a.seg = seg;
    a.B = B; a.S = S; a.heads = heads; a.scale = scale;
    a.colsum_lo = colsum_lo; a.colsum_hi = colsum_hi; a.qsplit = qsplit; a.valid_q_only = valid_q_only; a.weight = weight;
    // the persistent kernels claim their items from the caller's counters (ABI v7); without a workspace the one-shot kernels run
    MERLOT_CHECK(!workspace || (workspace_bytes >= ATTN_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0), MERLOT_ESHAPE,
                 "attention_fwd: workspace must be merlot_attention_workspace_bytes() bytes, 4-byte aligned, zero on entry");
    a.ctr = (unsigned int*)workspace;
    // K and V of one (batch, head) resident in LDS (attention_res.inc): with side outputs (they come from the same launch), and
    // for the plain forward of short unmasked sequences (the ViT pass: K | V through the CU's memory pipe once instead of once
    // per 128-row block, 685 vs 780 us at the bench shape, profiles/r03_j_attention_res.txt; masked sequences: level -> tiled)
    bool res_plain = !valid && S > 64 && S <= 256;
    // round 5: the persistent, prefetching single-pass kernel (attention_pp.inc) takes the plain forward of unmasked sequences of
    // up to 224 tokens -- the ViT pass
    bool pp = pp_fwd_ok(a, want_cs);
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_ATTN_DBG")) a.dbg = atoi(e);
    if (const char* e = getenv("MERLOT_ATTN_RESFWD")) res_plain = res_plain && atoi(e) != 0;
    if (const char* e = getenv("MERLOT_ATTN_PP")) pp = pp && atoi(e) != 0;
#endif
    if (pp) {
        rc = pp_fwd(a, (hipStream_t)stream);
        return pp_status(rc, "merlot_attention_fwd", workspace, (hipStream_t)stream);
    }
    // ... and its two-half sibling the plain forward of 257 .. 352 tokens, masked (the joint encoder in a training step) or not (the ViT of a 192 x 352 frame)
    bool ppm = ppm_fwd_ok(a, want_cs);
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_ATTN_PP")) ppm = ppm && atoi(e) != 0;
#endif
    if (ppm) {
        rc = ppm_fwd(a, (hipStream_t)stream);
        return pp_status(rc, "merlot_attention_fwd", workspace, (hipStream_t)stream);
    }
    if (S <= RES_MAX_S && (want_cs || res_plain)) {
        rc = res_fwd(a, (hipStream_t)stream);
        return rc ? rc : merlot_launch_status("merlot_attention_fwd");
    }
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_ATTN_RES")) {     // the resident forward for calls WITHOUT side outputs
        if (atoi(e) != 0 && S <= RES_MAX_S) {
            rc = res_fwd(a, (hipStream_t)stream);
            return rc ? rc : merlot_launch_status("merlot_attention_fwd");
        }
    }
#endif
    if (valid)
        hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(cdiv(S, 128), heads, B), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(cdiv(S, 128), heads, B), dim3(256), 0, (hipStream_t)stream, a);
    if (want_cs)                                         // long sequences: the separate column-sum pass
        hipLaunchKernelGGL(attn_colsum_kernel, dim3(cdiv(S, 32), heads, B), dim3(64), 0, (hipStream_t)stream, a);
    return merlot_launch_status("merlot_attention_fwd");
}

static int attention_bwd_impl(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                              int64_t lddo, const float* lse, const uint8_t* valid, const int32_t* seg, void* dqkv,
                              int64_t lddqkv, float* delta, int B, int S, int heads, float scale,
                              float* log_lo, float* log_hi, int log_qsplit, float log_weight,
                              void* workspace, int64_t workspace_bytes, merlot_stream_t stream,
                              void* dqkv8, int64_t lddqkv8, int q8_fmt, float* q8_scale) {
    int rc = check_attn(qkv, ld, B, S, heads);
    if (rc) return rc;
    MERLOT_CHECK(out && dout && lse && dqkv && delta, MERLOT_ESHAPE, "attention_bwd: null operand");
    MERLOT_CHECK(ldo % 8 == 0 && lddo % 8 == 0 && lddqkv % 4 == 0 && lddqkv >= 3 * heads * 64, MERLOT_EALIGN,
                 "attention_bwd: bad leading dims");
    AttnArgs a{};
    a.qkv = (const bf16*)qkv; a.ld = ld; a.out = (bf16*)out; a.ldo = ldo; a.dout = (const bf16*)dout; a.lddo = lddo;
    MERLOT_CHECK(!seg || valid, MERLOT_ESHAPE, "attention: a segment mask needs the validity mask too");
    a.lse = (float*)lse; a.delta = delta; a.valid = valid; a.seg = seg; a.dqkv = (bf16*)dqkv; a.lddqkv = lddqkv;
    a.B = B; a.S = S; a.heads = heads; a.scale = scale;
    MERLOT_CHECK(!workspace || (workspace_bytes >= ATTN_WORKSPACE_BYTES && ((uintptr_t)workspace & 3) == 0), MERLOT_ESHAPE,
                 "attention_bwd: workspace must be merlot_attention_workspace_bytes() bytes, 4-byte aligned, zero on entry");
    a.ctr = (unsigned int*)workspace;
    // the attention LOG side output (valid pairs only), taken from the backward instead of the forward: in the fused kernel it is
    // two lane accumulators of the dK / dV pass; on every other path the tiled column-sum kernel recomputes P (as the forward would)
    const bool want_log = log_lo != nullptr || log_hi != nullptr;
    a.colsum_lo = log_lo; a.colsum_hi = log_hi; a.qsplit = log_qsplit; a.valid_q_only = 1; a.weight = log_weight;
    hipStream_t s = (hipStream_t)stream;
    // S <= 512 without a segment mask (every pass of the 224^2 step): ONE launch, K | V and then Q | dO resident in LDS
    // (attention_fb.inc) -- 10 instead of 16 .. 24 [S, 64] tensors through the CU's memory pipe per (batch, head), same results
    int fb_mode = fb_ok(a) ? 1 : 0;
    if (dqkv8) {
        MERLOT_CHECK(q8_scale && (q8_fmt == 0 || q8_fmt == 1) && lddqkv8 % 4 == 0 && lddqkv8 >= 3 * heads * 64 && ((uintptr_t)dqkv8 & 3) == 0, MERLOT_ESHAPE,
                     "attention_bwd_q8: bad copy arguments");
        MERLOT_CHECK(!fb_mode && !pp_bwd_ok(a), MERLOT_ESHAPE,
                     "attention_bwd_q8: S = %d runs a kernel without the 8-bit output (merlot_attention_bwd_writes_q8 says which shapes have it)", S);
        a.dqkv8 = (uint8_t*)dqkv8; a.lddqkv8 = lddqkv8; a.q8_scale = q8_scale; a.q8_amax = reinterpret_cast<unsigned int*>(q8_scale + 3); a.q8_fmt = q8_fmt;
    }
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_ATTN_DBG")) a.dbg = atoi(e);
    if (const char* e = getenv("MERLOT_ATTN_FB")) fb_mode = fb_ok(a) ? atoi(e) : 0;
#endif
    // round 5: the persistent, prefetching backward of unmasked sequences of up to 224 tokens (the ViT pass; attention_pp.inc)
    bool pp = pp_bwd_ok(a);
#ifdef MERLOT_EXPERIMENTS
    if (const char* e = getenv("MERLOT_ATTN_PP")) pp = pp && atoi(e) != 0;
#endif
    if (pp) {
        rc = pp_bwd(a, s);
        return pp_status(rc, "merlot_attention_bwd", workspace, s);
    }
    if (want_log && !valid) {
        // an unmasked stack with the attention log (no product configuration): the log comes from the tiled column-sum kernel whichever kernel forms the
        // gradients, so they are formed by the one the plain call uses -- the same values with and without the log
        AttnArgs g = a;
        g.colsum_lo = g.colsum_hi = nullptr;
        bool ppg = pp_bwd_ok(g);
#ifdef MERLOT_EXPERIMENTS
        if (const char* e = getenv("MERLOT_ATTN_PP")) ppg = ppg && atoi(e) != 0;
#endif
        if (ppg) {
            hipLaunchKernelGGL(attn_colsum_kernel, dim3(cdiv(S, 32), heads, B), dim3(64), 0, s, a);
            rc = pp_bwd(g, s);
            return rc ? rc : merlot_launch_status("merlot_attention_bwd");
        }
    }
    if (fb_mode) {
        if (want_log && !fb_log_ok(a)) hipLaunchKernelGGL(attn_colsum_kernel, dim3(cdiv(S, 32), heads, B), dim3(64), 0, s, a);
        rc = fb_bwd(a, s);
        return rc ? rc : merlot_launch_status("merlot_attention_bwd");
    }
    // (round 6) a masked launch's log comes out of the dK / dV kernel's own P (attn_bwd_dkdv_kernel<true, true>) when its split is a multiple of 4
    const bool log_in_dkdv = want_log && valid != nullptr && (log_qsplit % 4 == 0);
    if (want_log && !log_in_dkdv) hipLaunchKernelGGL(attn_colsum_kernel, dim3(cdiv(S, 32), heads, B), dim3(64), 0, s, a);
    // everything else (longer sequences -- BASELINE config #5's S = 578 / 2832 --, segment masks, S <= 64, unaligned outputs):
    // the tiled pair, dQ (+ delta) then dK / dV, each recomputing S and dP.  (Round 3's persistent streaming dQ kernel sat between
    // the two; since the fused kernel took every sequence <= 512 it was reachable only for misaligned outputs and was retired.)
    if (valid) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3(cdiv(S, 128), heads, B), dim3(256), 0, s, a);
        if (log_in_dkdv) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, true>), dim3(cdiv(S, 128), heads, B), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, false>), dim3(cdiv(S, 128), heads, B), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3(cdiv(S, 128), heads, B), dim3(256), 0, s, a);
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, false>), dim3(cdiv(S, 128), heads, B), dim3(256), 0, s, a);
    }
    return merlot_launch_status("merlot_attention_bwd");
}

extern "C" int merlot_attention_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                                    int64_t lddo, const float* lse, const uint8_t* valid, const int32_t* seg, void* dqkv,
                                    int64_t lddqkv, float* delta, int B, int S, int heads, float scale,
                                    float* log_lo, float* log_hi, int log_qsplit, float log_weight,
                                    void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    return attention_bwd_impl(qkv, ld, out, ldo, dout, lddo, lse, valid, seg, dqkv, lddqkv, delta, B, S, heads, scale, log_lo, log_hi, log_qsplit, log_weight,
                              workspace, workspace_bytes, stream, nullptr, 0, 0, nullptr);
}
// ABI v9: merlot_attention_bwd that ALSO writes dqkv8 = f8(clamp(bf16(dqkv) * q8_scale[0])) and max-es max|dqkv| into q8_scale[3] (delayed scaling, as the
// GEMM producers) -- on the shapes the tiled dQ / dK dV pair takes: longer than 512 tokens, or with a segment mask, or at most 64 tokens
extern "C" int merlot_attention_bwd_writes_q8(int S, int has_segment_mask) { return (S > FB_MAX_S || has_segment_mask || S <= 64) ? 1 : 0; }
extern "C" int merlot_attention_bwd_q8(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                                       int64_t lddo, const float* lse, const uint8_t* valid, const int32_t* seg, void* dqkv,
                                       int64_t lddqkv, float* delta, int B, int S, int heads, float scale,
                                       float* log_lo, float* log_hi, int log_qsplit, float log_weight,
                                       void* dqkv8, int64_t lddqkv8, int q8_fmt, float* q8_scale,
                                       void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(dqkv8 && q8_scale, MERLOT_ESHAPE, "attention_bwd_q8: null copy arguments");
    return attention_bwd_impl(qkv, ld, out, ldo, dout, lddo, lse, valid, seg, dqkv, lddqkv, delta, B, S, heads, scale, log_lo, log_hi, log_qsplit, log_weight,
                              workspace, workspace_bytes, stream, dqkv8, lddqkv8, q8_fmt, q8_scale);
}

extern "C" int merlot_attention_colsum(const void* qkv, int64_t ld, const float* lse, const uint8_t* valid,
                                       const int32_t* seg, float* colsum_lo, float* colsum_hi, int qsplit,
                                       int valid_q_only, float weight,
                                       int B, int S, int heads, float scale, merlot_stream_t stream) {
    int rc = check_attn(qkv, ld, B, S, heads);
    if (rc) return rc;
    MERLOT_CHECK(lse && (colsum_lo || colsum_hi), MERLOT_ESHAPE, "attention_colsum: null operand");
    AttnArgs a{};
    MERLOT_CHECK(!seg || valid, MERLOT_ESHAPE, "attention: a segment mask needs the validity mask too");
    a.qkv = (const bf16*)qkv; a.ld = ld; a.lse = (float*)lse; a.valid = valid; a.seg = seg;
    a.colsum_lo = colsum_lo; a.colsum_hi = colsum_hi; a.qsplit = qsplit; a.valid_q_only = valid_q_only;
    a.weight = weight; a.B = B; a.S = S; a.heads = heads; a.scale = scale;
    hipLaunchKernelGGL(attn_colsum_kernel, dim3(cdiv(S, 32), heads, B), dim3(64), 0, (hipStream_t)stream, a);
    return merlot_launch_status("merlot_attention_colsum");
}

#ifdef MERLOT_EXPERIMENTS
extern "C" int merlot_probe_attn_trace(void* dst, int64_t bytes, merlot_stream_t stream) {
    MERLOT_CHECK(dst && bytes > 0 && bytes <= (int64_t)sizeof(long long) * 8 * 16 * 8, MERLOT_ESHAPE, "merlot_probe_attn_trace: bad size");
    hipError_t e = hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_attn_trace), (size_t)bytes, 0, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipMemcpyFromSymbolAsync: %s", hipGetErrorString(e));
    return MERLOT_OK;
}
#endif
