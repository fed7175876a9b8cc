/* libmerlot_hip.so -- C-ABI of the MI355X-native MERLOT pretraining hot path.
 *
 * The reference (rowanz/merlot) is pure Python on TF-1.15/XLA: it has NO FFI / plugin registry.
 * Its seam is the Python surface `MerlotModel` + `merlot.yaml` (model/modeling.py:47-668), which
 * merlot_amd/modeling.py mirrors; this header is the boundary BELOW that surface.  Every entry
 * point cites the reference code whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit dims / leading dimensions (in ELEMENTS), a hipStream_t
 *     passed as void*.  No torch types.  The caller owns every buffer (including workspaces); the
 *     library never allocates, never synchronises, keeps no global state => re-entrant per stream.
 *   - returns MERLOT_OK (0) or a negative MERLOT_E* code; merlot_last_error() gives a thread-local
 *     message for the last failure.
 *   - "bf16" buffers are uint16 bit patterns of bfloat16; "f32" are IEEE float.
 *   - activations are row-major [rows, features]; Linear weights are held [out, in] ("Wt").
 */
#ifndef MERLOT_HIP_H
#define MERLOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MERLOT_OK 0
#define MERLOT_ESHAPE (-1)
#define MERLOT_EDTYPE (-2)
#define MERLOT_EALIGN (-3)
#define MERLOT_ELAUNCH (-4)

typedef void* merlot_stream_t;

/* Bumped whenever a signature of this header changes.  merlot_abi_version() returns the value the library was built with;
 * a binding must compare the two before its first call (merlot_amd/lib.py does, and refuses a mismatching library). */
#define MERLOT_ABI_VERSION 11

const char* merlot_last_error(void);
int merlot_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Dense contractions (MFMA).  Replaces tf.layers.dense / tf.matmul at utils/transformer.py:21-25,
 * 67-82,98,120,130-135,149-161 and their tf.gradients backward (utils/optimization.py:176).
 * ---------------------------------------------------------------------------------------------- */
enum merlot_epilogue {
    /* alpha * acc + bias is ONE fused multiply-add of the fp32 accumulator in the persistent (ping-pong) bf16 kernel -- interior and
     * boundary tiles alike (round 6) --; the ring kernels and the fp8 kernels round alpha * acc first.  With alpha == 1 all agree. */
    MERLOT_EPI_NONE = 0,          /* C = alpha*acc (+bias)                                         */
    MERLOT_EPI_GELU = 1,          /* u = alpha*acc+bias ; aux_out = u (optional) ; C = gelu(u)     */
    MERLOT_EPI_RESIDUAL = 2,      /* C = aux_in + dropout(alpha*acc+bias)                          */
    MERLOT_EPI_DGELU = 3          /* C = (alpha*acc) * gelu'(aux_in)                               */
};

/* C[M,N] = epilogue(alpha * A[M,K] . Bt[N,K]^T).  A, Bt bf16, K-contiguous; K % 64 == 0;
 * lda, ldb % 8 == 0.  C is bf16 (out_f32=0) or f32 (out_f32=1; accumulate=1 adds into C).
 * bias: f32 [N] or NULL.  aux_in / aux_out: bf16 [M, ld*] (see enum).  dropout_p in [0,1):
 * keep-mask is a counter-based hash of (dropout_seed, m*N+n) -- one 32-bit hash per pair of elements, p resolved to
 * 2^-16 -- survivors scaled 1/(1-p) (utils/model_utils.py:335-349); merlot_dropout_apply / merlot_ln_bwd regenerate
 * the same mask.  GELU / GELU' in the epilogues are the exact-erf forms of utils/model_utils.py:96-110 evaluated by
 * degree-10 polynomials (abs error 1.6e-5 / 1.1e-4 over the whole line, below the bf16 rounding of C; scripts/fit_gelu_poly.py).
 * Large problems run on a persistent kernel whose workgroups CLAIM their tiles through a counter block in CALLER-owned
 * `workspace` (merlot_gemm_nt_workspace_bytes() bytes, 4-byte aligned, zero on entry; the last workgroup out leaves it
 * zero again, so one block serves every launch of a stream and needs zeroing once, at allocation).  The library itself
 * keeps no device or host state: launches on different streams are independent as long as each stream uses its own
 * block.  workspace == NULL is accepted only for shapes merlot_gemm_bf16_nt_plan() maps to a non-claiming kernel
 * (plans other than 22); otherwise MERLOT_ESHAPE.  An A operand of 4 GiB or more is cut into row ranges of that kernel internally.
 * colsum_out (optional, f32 [N], ACCUMULATED; bf16 output only): column sums of the stored C -- the bias gradient of the
 * layer whose output gradient this launch produces (utils/transformer.py:149-153) -- fused into the epilogue of the
 * persistent kernel, otherwise computed by merlot_colsum_bf16 right behind the GEMM. */
int merlot_gemm_bf16_nt(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc,
                        int64_t M, int64_t N, int64_t K, float alpha, int epilogue, int out_f32, int accumulate,
                        const float* bias, const void* aux_in, int64_t ld_aux_in, void* aux_out,
                        int64_t ld_aux_out, float dropout_p, uint64_t dropout_seed, float* colsum_out,
                        void* workspace, int64_t workspace_bytes, merlot_stream_t stream);
int64_t merlot_gemm_nt_workspace_bytes(void);

/* ABI v8 (round 6; SURVEY 8(b) `merlot_ln_residual_fwd`): the residual sub-layer tail AND the LayerNorm that follows it, one call --
 *     C      = aux_in + dropout(alpha * A . Bt^T + bias)          (utils/transformer.py:130-136, 158-162: dense + dropout + residual)
 *     ln_out = LayerNorm(C) * gamma + beta,  mean / rstd [M]      (utils/model_utils.py:113-130; utils/transformer.py:214, 220)
 * bf16 C and ln_out, dense ([M, N], ldc == ld_ln == N, N % 256 == 0).  Where merlot_gemm_bf16_nt_ln_plan() says 1 (N == 768, at least 96 row
 * blocks of 256, the persistent kernel's shape) and the operands are 16-byte aligned, ONE launch does both: each 256 x 256 tile leaves per-row
 * partial statistics of its stored values, and the last of a row block's three tiles to finish normalises the block from the L2 (no second
 * launch, no HBM read of C).  Otherwise, and for the rows of a ragged last row block, the same GEMM is followed by the merlot_ln_fwd kernel.
 * Statistics are combined from 64-column segments (sum, centred sum of squares; exact pairwise combination), so mean / rstd agree with
 * merlot_ln_fwd to fp32 rounding, not bit for bit.
 * ln_workspace: merlot_gemm_nt_ln_workspace_bytes(M, N) bytes, 16-byte aligned, its first ceil(M/256) uint32 ZERO on entry and left zero
 * (arrival counters; the rest is scratch).  The layout is a function of M: a block serves launches of ONE M (and N) on one stream at a time -- reuse it for
 * another M only after clearing it again.  workspace: as merlot_gemm_bf16_nt. */
int64_t merlot_gemm_nt_ln_workspace_bytes(int64_t M, int64_t N);
int merlot_gemm_bf16_nt_ln_plan(int64_t M, int64_t N, int64_t K);
int merlot_gemm_bf16_nt_ln(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                           float alpha, const float* bias, const void* aux_in, int64_t ld_aux_in, float dropout_p, uint64_t dropout_seed,
                           const float* ln_gamma, const float* ln_beta, void* ln_out, int64_t ld_ln, float* ln_mean, float* ln_rstd,
                           float ln_eps, void* ln_workspace, int64_t ln_workspace_bytes, void* workspace, int64_t workspace_bytes,
                           merlot_stream_t stream);

/* Which kernel merlot_gemm_bf16_nt runs for a problem size (the choice depends on M, N, K only); -1 for sizes the
 * entry point rejects.  Tests use it to assert that a shape exercises the kernel they mean to check. */
#define MERLOT_NT_KERNEL_RING_128x256 11    /* gemm_nt_ring_kernel<Cfg<2,4,2,2,32,3>>: 2 workgroups / CU            */
#define MERLOT_NT_KERNEL_RING_256x64 14     /* gemm_nt_ring_kernel<Cfg<4,1,2,2,32,3>>: narrow outputs               */
#define MERLOT_NT_KERNEL_RING_256x128 15    /* gemm_nt_ring_kernel<Cfg<4,1,2,4,32,3>>                               */
/* (20, 21: the lock-step persistent kernels of rounds 1-3, retired in ABI v5) */
#define MERLOT_NT_KERNEL_P8 22              /* gemm_nt_p8_kernel: persistent 256x256, BK 64, two wave groups in ping-pong */
int merlot_gemm_bf16_nt_plan(int64_t M, int64_t N, int64_t K);

/* ---- fp8 (OCP e4m3fn) forward path of BASELINE config #5.  No reference counterpart (the reference's precision policy is
 * bf16 compute / fp32 parameters, utils/model_utils.py:572-602); contract: SURVEY.md 7(vii).
 * merlot_quantize_e4m3: y[rows, cols] (1 byte per element) = e4m3(clamp(x * s, +-448)), s = 448 / max|x| over the whole
 * tensor (per-tensor "current" scaling), x bf16.  scale = device float[3], written: {s, 1/s, max|x|}.
 * cols, ldx, ldy multiples of 8. */
int merlot_quantize_e4m3(const void* x, int64_t rows, int64_t cols, int64_t ldx, void* y, int64_t ldy, float* scale,
                         merlot_stream_t stream);
/* amax[g] = max|x[:, g-th of `groups` equal column groups]| over a [rows, cols] bf16 view (row stride ldx), one pass;
 * amax = device float[groups] (zeroed here), groups <= 4 (groups = 3 over the fused QKV tensor gives max|Q|, max|K|, max|V|). */
int merlot_amax_bf16(const void* x, int64_t rows, int64_t cols, int64_t ldx, int groups, float* amax, merlot_stream_t stream);
/* (round 6: merlot_attention_fwd_fp8 -- Q K^T / P V on the e4m3 MFMA, operands quantised on the fly -- was correct and 2-19 % SLOWER than the bf16 kernel on every
 * shape for three rounds; removed in ABI v9, VERDICT r5 #6 "fix or delete") */
/* C[M,N] = epilogue(alpha * scale_a[0] * scale_b[0] * A8[M,K] * B8t[N,K]^T + bias): merlot_gemm_bf16_nt on e4m3 operands.
 * scale_a / scale_b point at the DEQUANTISATION factor of each operand in device memory (&scale[1] of
 * merlot_quantize_e4m3).  A may instead (or also) carry per-ROW factors row_scale_a (f32 [M], from merlot_ln_fwd_q8;
 * then scale_a may be NULL): C[m, :] is multiplied by row_scale_a[m].  fp32 accumulation; epilogues, bias, aux_in / aux_out, dropout and the C dtypes as for the bf16
 * entry.  K % 128 == 0, K >= 256, lda / ldb in elements (= bytes) and multiples of 16. */
int merlot_gemm_fp8_nt(const void* A8, int64_t lda, const float* scale_a, const float* row_scale_a, const void* B8t, int64_t ldb,
                       const float* scale_b, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                       int epilogue, int out_f32, const float* bias, const void* aux_in, int64_t ld_aux_in,
                       void* aux_out, int64_t ld_aux_out, float dropout_p, uint64_t dropout_seed,
                       void* workspace, int64_t workspace_bytes, merlot_stream_t stream);   /* workspace: as merlot_gemm_bf16_nt, required */

/* Weight gradient: C[M,N] (f32) (+)= alpha * sum_r A[r,M] * B[r,N].  A, B bf16 row-major with the
 * reduction index r as the SLOW dim (activations / output grads as stored).  M, N even.
 * The reduction is split over the grid; partial tiles go through the caller-owned `workspace` (f32, at least
 * merlot_gemm_bf16_tn_workspace_bytes(M, N, R) bytes; may be NULL when that is 0) -- no atomics. */
int64_t merlot_gemm_bf16_tn_workspace_bytes(int64_t M, int64_t N, int64_t R);
int merlot_gemm_bf16_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                        int64_t M, int64_t N, int64_t R, float alpha, int accumulate, void* workspace,
                        int64_t workspace_bytes, merlot_stream_t stream);
/* ABI v8: merlot_gemm_bf16_tn that ALSO accumulates colsum_a[m] += sum_r A[r, m] for m < colsum_m (f32; colsum_a NULL: exactly merlot_gemm_bf16_tn) --
 * the bias gradient that goes with this weight gradient, taken from the operand fragments the ping-pong kernel holds in registers instead of a second
 * pass over A (the Q third of the fused QKV bias, utils/transformer.py:21-25 under tf.gradients); shapes another kernel takes: merlot_colsum_bf16
 * behind the GEMM, same result up to summation order. */
int merlot_gemm_bf16_tn_cs(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                           int64_t M, int64_t N, int64_t R, float alpha, int accumulate, float* colsum_a, int64_t colsum_m,
                           void* workspace, int64_t workspace_bytes, merlot_stream_t stream);

/* ---- ABI v9: the weight gradient on 8-bit float operands (judge row g1 / BASELINE configs[4]; no reference counterpart -- the reference's precision
 * policy is bf16 compute / fp32 parameters, utils/model_utils.py:572-602; the contraction itself is tf.gradients of the dense layers,
 * utils/transformer.py:141-163).
 * merlot_quantize_f8: y[rows_pad, cols] (1 byte per element) = f8(clamp(x * s)), x bf16 [rows, cols]; rows [rows, rows_pad) of y are written as zeros.
 *   fmt 0 = OCP e4m3fn (max 448), 1 = OCP e5m2 (max 57 344).  scale = device float[4] {s, 1/s, amax s was made from, amax of the tensor just quantised}:
 *   delayed = 0 ("current"): s = max / max|x| of THIS tensor (an amax pass, then the convert pass); delayed = 1: one pass, s from the amax the previous call
 *   on this block recorded (scale[3]); the pass records this tensor's amax for the next call; values beyond the old range saturate.  The first call on a
 *   block must be a current one.  cols, ldx, ldy multiples of 8. */
int merlot_quantize_f8(const void* x, int64_t rows, int64_t cols, int64_t ldx, void* y, int64_t ldy, int64_t rows_pad, int fmt, int delayed,
                       float* scale, merlot_stream_t stream);
/* C[M,N] (f32) (+)= alpha * deq_a[0] * deq_b[0] * sum_r A8[r,M] * B8[r,N]: merlot_gemm_bf16_tn on 8-bit operands as stored (reduction index slow),
 * fmt_a / fmt_b 0 = e4m3, 1 = e5m2, deq_a / deq_b the DEQUANTISATION factors in device memory (&scale[1] of merlot_quantize_f8), fp32 accumulation
 * (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales).  R % 128 == 0 and >= 2048 (pad with zero rows: merlot_quantize_f8's rows_pad), M, N >= 128,
 * at most 256 tiles of 256 x 256, lda / ldb in elements (= bytes), multiples of 16 and >= M / N rounded up to 16, N % 4 == 0.  workspace: f32, at least
 * merlot_gemm_f8_tn_workspace_bytes(M, N, R) bytes (0 for shapes the entry refuses). */
int64_t merlot_gemm_f8_tn_workspace_bytes(int64_t M, int64_t N, int64_t R);
int merlot_gemm_f8_tn(const void* A8, int64_t lda, int fmt_a, const float* deq_a, const void* B8, int64_t ldb, int fmt_b, const float* deq_b,
                      float* C, int64_t ldc, int64_t M, int64_t N, int64_t R, float alpha, int accumulate, void* workspace,
                      int64_t workspace_bytes, merlot_stream_t stream);

/* The 8-bit copy from the PRODUCING launch instead of a quantising pass (what makes the 8-bit weight gradient pay: profiles/r06_j_f8_tn.txt):
 * merlot_gemm_bf16_nt with the DGELU epilogue / merlot_gemm_fp8_nt with the GELU epilogue (+ its pre-activation output aux_out) that ALSO write
 * q8_out[M, N] = f8(clamp(bf16(C) * q8_scale[0])) (q8_fmt 0 = e4m3, 1 = e5m2; the fp8-operand entry: e4m3 only) and max |bf16(C)| into q8_scale[3]
 * (atomic max).  C may be NULL: the bf16 output is then not stored at all.  DELAYED scaling: q8_scale[0] comes from an earlier step's amax --
 * merlot_f8_scale_rotate(blocks, n, fmts) on the stream in front of the producers turns every block's recorded amax ([3], if > 0) into its {s, 1/s, amax}
 * and clears the record; a block's first scale comes from merlot_quantize_f8 (current).  N a multiple of 256, operands 16-byte aligned, leading
 * dimensions multiples of 8 (fp8 operands: lda / ldb of 16); other arguments as the base entries. */
int merlot_f8_scale_rotate(float* blocks, int n, const int32_t* fmts, merlot_stream_t stream);
/* C[M,N] (bf16) = alpha * scale_a[0] * scale_b[0] * A8[M,K] * B8t[N,K]^T + bias with A8 in e4m3 (fmt_a 0) or e5m2 (fmt_a 1: a gradient tensor -- the input-gradient
 * GEMM that reads the copy the GELU' epilogue wrote), B8t e4m3; no epilogue option.  Shapes as merlot_gemm_fp8_nt. */
int merlot_gemm_f8_nt(const void* A8, int64_t lda, int fmt_a, const float* scale_a, const void* B8t, int64_t ldb, const float* scale_b,
                      void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, const float* bias, void* workspace,
                      int64_t workspace_bytes, merlot_stream_t stream);

int merlot_gemm_bf16_nt_q8(const void* A, int64_t lda, const void* Bt, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                           float alpha, int epilogue, const float* bias, const void* aux_in, int64_t ld_aux_in, float* colsum_out,
                           void* q8_out, int64_t ld_q8, int q8_fmt, float* q8_scale, void* workspace, int64_t workspace_bytes,
                           merlot_stream_t stream);
int merlot_gemm_fp8_nt_q8(const void* A8, int64_t lda, const float* scale_a, const float* row_scale_a, const void* B8t, int64_t ldb,
                          const float* scale_b, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, int epilogue,
                          const float* bias, void* aux_out, int64_t ld_aux_out, void* q8_out, int64_t ld_q8, int q8_fmt, float* q8_scale,
                          void* workspace, int64_t workspace_bytes, merlot_stream_t stream);

/* Patch-embed 16x16/16 conv (utils/vision_transformer.py:193-205) as im2col + MFMA GEMM.
 * image: bf16 NHWC [n_img, H, W, 3] in [0,1]; patches: bf16 [n_img*(H/P)*(W/P), P*P*3], k = (py,px,c) = HWIO flattening,
 * value = pixel + shift (the `image - 0.5` of :193 is applied here). */
int merlot_im2col_patches(const void* image, void* patches, int n_img, int H, int W, int P, float shift,
                          merlot_stream_t stream);
/* out[rows, hidden] (bf16) = patches(image - 0.5) . Wt^T + bias.  Wt: bf16 [hidden, P*P*3]; `patches` is a caller-owned
 * buffer (see above) that this call FILLS and the weight gradient re-uses. */
int merlot_patch_embed_fwd(const void* image, int n_img, int H, int W, int P, const void* Wt, const float* bias,
                           void* patches, void* out, int hidden, void* workspace, int64_t workspace_bytes,
                           merlot_stream_t stream);                                    /* workspace: as merlot_gemm_bf16_nt */
/* dWt[hidden, K] (f32) (+)= sum_rows dY[row, hidden] * patches[row, k]  (K = P*P*3; workspace as for
 * merlot_gemm_bf16_tn(hidden, K, rows)). */
int merlot_patch_embed_wgrad(const void* patches, int64_t rows, int K, const void* dY, float* dWt, int hidden,
                             int accumulate, void* workspace, int64_t workspace_bytes, merlot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (utils/model_utils.py:113-130): fp32 statistics, population variance, eps inside rsqrt.
 * ---------------------------------------------------------------------------------------------- */
/* y = LN(x) ; x is bf16 (x_f32=0) or f32 ; y_bf16 and/or y_f32 may be NULL ; mean/rstd f32 [rows]
 * (may be NULL when no backward is needed).  H % 256 == 0, H <= 2048. */
int merlot_ln_fwd(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                  float* mean, float* rstd, int64_t rows, int H, float eps, merlot_stream_t stream);
/* merlot_ln_fwd that ALSO emits the e4m3 copy the fp8 GEMM behind this LayerNorm consumes (config #5: QKV, fc1):
 * y_fp8[rows, H] = e4m3(bf16(y) * 448 / max|bf16(y[row])|), row_scale[row] = max|.| / 448 (per-ROW scaling: the wave that
 * normalises a row owns all of it).  y_bf16 may be NULL when no consumer needs it. */
int merlot_ln_fwd_q8(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, void* y_fp8,
                     float* row_scale, float* mean, float* rstd, int64_t rows, int H, float eps, merlot_stream_t stream);
/* ABI v9: the same with ONE per-tensor factor: y_fp8 = e4m3(clamp(bf16(y) * tscale[0])) -- tscale = the tensor's merlot_quantize_f8 block (delayed scaling:
 * merlot_f8_scale_rotate in front), max|bf16(y)| of this launch is max-ed into tscale[3].  The copy both the e4m3 forward GEMM (scale_a = &tscale[1]) and the
 * 8-bit weight gradient (whose reduction index a per-row factor would lie along) read as stored. */
int merlot_ln_fwd_q8t(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, void* y_fp8, float* tscale,
                      float* mean, float* rstd, int64_t rows, int H, float eps, merlot_stream_t stream);
/* dx = LN'(dy) (+ dres) ; dgamma/dbeta (f32 [H]) are ACCUMULATED with atomics.  dy, x, dres, dx each bf16
 * or f32 per flag; dres may be NULL.
 * Optional fused tail for the residual stream (all NULL/0 to disable): dcolsum[H] += column sums of d_branch, where
 * d_branch = dx when drop_p == 0, else d_branch = dropout'(dx) with the (drop_seed, row*H+col) mask of the forward
 * MERLOT_EPI_RESIDUAL epilogue, also written to dx_drop (bf16 [rows, H]).  d_branch is the gradient of the previous
 * sub-layer's branch output and dcolsum its bias gradient (utils/transformer.py:136,162,214,220). */
int merlot_ln_bwd(const void* dy, int dy_f32, const void* x, int x_f32, const float* mean, const float* rstd,
                  const float* gamma, const void* dres, int dres_f32, void* dx, int dx_f32, float* dgamma,
                  float* dbeta, int64_t rows, int H, void* dx_drop, float drop_p, uint64_t drop_seed,
                  float* dcolsum, merlot_stream_t stream);
/* ABI v9: merlot_ln_bwd that ALSO writes db8[rows, H] = f8(clamp(d_branch * db8_scale[0])) -- the 8-bit float copy (db8_fmt 0 = e4m3, 1 = e5m2) of the branch
 * gradient as the next kernels read it (bf16-rounded dx, behind the dropout mask when drop_p > 0): the A operand of that sub-layer's 8-bit weight gradient
 * (merlot_gemm_f8_tn) without a quantising pass.  db8_scale = the tensor's merlot_quantize_f8 block (delayed scaling), max|d_branch| goes to [3]. */
int merlot_ln_bwd_q8(const void* dy, int dy_f32, const void* x, int x_f32, const float* mean, const float* rstd,
                     const float* gamma, const void* dres, int dres_f32, void* dx, int dx_f32, float* dgamma,
                     float* dbeta, int64_t rows, int H, void* dx_drop, float drop_p, uint64_t drop_seed,
                     float* dcolsum, void* db8, int db8_fmt, float* db8_scale, merlot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention (utils/transformer.py:98-127), head_dim = 64.
 * qkv: bf16 [B*S, ld] rows = tokens; within a row q | k | v each `heads*64` wide, head h at h*64.
 * valid: uint8 [B,S] or NULL (=all valid).  mask(b,i,j) = valid[b,i] & valid[b,j]; a masked score is
 * exactly -1e10 (NOT -inf): a padded query row attends uniformly over all S keys (:109-112).
 * ---------------------------------------------------------------------------------------------- */
/* seg (optional, needs valid): int32 [S] segment id per position, shared by all batch rows -- the
 * `disable_pairwise_lang_attn` block mask of model/modeling.py:160-168: a pair of valid tokens is additionally masked
 * (score exactly -1e10) unless seg[q] == seg[k] or one of the two is 0. */
/* colsum_lo / colsum_hi (optional, f32 [B,S], ACCUMULATED): the side outputs of merlot_attention_colsum below, produced by
 * the same launch (S <= 512: from the K tile still resident in LDS; longer sequences: a second pass); needs lse. */
/* workspace (ABI v7; optional): merlot_attention_workspace_bytes() bytes, 4-byte aligned, ZERO on entry and left zero -- the item-claim
 * counters of the persistent kernels (csrc/attention_pp.inc: one workgroup per CU walks the (batch, head) items with the next items'
 * operands in flight; unmasked 65 .. 224 tokens forward and backward, masked 257 .. 352 tokens forward without side outputs).  Caller-owned
 * like the GEMMs' (no library state): one block per concurrently used stream -- two launches in flight on ONE block are undefined
 * behaviour (the kernels carry no launch epoch: a non-zero counter makes a launch skip its first items, silently).  A launch that returns
 * MERLOT_ELAUNCH has had its block cleared again by the library (a memset on `stream`); after a device fault the caller clears it.
 * Without it (NULL) the one-launch-per-item kernels run. */
int64_t merlot_attention_workspace_bytes(void);
int merlot_attention_fwd(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, const uint8_t* valid,
                         const int32_t* seg, int B, int S, int heads, float scale, float* colsum_lo, float* colsum_hi,
                         int qsplit, int valid_q_only, float weight, void* workspace, int64_t workspace_bytes,
                         merlot_stream_t stream);
/* dqkv (bf16, same layout as qkv) from dout.  delta: f32 workspace [B*heads*S] (written by the dQ kernel, read by the dK/dV kernel).
 * log_lo / log_hi (optional, f32 [B,S], ACCUMULATED; ABI v5): merlot_attention_colsum's valid_q_only = 1 output (the attention LOG of
 * model/modeling.py:186-203: per-key sums of P over the valid query rows below / from log_qsplit, times log_weight) taken from the
 * BACKWARD, where P is recomputed anyway -- S <= 512 without a segment mask: two lane accumulators of the fused kernel's dK / dV
 * pass; otherwise the tiled column-sum kernel is launched here.  A training step that only needs the log at its end passes the buffers
 * here and calls merlot_attention_fwd without them. */
int merlot_attention_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo,
                         const float* lse, const uint8_t* valid, const int32_t* seg, void* dqkv, int64_t lddqkv,
                         float* delta, int B, int S, int heads, float scale, float* log_lo, float* log_hi, int log_qsplit,
                         float log_weight, void* workspace, int64_t workspace_bytes, merlot_stream_t stream);
/* ABI v9: merlot_attention_bwd that ALSO writes dqkv8[B*S, lddqkv8] = f8(clamp(bf16(dqkv) * q8_scale[0])) (q8_fmt 0 = e4m3, 1 = e5m2; q8_scale = the tensor's
 * merlot_quantize_f8 block, delayed scaling; max|dqkv| of the launch is max-ed into q8_scale[3]) -- the operand of the QKV weight / input gradients on 8-bit
 * operands without a quantising pass.  Only the shapes the tiled dQ / dK dV kernel pair takes have it: merlot_attention_bwd_writes_q8(S, has_segment_mask) == 1
 * (more than 512 tokens -- BASELINE configs[4]'s 578 / 2 832 --, a segment mask, or at most 64 tokens); other shapes are refused (use the plain entry and
 * merlot_quantize_f8). */
int merlot_attention_bwd_writes_q8(int S, int has_segment_mask);
int merlot_attention_bwd_q8(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t lddo,
                            const float* lse, const uint8_t* valid, const int32_t* seg, void* dqkv, int64_t lddqkv, float* delta,
                            int B, int S, int heads, float scale, float* log_lo, float* log_hi, int log_qsplit, float log_weight,
                            void* dqkv8, int64_t lddqkv8, int q8_fmt, float* q8_scale,
                            void* workspace, int64_t workspace_bytes, merlot_stream_t stream);
/* Side outputs the reference takes from its stacked [B,layers,S,S] head-mean probabilities, without
 * materialising SxS:  colsum_lo[b,key] += weight * sum_h sum_{q <  qsplit} P[b,h,q,key]
 *                     colsum_hi[b,key] += weight * sum_h sum_{q >= qsplit} P[b,h,q,key]
 * valid_q_only=0: every query row counts (padded rows attend uniformly) -- the attention_summs that feed
 *   masking, model/modeling.py:428-431 (use qsplit = S, colsum_hi = NULL);
 * valid_q_only=1: only (valid query, valid key) pairs count -- the four viz/lang block sums of the
 *   attention log, model/modeling.py:186-203 (qsplit = P, the host sums key ranges). */
int merlot_attention_colsum(const void* qkv, int64_t ld, const float* lse, const uint8_t* valid, const int32_t* seg,
                            float* colsum_lo, float* colsum_hi, int qsplit, int valid_q_only, float weight, int B, int S,
                            int heads, float scale, merlot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / gather / reduction helpers.
 * ---------------------------------------------------------------------------------------------- */
int merlot_cast_f32_bf16(const float* src, void* dst, int64_t n, merlot_stream_t stream);
/* dst[C, ld_dst] (bf16) : dst[c, r] = src[r, c] for the f32 matrix src[R,C]; ld_dst >= R */
int merlot_cast_transpose_f32_bf16(const float* src, void* dst, int64_t R, int64_t C, int64_t ld_dst,
                                   merlot_stream_t stream);
/* the same for many matrices in one launch.  jobs: device int64 [njobs][6] = {src offset (elements from src_base), dst
 * offset (elements from dst_base), R, C, ld_dst, index of the job's first 64x64 tile}, sorted by first tile;
 * total_tiles = sum over jobs of ceil(R/64)*ceil(C/64). */
int merlot_cast_transpose_batched(const float* src_base, void* dst_base, const void* jobs, int njobs,
                                  int64_t total_tiles, merlot_stream_t stream);
/* out[N] (+)= sum_t x[t, n]   (bias gradients) ; x bf16 */
int merlot_colsum_bf16(const void* x, int64_t ld, float* out, int64_t T, int64_t N, int accumulate,
                       merlot_stream_t stream);
/* out[r,:] = a[ia[r],:] + b[ib[r],:] + c[ic[r],:] + d[id[r],:]  (f32 tables H wide; a NULL table or an
 * index < 0 contributes 0; a NULL index array means identity).  `a` may be bf16 (a_bf16=1).  The embedding sums
 * at model/modeling.py:262-297 (word + position), :299-337 (pooled grid + img_idx_pe + final_pe) and
 * utils/vision_transformer.py:231-233 (patches + cls/pos embeddings), as gathers -- never one-hot matmuls. */
int merlot_gather_add4(const void* a, int a_bf16, const int32_t* ia, const float* b, const int32_t* ib,
                       const float* c, const int32_t* ic, const float* d, const int32_t* id, float* out, int64_t rows,
                       int H, merlot_stream_t stream);
/* y = dropout(x) on bf16 [rows, N] with the same counter-based keep mask (seed, r*N+n) as the GEMM
 * MERLOT_EPI_RESIDUAL epilogue; used for the backward of that epilogue and for the embedding dropout
 * (utils/model_utils.py:335-349, model/modeling.py:294). */
int merlot_dropout_apply(const void* x, void* y, int64_t rows, int64_t N, float p, uint64_t seed,
                         merlot_stream_t stream);
/* table[idx[r],:] += src[r,:]  (f32, atomics) ; idx<0 skipped.  Backward of the gathers. */
int merlot_scatter_add_rows(const float* src, const int32_t* idx, float* table, int64_t rows, int H,
                            merlot_stream_t stream);
/* The same for MANY rows onto FEW distinct table rows (the word-embedding gradient: model/modeling.py:262-296's gather, backward): the
 * caller sorts -- perm = a stable argsort of idx (int32), sorted_idx = idx[perm] -- and rows of one index are summed in registers
 * and written once; atomics only where a run of equal indices crosses a block boundary.  Negative indices are skipped.  H % 4 == 0. */
int merlot_scatter_add_sorted(const float* src, const int32_t* perm, const int32_t* sorted_idx, float* table, int64_t rows, int H,
                              merlot_stream_t stream);
/* 2x2 VALID average pool of a [n_img, h1, w1, H] bf16 grid taken from rows [n, cls_skip + h*w1 + w] of
 * x[n_img, S, H]; writes f32 [n_img, 1 + h2*w2, H] with row 0 = x[n,0,:] (the CLS slot)
 * (utils/vision_transformer.py:251-267 + model/modeling.py:101-105). */
int merlot_cls_avgpool_fwd(const void* x, float* out, int n_img, int h1, int w1, int cls_skip, int pool, int H,
                           merlot_stream_t stream);
int merlot_cls_avgpool_bwd(const float* dout, void* dx, int n_img, int h1, int w1, int cls_skip, int pool, int H,
                           merlot_stream_t stream);

/* Row softmax cross-entropy (utils/model_utils.py:313-332): logits f32 [rows, ld] (first C columns
 * used), labels int32.  loss[r] = -log_softmax(logits[r])[label[r]] ; argmax[r] (first max).
 * If dlogits != NULL: dlogits[r, :C] = rowscale[r] * (softmax - onehot) written as bf16 (dl_bf16=1,
 * leading dim ld_dl, columns C..ld_dl zero-filled) or f32. */
int merlot_softmax_ce(const float* logits, int64_t ld, const int32_t* labels, float* loss, int32_t* argmax,
                      const float* rowscale, void* dlogits, int dl_bf16, int64_t ld_dl, int64_t rows, int C,
                      merlot_stream_t stream);

/* MLM head tail as ONE call (model/modeling.py:217-223 logits = h . E^T + output_bias, :539-545 cross-entropy; SURVEY 8(b)
 * `merlot_vocab_ce`): logits GEMM (bf16 operands, fp32 accumulate, fp32 logits) + merlot_softmax_ce through the caller's `scratch`
 * (fp32 logits, rows x ceil64(V) floats).  merlot_vocab_ce_scratch_bytes(T, V) = room for all T rows (one GEMM + one softmax launch,
 * the fastest form: 3.7 ms at 12 800 x 50 370); with less scratch (>= 256 rows) the rows are processed in chunks -- cache-sized
 * chunks were measured and are slower (profiles/r03_f_vocab_ce.txt), they exist for memory-constrained callers.
 * h: bf16 [T, K]; table: bf16 [>= V rows, K] (the tied word-embedding table); out_bias f32 [V] or NULL; targets int32 [T];
 * rowscale f32 [T] or NULL; outputs loss f32 [T], argmax int32 [T] (or NULL), dlogits bf16 [T, ld_dl] (= rowscale * (softmax -
 * onehot), columns V..ld_dl zero-filled; or NULL).  nt_workspace: as merlot_gemm_bf16_nt. */
int64_t merlot_vocab_ce_scratch_bytes(int64_t T, int64_t V);
int merlot_vocab_ce_fwd(const void* h, int64_t ldh, const void* table, int64_t ldt, const float* out_bias, const int32_t* targets,
                        const float* rowscale, float* loss, int32_t* argmax, void* dlogits, int64_t ld_dl, int64_t T, int64_t V,
                        int64_t K, void* scratch, int64_t scratch_bytes, void* nt_workspace, int64_t nt_workspace_bytes,
                        merlot_stream_t stream);

/* x * rsqrt(max(sum x^2, 1e-12))  (tf.math.l2_normalize, model/modeling.py:43) ; f32 [rows, H] */
int merlot_l2norm_fwd(const float* x, float* y, float* inv_norm, int64_t rows, int H, merlot_stream_t stream);
int merlot_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t rows, int H,
                      merlot_stream_t stream);
/* exact erf GELU on f32 buffers (heads) */
int merlot_gelu_fwd(const float* x, float* y, int64_t n, merlot_stream_t stream);
int merlot_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, merlot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Integer / index work (bit-exact given the explicit noise inputs).
 * ---------------------------------------------------------------------------------------------- */
/* model/modeling.py:381-489 + utils/model_utils.py:640-649.  One workgroup per group row.
 * ids [B,L] int32; attention_summs f32 [B,L] (NULL => masking_use_attn False); gumbel f32 [B,L];
 * span_lower/upper int32 [B,nm] (NULL => no spanbert); random_ids, option int32 [B,L];
 * log_nontopk/log_topk = float32 log() of the two mask weights, w_nontopk/w_topk the weights,
 * max_weight = reduce_max(mask_weight) over the whole batch (modeling.py:468; known on the host).
 * outputs masked_ids [B,L], masked_idx [B,nm] (ascending).  L <= 1024. */
int merlot_mask_inputs(const int32_t* ids, const float* attention_summs, const float* gumbel,
                       const int32_t* span_lower, const int32_t* span_upper, const int32_t* random_ids,
                       const int32_t* option, int32_t* masked_ids, int32_t* masked_idx, int B, int L, int num_topk,
                       int num_to_mask, float w_nontopk, float w_topk, float log_nontopk, float log_topk,
                       float max_weight, int mask_token, merlot_stream_t stream);
/* model/modeling.py:598-620, 635, 649-652: labels int32 [B*n*n], weights f32 [B*n*n]. */
int merlot_temporal_labels(const int32_t* video_src_ids, const int32_t* shuffled_idx, int32_t* labels,
                           float* weights, int B, int n, merlot_stream_t stream);
/* model/dataloader.py:224-257: shuffled_idx [B*n] from explicit draws. */
int merlot_shuffled_idx(const int32_t* num_shuffle, const float* u_select, const float* u_perm, int32_t* out, int B,
                        int n, int shuffle_offset, merlot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * AdamW with bias correction and bf16 m / sign-encoded bf16 v (utils/optimization.py:267-288,339-416).
 * param/grad f32 [n]; m, v bf16 bit patterns (state_bf16=1) or f32.  lr already includes the schedule
 * scale and sqrt(bc2)/bc1.  beta1/beta2 are doubles so that (1 - beta) is rounded to f32 from the double value,
 * as the reference's python-side `1.0 - beta_1` is.  grad_scale multiplies the gradient first (1/world for mean).
 * wd_flags: NULL (decay everywhere) or one byte per 64 consecutive elements, 0 = no weight decay for that chunk --
 * the regex `param_overrides` of utils/optimization.py:125-151 flattened onto the 64-element-aligned arena, so the
 * whole model updates in ONE launch.
 * ---------------------------------------------------------------------------------------------- */
int merlot_adamw_step(float* param, const float* grad, void* m, void* v, int64_t n, float lr, double beta1,
                      double beta2, float eps, float weight_decay, float grad_scale, const uint8_t* wd_flags,
                      int state_bf16, merlot_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ResNet-hybrid stem pieces: utils/vision_transformer.py:8-170 (conv2d_fixed_padding,
 * bottleneck_block, lite_resnet50) and utils/model_utils.py:133-222 (group_norm).  Activations NHWC bf16, C % 8 == 0
 * (C == 3 allowed for the first convolution's im2col).  The convolutions themselves run on merlot_gemm_bf16_nt /
 * merlot_gemm_bf16_tn: 1x1 directly on [N*H*W, C], 3x3 on the im2col matrix.
 * ---------------------------------------------------------------------------------------------- */
/* out[(n,yo,xo)][(ky,kx,c)] = x[n][yo*s+ky-1][xo*s+kx-1][c] + shift (0 outside the image), row stride Kp >= 9*C,
 * columns 9*C..Kp-1 zero.  Padding (1,1) = TF SAME at stride 1 = fixed_padding(3) + VALID at stride 2 (:8-19,43). */
int merlot_im2col3x3(const void* x, void* out, int N, int H, int W, int C, int stride, int Kp, float shift,
                     merlot_stream_t stream);
/* input gradient of the same gather: dx[n][y][x][c] = sum over taps of dpatches[(n,yo,xo)][(ky,kx,c)]. */
int merlot_col2im3x3(const void* dpatches, void* dx, int N, int H, int W, int C, int stride, int Kp,
                     merlot_stream_t stream);
/* 3x3 convolution, stride 1, SAME padding, as an IMPLICIT GEMM (csrc/conv_gemm.hip): the gather of merlot_im2col3x3 happens in the
 * kernel's LDS-DMA source addresses, no patch matrix is written.  Replaces tf.layers.conv2d(kernel 3, strides 1, padding SAME) of
 * utils/vision_transformer.py:40-56 for C % 32 == 0 (every 3x3 convolution of the released stem except the root).
 *   y[(n,yo,xo)][co] = sum over (ky,kx,c) of x[n][yo+ky-1][xo+kx-1][c] * w[co][(ky,kx,c)]
 * x: [n_img,H,W,C] bf16; w: [Co, ldw] bf16 with ldw >= 9*C (the wb operand of merlot_weight_std_fwd); y: [n_img*H*W, ldy] bf16;
 * Co % 8 == 0.  Same K order and MFMA sequence as merlot_im2col3x3 + merlot_gemm_bf16_nt on the same tile: bit-identical to it.
 * The layer's input gradient is the same call on dY with w' = [C][(2-ky, 2-kx, co)] (taps flipped, channels swapped).
 * `zeros`: >= 16 bytes of zeroed device memory, 16-B aligned (the source of a tap that leaves the image). */
int merlot_conv3x3_bf16(const void* x, const void* w, int64_t ldw, void* y, int64_t ldy, int n_img, int H, int W, int C, int Co,
                        const void* zeros, merlot_stream_t stream);
/* Weight gradient of the same layer, implicit as well (no patch matrix): dw[co][(ky,kx,c)] (+)= sum over pixels of
 * dy[pix][co] * x[pix shifted by the tap][c], fp32 [Co, lddw], lddw >= 9*C.  dy: [n_img*H*W, lddy] bf16.  The pixel axis is split over
 * workgroups; partial tiles go to `workspace` (merlot_conv3x3_wgrad_workspace_bytes(), fp32, 16-B aligned, caller-owned) and are folded
 * without atomics, like merlot_gemm_bf16_tn.  Replaces the kernel gradient of tf.layers.conv2d (utils/vision_transformer.py:40-56). */
int64_t merlot_conv3x3_wgrad_workspace_bytes(int n_img, int H, int W, int C, int Co);
int merlot_conv3x3_wgrad_bf16(const void* dy, int64_t lddy, const void* x, float* dw, int64_t lddw, int n_img, int H, int W, int C,
                              int Co, int accumulate, const void* zeros, void* workspace, int64_t workspace_bytes,
                              merlot_stream_t stream);
/* Weight standardisation of the hybrid stem's kernels (utils/vision_transformer.py:52-56): per output channel over (kh, kw, ci),
 * population variance, eps 1e-5.  k: fp32 master, HWIO = [K, Co].  Writes khat fp32 [K, Co], rstd [Co], the NT operand
 * wb bf16 [Co, Kp] and the dgrad operand wbT bf16 [Kp, Cop] (paddings untouched: zero them once). */
int merlot_weight_std_fwd(const float* k, int K, int Co, float* khat, float* rstd, void* wb, int Kp, void* wbT, int Cop,
                          merlot_stream_t stream);
/* gk[K, Co] += rstd * (dkhat - mean_K(dkhat) - khat * mean_K(dkhat * khat)); dkhat_t = the wgrad GEMM's output, [Co, ld]. */
int merlot_weight_std_bwd(const float* dkhat_t, int64_t ld, const float* khat, const float* rstd, int K, int Co, float* gk,
                          merlot_stream_t stream);
/* Every kernel of the stem in ONE launch each way (52 kernels of a few KB to a few MB).  jobs: device int64 table, offsets in elements
 * from the respective base.  Forward job = {k, K, Co, khat, rstd, wb, Kp, wbT, Cop, wdg (< 0: none), Cin, first block}; wdg (3x3 kernels) is
 * the operand of the input gradient as an implicit convolution of dY: wdg[ci][(2-ky, 2-kx, co)] = khat[(ky,kx,ci)][co], bf16 [Cin, 9 Co].
 * Backward job = {dkhat_t, ld, khat, rstd, K, Co, gk, first block}; total_blocks = sum over jobs of ceil(Co / 16). */
int merlot_weight_std_fwd_batched(const float* k_base, const void* jobs, int njobs, int64_t total_blocks, float* khat_base,
                                  float* rstd_base, void* wb_base, void* wbT_base, void* wdg_base, merlot_stream_t stream);
int merlot_weight_std_bwd_batched(const float* dk_base, const void* jobs, int njobs, int64_t total_blocks, const float* khat_base,
                                  const float* rstd_base, float* gk_base, merlot_stream_t stream);
/* y = [relu]( (x - mean) * rsqrt(var + eps) * gamma + beta [+ res] ), moments per (sample, group) over (H, W, C/G) from
 * one pass (var = E[x^2] - E[x]^2, :196-201).  stats: f32 [N, G, 2] = {mean, rsqrt(var + eps)}, written here, kept
 * for the backward.  res may be NULL. */
int merlot_groupnorm_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* stats,
                         int N, int H, int W, int C, int G, float eps, int relu, merlot_stream_t stream);
/* dy' = relu ? dy * (y > 0) : dy.  dgamma / dbeta (f32 [C]) are ACCUMULATED; gsum: f32 [N, G, 2] scratch; dx bf16;
 * dres (optional) = dy' for the residual branch.  y may be NULL for a relu layer WITHOUT a residual add (ABI v5): the mask is then
 * recomputed from x as (x - mean) * rstd * gamma + beta > 0, the forward's own expression -- one tensor pass less in each of the two
 * backward kernels; beta is read only for that. */
int merlot_groupnorm_bwd(const void* dy, const void* y, const void* x, const float* stats, const float* gamma, const float* beta,
                         float* dgamma, float* dbeta, float* gsum, void* dx, void* dres, int N, int H, int W, int C, int G,
                         float eps, int relu, merlot_stream_t stream);
/* GroupNorm in ONE launch per direction (ABI v10; same results as the two entries above up to the order of the moment sums): every workgroup
 * keeps its slice of a sample in registers between "reduce the moments" and "normalise", the slices of a sample meet at a counter in the
 * caller-owned workspace -- x is read once forward and x | dy once backward (5 tensor passes per layer instead of 8; the as-shipped
 * 192 x 352 hybrid stem of model/configs/merlot.yaml:30,36 spends 17 % of its step in GroupNorm, utils/model_utils.py:133-222).
 * ws: merlot_groupnorm_fused_workspace_bytes(N, C, G) bytes (4 KiB + the sample's sums per sample: the arrival counters of consecutive samples lie in
 * different memory channels), 16-B aligned, any content (zeroed by the call, on `stream`); one block per call in flight.  All other arguments as in
 * merlot_groupnorm_fwd / merlot_groupnorm_bwd.
 * The backward's workspace (ABI v11) is merlot_groupnorm_bwd_fused_workspace_bytes(N, H, W, C, G): besides the control blocks one slot of (C + G) float pairs per
 * (sample, slice) -- every workgroup STORES its slice's sums there and arrives with one atomic, the sample's last arriver adds the slots up (ABI v10's backward sent four
 * returning device-scope atomics per channel and workgroup: x1.4 ... x15 the time of merlot_groupnorm_bwd, profiles/r06_z2_gn_fused_shapes.txt).  MEASURED, v11
 * (profiles/r06_z5_gn_fused_bwd.txt): same gradients, x0.93 ... x2.2 the time of merlot_groupnorm_bwd (faster on one shape of thirteen), and above 64 slices per sample (its resident
 * workgroups per XCD) calls stall or never return -- see csrc/conv.hip GN_FUSED_MAX_SLICES: both one-launch entries refuse more than 40 slices per sample (MERLOT_ESHAPE).
 * merlot_groupnorm_bwd_fused is exported for the A/B and value-tested; merlot_amd does not call it.
 * Measured at the thirteen GroupNorm shapes of the as-shipped stem, 896 frames (profiles/r06_z3_gn_fused_fwd.txt, r06_z4_gn_fused_fwd.txt): the FORWARD is faster than
 * merlot_groupnorm_fwd on every shape (-3 ... -50 %) and is what merlot_amd runs. */
int64_t merlot_groupnorm_fused_workspace_bytes(int N, int C, int G);
int64_t merlot_groupnorm_bwd_fused_workspace_bytes(int N, int H, int W, int C, int G);
int merlot_groupnorm_fwd_fused(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* stats,
                               int N, int H, int W, int C, int G, float eps, int relu, void* ws, int64_t ws_bytes, merlot_stream_t stream);
int merlot_groupnorm_bwd_fused(const void* dy, const void* y, const void* x, const float* stats, const float* gamma, const float* beta,
                               float* dgamma, float* dbeta, float* gsum, void* dx, void* dres, int N, int H, int W, int C, int G,
                               float eps, int relu, void* ws, int64_t ws_bytes, merlot_stream_t stream);
/* tf.nn.avg_pool2d(ksize 2, strides 2) on even H, W; the backward takes dy [N, H/2, W/2, C] and writes dx [N, H, W, C]. */
int merlot_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, merlot_stream_t stream);
int merlot_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, merlot_stream_t stream);

/* ---- input pipeline: frame preprocessing ---------------------------------------------------------
 * One job per frame: decoded JPEG (HWC uint8, device) -> convert_image_dtype * resize_images(method, align_corners=True)
 * to [scaled_h, scaled_w] -> crop at (offset_y, offset_x) -> zero-pad to [out_h, out_w] -> where(is_finite) ->
 * brightness / contrast augment + clip -> bf16 (model/dataloader.py:72-97, utils/model_utils.py:758-940).  The random
 * quantities (scale, offsets, method, augment choice and factors) are drawn by the host and arrive in the job. */
typedef struct {
    int64_t src_offset;           /* bytes into `src` of this frame's image */
    int32_t src_h, src_w;
    int32_t scaled_h, scaled_w;
    int32_t method;               /* tf.image.ResizeMethod: 0 bilinear, 1 nearest, 2 bicubic, 3 area */
    int32_t offset_y, offset_x;
    int32_t aug_kind;             /* 0 none, 1 brightness, 2 contrast */
    float factor[3];
    float reserved;
} merlot_image_job_t;
int64_t merlot_image_frames_workspace_bytes(int n_img, int out_h, int out_w);
/* jobs_host / jobs_dev: the same n_img-entry table in host memory (validated before any launch) and device memory (read
 * by the kernels).  dst: bf16 [n_img, out_h, out_w, 3]. */
int merlot_image_frames(const uint8_t* src, int64_t src_bytes, const merlot_image_job_t* jobs_host,
                        const merlot_image_job_t* jobs_dev, int n_img, void* dst, int out_h, int out_w, void* workspace,
                        int64_t workspace_bytes, merlot_stream_t stream);

/* ---- JPEG decode (`tf.image.decode_jpeg(x, channels=3)`, model/dataloader.py:72-77) split host / GPU ------------------------
 * Host: markers + Huffman decode of a baseline 8-bit YCbCr JPEG (4:4:4 or 4:2:0, one interleaved scan, restart intervals ok)
 * into QUANTISED coefficients, natural order, component planes of whole blocks.  GPU: dequantise, inverse DCT (libjpeg's
 * jidctint "islow"), h2v2 "fancy" chroma upsampling, YCbCr -> RGB -- bit for bit libjpeg(-turbo)'s default decoder.
 * MERLOT_JPEG_UNSUPPORTED (progressive, grayscale, CMYK, 4:2:2, 12-bit, ...) means: decode this frame with the host library. */
#define MERLOT_JPEG_MALFORMED (-20)
#define MERLOT_JPEG_UNSUPPORTED (-21)
#define MERLOT_JPEG_CAPACITY (-22)
typedef struct {
    int32_t width, height;
    int32_t subsampling;      /* 1: 4:4:4, 2: 4:2:0 */
    int32_t blocks_w[3], blocks_h[3];   /* per component, whole MCUs */
    int64_t coef_offset[3];   /* element offset of the component's first block in the coefficient array */
    int64_t coef_count;       /* int16 elements in total */
    int64_t coef_base;        /* set by the CALLER: element offset of this image's coefficients in the batch buffer */
    int64_t dst_offset;       /* set by the CALLER: byte offset of this image's RGB output (height * width * 3 bytes) */
    int64_t plane_offset;     /* set by the CALLER: byte offset of this image's component planes in the workspace */
    uint16_t quant[3][64];    /* natural order */
} merlot_jpeg_info_t;
/* coef == NULL: fill `info` only (coef_count says how much room the call needs).  Returns MERLOT_OK or a MERLOT_JPEG_* code. */
int merlot_jpeg_entropy_decode(const uint8_t* data, int64_t n, merlot_jpeg_info_t* info, int16_t* coef, int64_t coef_capacity);
/* bytes of the component planes of one image: sum over components of blocks_w * 8 * blocks_h * 8 */
int64_t merlot_jpeg_plane_bytes(const merlot_jpeg_info_t* info);
/* n_img images: coef (device int16), infos_host / infos_dev (the same table), workspace (device, the component planes),
 * dst (device uint8): image i is written as [height, width, 3] RGB at dst + infos[i].dst_offset. */
int merlot_jpeg_idct_rgb(const int16_t* coef, const merlot_jpeg_info_t* infos_host, const merlot_jpeg_info_t* infos_dev, int n_img,
                         uint8_t* workspace, int64_t workspace_bytes, uint8_t* dst, int64_t dst_bytes, merlot_stream_t stream);

/* ---- host-side byte work (no GPU, no stream) ---------------------------------------------------------------------
 * CRC-32C (Castagnoli) of `n` bytes, extending `crc` (0 to start).  Used by the TF tensor-bundle checkpoint reader/writer
 * (the files utils/model_utils.py:388-413 and model/modeling.py:724-738 initialise from) and by the TFRecord framing
 * (data/process.py:236-256).  Returns the unmasked checksum in the low 32 bits; force_sw != 0 selects the table path. */
int64_t merlot_crc32c(uint64_t crc, const void* data, int64_t n, int force_sw);

/* One pass over a serialized tf.train.Example (the records of data/process.py:236-256; `_decode_record`,
 * model/dataloader.py:33-54).  rows[i] = {key offset, key length, kind (1 bytes, 2 float, 3 int64, 0 empty), a, b, count}:
 * bytes: a / b = offset / length of the first value; float / int64: a = first index into fvals / ivals.  Returns the
 * number of features, -1 for malformed input, -2 if an output array is too small. */
int64_t merlot_example_index(const void* buf, int64_t n, int64_t* rows, int64_t max_rows, int64_t* ivals, int64_t max_ivals,
                             float* fvals, int64_t max_fvals);

#ifdef __cplusplus
}
#endif
#endif /* MERLOT_HIP_H */
