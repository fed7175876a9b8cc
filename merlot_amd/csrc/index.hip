// Integer / index work of the MERLOT hot path on the GPU -- bit-exact given the explicit noise inputs.
// TF semantics reproduced: top_k = descending, ties -> lower index (rank by counting); argmax = first max;
// sort ascending.  Floating point is restricted to single IEEE fp32 operations in the reference's order
// (no contraction) so comparisons cannot flip.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int MAXL = 1024;

// rank of element l in a descending, index-stable order
__device__ __forceinline__ int desc_rank(const float* v, int L, int l) {
    const float x = v[l];
    int r = 0;
    for (int j = 0; j < L; ++j) {
        const float y = v[j];
        r += (y > x || (y == x && j < l)) ? 1 : 0;
    }
    return r;
}

__global__ __launch_bounds__(256) void mask_inputs_kernel(
    const int32_t* __restrict__ ids, const float* __restrict__ summs, const float* __restrict__ gumbel,
    const int32_t* __restrict__ span_lower, const int32_t* __restrict__ span_upper,
    const int32_t* __restrict__ random_ids, const int32_t* __restrict__ option, int32_t* __restrict__ masked_ids,
    int32_t* __restrict__ masked_idx, int L, int num_topk, int nm, float w_nontopk, float w_topk, float log_nontopk,
    float log_topk, float max_weight, int mask_token) {
    __shared__ float key[MAXL];
    __shared__ float weight[MAXL];
    __shared__ float special[MAXL];
    __shared__ int sel[MAXL];     // idx list (reversed top-k), later span start
    __shared__ int sel_end[MAXL];
    __shared__ int flag[MAXL];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int32_t* idr = ids + (int64_t)b * L;

    // ---- modeling.py:423-437 : attention top-k -> mask weights
    for (int l = tid; l < L; l += 256) {
        const float sp = idr[l] < 100 ? 1.0f : 0.0f;
        special[l] = sp;
        key[l] = summs ? summs[(int64_t)b * L + l] * (1.0f - sp) : 0.0f;
    }
    __syncthreads();
    for (int l = tid; l < L; l += 256) {
        const bool important = summs && (desc_rank(key, L, l) < num_topk);
        weight[l] = important ? w_topk : w_nontopk;
        flag[l] = important ? 1 : 0;
    }
    __syncthreads();
    // ---- :442-445 + model_utils.py:640-649 : Gumbel top-k without replacement, reversed
    for (int l = tid; l < L; l += 256) {
        const float log_mask = (flag[l] ? log_topk : log_nontopk) - 1e8f * special[l];
        key[l] = log_mask + gumbel[(int64_t)b * L + l];
    }
    __syncthreads();
    for (int l = tid; l < L; l += 256) {
        const int r = desc_rank(key, L, l);
        if (r < nm) sel[nm - 1 - r] = l;   // idx[:, ::-1]
    }
    __syncthreads();
    if (span_lower) {
        // ---- :447-469 SpanBERT extension
        for (int k = tid; k < nm; k += 256) {
            const int c = sel[k];
            sel_end[k] = c + span_upper[(int64_t)b * nm + k];
            flag[k] = c - span_lower[(int64_t)b * nm + k];   // span start (flag reused; nm <= L)
        }
        __syncthreads();
        for (int l = tid; l < L; l += 256) {
            int first = 0;                                    // argmax of an all-false column is 0
            for (int k = 0; k < nm; ++k)
                if (l >= flag[k] && l <= sel_end[k]) { first = k; break; }
            float wm = (float)first * (1.0f - special[l]);
            wm = wm + __fdiv_rn(0.5f * weight[l], max_weight);
            key[l] = wm;
        }
        __syncthreads();
        for (int l = tid; l < L; l += 256) sel[l] = (desc_rank(key, L, l) < nm) ? 1 : 0;
        __syncthreads();
    } else {
        for (int l = tid; l < L; l += 256) sel_end[l] = 0;
        __syncthreads();
        for (int k = tid; k < nm; k += 256) sel_end[sel[k]] = 1;
        __syncthreads();
        for (int l = tid; l < L; l += 256) sel[l] = sel_end[l];
        __syncthreads();
    }
    // ---- :473-487 : sorted indices + 80/10/10 replacement
    for (int l = tid; l < L; l += 256) {
        const int64_t gi = (int64_t)b * L + l;
        const int do_mask = sel[l];
        const int opt = option[gi] * do_mask;
        const int32_t orig = idr[l];
        masked_ids[gi] = opt == 0 ? orig : (opt == 1 ? mask_token : random_ids[gi]);
        if (do_mask) {
            int pos = 0;
            for (int j = 0; j < l; ++j) pos += sel[j];
            masked_idx[(int64_t)b * nm + pos] = l;
        }
    }
}

__global__ void temporal_labels_kernel(const int32_t* __restrict__ vsrc, const int32_t* __restrict__ sidx,
                                       int32_t* __restrict__ labels, float* __restrict__ weights, int B, int n) {
    const int64_t total = (int64_t)B * n * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / (n * n);
        const int rem = (int)(i - b * n * n);
        const int a = rem / n, c = rem % n;     // xa index, xb index   (modeling.py:600-601)
        const int base = (a == c ? 1 : 0) + (a < c ? 2 : 0) + (a > c ? 3 : 0);
        // is_same_video[b, a, c] = vsrc[b, c] == vsrc[b, a]  (:611, symmetric)
        const bool same = vsrc[b * n + a] == vsrc[b * n + c];
        labels[i] = same ? base : 0;
        const bool easy = (sidx[b * n + a] < 64) && (sidx[b * n + c] < 64);   // :635,649
        weights[i] = (easy ? 0.0f : 1.0f) * 0.99f + 0.01f;                      // :650
    }
}

// dataloader.py:248-254 with explicit draws; argsort = ascending rank (stable)
__global__ void shuffled_idx_kernel(const int32_t* __restrict__ num_shuffle, const float* __restrict__ u_select,
                                    const float* __restrict__ u_perm, int32_t* __restrict__ out, int B, int n,
                                    int offset) {
    const int64_t total = (int64_t)B * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / n;
        const int j = (int)(i - b * n);
        // argsort(u)[j] = index of the element with ascending rank j
        int sel_j = 0, perm_j = 0;
        for (int c = 0; c < n; ++c) {
            int rs = 0, rp = 0;
            for (int d = 0; d < n; ++d) {
                rs += (u_select[b * n + d] < u_select[b * n + c] || (u_select[b * n + d] == u_select[b * n + c] && d < c));
                rp += (u_perm[b * n + d] < u_perm[b * n + c] || (u_perm[b * n + d] == u_perm[b * n + c] && d < c));
            }
            if (rs == j) sel_j = c;
            if (rp == j) perm_j = c;
        }
        out[i] = (sel_j < num_shuffle[b]) ? offset + perm_j : j;
    }
}

}  // namespace

extern "C" int merlot_mask_inputs(const int32_t* ids, const float* attention_summs, const float* gumbel,
                                  const int32_t* span_lower, const int32_t* span_upper, const int32_t* random_ids,
                                  const int32_t* option, int32_t* masked_ids, int32_t* masked_idx, int B, int L,
                                  int num_topk, int num_to_mask, float w_nontopk, float w_topk, float log_nontopk,
                                  float log_topk, float max_weight, int mask_token, merlot_stream_t stream) {
    MERLOT_CHECK(ids && gumbel && random_ids && option && masked_ids && masked_idx, MERLOT_ESHAPE,
                 "merlot_mask_inputs: null operand");
    MERLOT_CHECK(B > 0 && L > 0 && L <= MAXL, MERLOT_ESHAPE, "merlot_mask_inputs: L=%d out of range (1..%d)", L, MAXL);
    MERLOT_CHECK(num_to_mask > 0 && num_to_mask <= L && num_topk >= 0 && num_topk <= L, MERLOT_ESHAPE,
                 "merlot_mask_inputs: bad num_to_mask / num_topk");
    MERLOT_CHECK((span_lower == nullptr) == (span_upper == nullptr), MERLOT_ESHAPE,
                 "merlot_mask_inputs: span_lower/span_upper must both be given or both NULL");
    hipLaunchKernelGGL(mask_inputs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, attention_summs, gumbel,
                       span_lower, span_upper, random_ids, option, masked_ids, masked_idx, L, num_topk, num_to_mask,
                       w_nontopk, w_topk, log_nontopk, log_topk, max_weight, mask_token);
    return merlot_launch_status("merlot_mask_inputs");
}

extern "C" int merlot_temporal_labels(const int32_t* video_src_ids, const int32_t* shuffled_idx, int32_t* labels,
                                      float* weights, int B, int n, merlot_stream_t stream) {
    MERLOT_CHECK(video_src_ids && shuffled_idx && labels && weights && B > 0 && n > 0, MERLOT_ESHAPE,
                 "merlot_temporal_labels: bad args");
    const int64_t total = (int64_t)B * n * n;
    int g = (int)((total + 255) / 256);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(temporal_labels_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, video_src_ids, shuffled_idx,
                       labels, weights, B, n);
    return merlot_launch_status("merlot_temporal_labels");
}

extern "C" int merlot_shuffled_idx(const int32_t* num_shuffle, const float* u_select, const float* u_perm, int32_t* out,
                                   int B, int n, int shuffle_offset, merlot_stream_t stream) {
    MERLOT_CHECK(num_shuffle && u_select && u_perm && out && B > 0 && n > 0 && n <= 64, MERLOT_ESHAPE,
                 "merlot_shuffled_idx: bad args");
    const int64_t total = (int64_t)B * n;
    int g = (int)((total + 255) / 256);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(shuffled_idx_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, num_shuffle, u_select, u_perm, out,
                       B, n, shuffle_offset);
    return merlot_launch_status("merlot_shuffled_idx");
}
