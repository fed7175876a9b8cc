// Frame preprocessing of the input pipeline on the GPU: decoded JPEG bytes -> model-ready bf16
// frames, one launch pair per batch.  Restates, per frame (model/dataloader.py:72-97):
//   convert_image_dtype(uint8 -> f32)            x * (1/255)
//   resize_and_pad (utils/model_utils.py:860-940) tf.image.resize_images(method in {bilinear, nearest, bicubic, area},
//                                                align_corners=True) to [scaled_h, scaled_w], crop at (offset_y,
//                                                offset_x), zero-pad to [out_h, out_w]
//   where(is_finite)                              (:86-87)
//   lightweight_image_augment (:758-842)          brightness: x*f[c] | contrast: (x-mean[c])*f[c]+mean[c], clip [0,1]
//   cast bf16                                     (:95-97)
// The four resize kernels follow tensorflow==1.15.5 core/kernels/resize_{bilinear,nearest_neighbor,bicubic,area}_op.cc
// with the legacy (non half-pixel) scaler, restated from the published algorithm (TensorFlow is absent: parity unpinned).
// HBM-bound byte work: a thread owns one output pixel (3 channels), reads <= 16 source pixels through L2, writes 6 B.
#include "common.h"

// every product and sum is rounded separately, as in the TF CPU kernels this restates (no fused multiply-add)
#pragma clang fp contract(off)

namespace {

struct Job {
    int64_t src_offset;            // bytes into `src` of this frame's HWC uint8 image
    int32_t src_h, src_w;
    int32_t scaled_h, scaled_w;
    int32_t method;                // tf.image.ResizeMethod: 0 bilinear, 1 nearest, 2 bicubic, 3 area
    int32_t offset_y, offset_x;
    int32_t aug_kind;              // 0 none, 1 brightness, 2 contrast
    float factor[3];
    float reserved;
};
static_assert(sizeof(Job) == sizeof(merlot_image_job_t), "job layout is part of the ABI");

__device__ __forceinline__ float resize_scale(int in, int out) {          // CalculateResizeScale, align_corners=True
    return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;            // (image_resizer_state.h)
}
__device__ __forceinline__ int bound(int v, int limit) { return min(limit - 1, max(0, v)); }

struct Px {
    float c[3];
};
__device__ __forceinline__ Px load_px(const uint8_t* img, int w, int y, int x) {
    const uint8_t* p = img + ((int64_t)y * w + x) * 3;
    const float k = 1.0f / 255.0f;                                        // convert_image_dtype: cast * (1 / max)
    Px r;
    r.c[0] = (float)p[0] * k;
    r.c[1] = (float)p[1] * k;
    r.c[2] = (float)p[2] * k;
    return r;
}

__device__ __forceinline__ void cubic_weights(float loc, int limit, int (&idx)[4], float (&w)[4]) {
    // GetWeightsAndIndices<LegacyScaler, false>: A = -0.75, weights tabulated at 1/1024 steps of the fraction
    const float A = -0.75f;
    const int in_loc = (int)floorf(loc);
    const float delta = loc - (float)in_loc;
    const int off = (int)lrintf(delta * 1024.f);
    const float x0 = (float)off * (1.0f / 1024.f), x1 = (float)(1024 - off) * (1.0f / 1024.f);
    auto near = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
    auto far = [&](float x) {
        x += 1.0f;
        return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    };
    idx[0] = bound(in_loc - 1, limit);
    idx[1] = bound(in_loc, limit);
    idx[2] = bound(in_loc + 1, limit);
    idx[3] = bound(in_loc + 2, limit);
    w[0] = far(x0);
    w[1] = near(x0);
    w[2] = near(x1);
    w[3] = far(x1);
}

// value of the resized image [scaled_h, scaled_w] at (sy, sx)
__device__ Px resized(const Job& j, const uint8_t* img, int sy, int sx) {
    const float hs = resize_scale(j.src_h, j.scaled_h), ws = resize_scale(j.src_w, j.scaled_w);
    Px o;
    if (j.method == 1) {                                                  // nearest: roundf (align_corners)
        const int y = min((int)roundf((float)sy * hs), j.src_h - 1), x = min((int)roundf((float)sx * ws), j.src_w - 1);
        return load_px(img, j.src_w, y, x);
    }
    if (j.method == 0) {                                                  // bilinear
        const float fy = (float)sy * hs, fx = (float)sx * ws;
        const float fy0 = floorf(fy), fx0 = floorf(fx);
        const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), j.src_h - 1);
        const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), j.src_w - 1);
        const float ly = fy - fy0, lx = fx - fx0;
        const Px tl = load_px(img, j.src_w, y0, x0), tr = load_px(img, j.src_w, y0, x1);
        const Px bl = load_px(img, j.src_w, y1, x0), br = load_px(img, j.src_w, y1, x1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = tl.c[c] + (tr.c[c] - tl.c[c]) * lx;
            const float bot = bl.c[c] + (br.c[c] - bl.c[c]) * lx;
            o.c[c] = top + (bot - top) * ly;
        }
        return o;
    }
    if (j.method == 2) {                                                  // bicubic
        int yi[4], xi[4];
        float yw[4], xw[4];
        cubic_weights((float)sy * hs, j.src_h, yi, yw);
        cubic_weights((float)sx * ws, j.src_w, xi, xw);
        float col[4][3];                                                  // TF interpolates along y first (cached per x index)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Px a = load_px(img, j.src_w, yi[0], xi[k]), b = load_px(img, j.src_w, yi[1], xi[k]);
            const Px c2 = load_px(img, j.src_w, yi[2], xi[k]), d = load_px(img, j.src_w, yi[3], xi[k]);
#pragma unroll
            for (int c = 0; c < 3; ++c) col[k][c] = a.c[c] * yw[0] + b.c[c] * yw[1] + c2.c[c] * yw[2] + d.c[c] * yw[3];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) o.c[c] = col[0][c] * xw[0] + col[1][c] * xw[1] + col[2][c] * xw[2] + col[3][c] * xw[3];
        return o;
    }
    // area: box [y*s, (y+1)*s) x [x*s, (x+1)*s) with fractional edge weights, indices clamped, result / (s_y * s_x)
    const float in_y = (float)sy * hs, in_y1 = (float)(sy + 1) * hs;
    const float in_x = (float)sx * ws, in_x1 = (float)(sx + 1) * ws;
    const int ys = (int)floorf(in_y), ye = (int)ceilf(in_y1);
    const int xs = (int)floorf(in_x), xe = (int)ceilf(in_x1);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int i = ys; i < ye; ++i) {
        const float wy = (float)i < in_y ? ((float)(i + 1) > in_y1 ? hs : (float)(i + 1) - in_y)
                                         : ((float)(i + 1) > in_y1 ? in_y1 - (float)i : 1.0f);
        float rs[3] = {0.f, 0.f, 0.f};
        for (int k = xs; k < xe; ++k) {
            const float wx = (float)k < in_x ? ((float)(k + 1) > in_x1 ? ws : (float)(k + 1) - in_x)
                                             : ((float)(k + 1) > in_x1 ? in_x1 - (float)k : 1.0f);
            const Px p = load_px(img, j.src_w, bound(i, j.src_h), bound(k, j.src_w));
#pragma unroll
            for (int c = 0; c < 3; ++c) rs[c] += p.c[c] * wx;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += rs[c] * wy;
    }
    const float inv = 1.0f / (hs * ws);
#pragma unroll
    for (int c = 0; c < 3; ++c) o.c[c] = acc[c] * inv;
    return o;
}

// the frame after resize + crop + pad + is_finite, at output pixel (y, x)
__device__ __forceinline__ Px frame_px(const Job& j, const uint8_t* src, int y, int x) {
    const int sy = y + j.offset_y, sx = x + j.offset_x;
    Px o = {{0.f, 0.f, 0.f}};
    if (sy < j.scaled_h && sx < j.scaled_w) {
        o = resized(j, src + j.src_offset, sy, sx);
#pragma unroll
        for (int c = 0; c < 3; ++c) o.c[c] = isfinite(o.c[c]) ? o.c[c] : 0.f;
    }
    return o;
}

constexpr int IMG_THREADS = 256;

// pass 1 (contrast frames only): per-block channel sums -> partial[img][block][3]
__global__ __launch_bounds__(IMG_THREADS) void image_sums_kernel(const uint8_t* __restrict__ src, const Job* __restrict__ jobs,
                                                                 int out_h, int out_w, float* __restrict__ partial) {
    const Job j = jobs[blockIdx.y];
    if (j.aug_kind != 2) return;
    const int pix = blockIdx.x * IMG_THREADS + threadIdx.x;
    Px v = {{0.f, 0.f, 0.f}};
    if (pix < out_h * out_w) v = frame_px(j, src, pix / out_w, pix % out_w);
    __shared__ float red[IMG_THREADS / 64][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = v.c[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int w = 0; w < IMG_THREADS / 64; ++w) s += red[w][threadIdx.x];
        partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = s;
    }
}

// pass 1b (contrast frames only): per-frame channel means from the block partials -- one workgroup per frame, a fixed
// reduction tree (strided per-thread sums, wave butterflies, waves in order): the same bits on every run
__global__ __launch_bounds__(IMG_THREADS) void image_means_kernel(const Job* __restrict__ jobs, const float* __restrict__ partial,
                                                                  int nblk, float count, float* __restrict__ means) {
    if (jobs[blockIdx.x].aug_kind != 2) return;
    __shared__ float red[IMG_THREADS / 64][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
        for (int b = threadIdx.x; b < nblk; b += IMG_THREADS) s += partial[((int64_t)blockIdx.x * nblk + b) * 3 + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int w = 0; w < IMG_THREADS / 64; ++w) s += red[w][threadIdx.x];
        means[blockIdx.x * 3 + threadIdx.x] = s / count;
    }
}

// pass 2: resize + crop + pad (+ augment) -> bf16 NHWC
__global__ __launch_bounds__(IMG_THREADS) void image_frames_kernel(const uint8_t* __restrict__ src, const Job* __restrict__ jobs,
                                                                   int out_h, int out_w, const float* __restrict__ means,
                                                                   bf16* __restrict__ dst) {
    const Job j = jobs[blockIdx.y];
    float mean[3] = {0.f, 0.f, 0.f};
    if (j.aug_kind == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) mean[c] = means[blockIdx.y * 3 + c];
    }
    const int pix = blockIdx.x * IMG_THREADS + threadIdx.x;
    if (pix >= out_h * out_w) return;
    Px v = frame_px(j, src, pix / out_w, pix % out_w);
    if (j.aug_kind == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v.c[c] = fminf(fmaxf(v.c[c] * j.factor[c], 0.f), 1.f);
    } else if (j.aug_kind == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v.c[c] = fminf(fmaxf((v.c[c] - mean[c]) * j.factor[c] + mean[c], 0.f), 1.f);
    }
    bf16* o = dst + ((int64_t)blockIdx.y * out_h * out_w + pix) * 3;
    o[0] = (bf16)v.c[0];
    o[1] = (bf16)v.c[1];
    o[2] = (bf16)v.c[2];
}

}  // namespace

extern "C" int64_t merlot_image_frames_workspace_bytes(int n_img, int out_h, int out_w) {
    if (n_img <= 0 || out_h <= 0 || out_w <= 0) return 0;
    const int64_t blocks = ((int64_t)out_h * out_w + IMG_THREADS - 1) / IMG_THREADS;
    return (int64_t)n_img * (blocks + 1) * 3 * sizeof(float);        // block partials + the per-frame means
}

extern "C" int merlot_image_frames(const uint8_t* src, int64_t src_bytes, const merlot_image_job_t* jobs_host,
                                   const merlot_image_job_t* jobs_dev, int n_img, void* dst, int out_h, int out_w,
                                   void* workspace, int64_t workspace_bytes, merlot_stream_t stream) {
    MERLOT_CHECK(src && jobs_host && jobs_dev && dst && n_img > 0, MERLOT_ESHAPE, "merlot_image_frames: null operand");
    MERLOT_CHECK(out_h > 0 && out_w > 0, MERLOT_ESHAPE, "merlot_image_frames: bad output size %dx%d", out_h, out_w);
    MERLOT_CHECK(workspace && workspace_bytes >= merlot_image_frames_workspace_bytes(n_img, out_h, out_w), MERLOT_ESHAPE,
                 "merlot_image_frames: workspace too small");
    bool any_contrast = false;
    for (int i = 0; i < n_img; ++i) {                    // the host copy of the table is validated before anything runs
        const merlot_image_job_t& j = jobs_host[i];
        MERLOT_CHECK(j.src_h > 0 && j.src_w > 0 && j.scaled_h > 0 && j.scaled_w > 0, MERLOT_ESHAPE,
                     "merlot_image_frames: frame %d has an empty source or target (%dx%d -> %dx%d)", i, j.src_h, j.src_w,
                     j.scaled_h, j.scaled_w);
        MERLOT_CHECK(j.src_offset >= 0 && j.src_offset + (int64_t)j.src_h * j.src_w * 3 <= src_bytes, MERLOT_ESHAPE,
                     "merlot_image_frames: frame %d lies outside the source buffer", i);
        MERLOT_CHECK(j.method >= 0 && j.method <= 3, MERLOT_ESHAPE, "merlot_image_frames: frame %d: resize method %d", i,
                     j.method);
        MERLOT_CHECK(j.offset_y >= 0 && j.offset_x >= 0 && j.aug_kind >= 0 && j.aug_kind <= 2, MERLOT_ESHAPE,
                     "merlot_image_frames: frame %d: bad crop offset / augment kind", i);
        any_contrast |= j.aug_kind == 2;
    }
    const dim3 grid((out_h * out_w + IMG_THREADS - 1) / IMG_THREADS, n_img);
    float* partial = (float*)workspace;
    float* means = partial + (int64_t)n_img * grid.x * 3;
    if (any_contrast) {
        hipLaunchKernelGGL(image_sums_kernel, grid, dim3(IMG_THREADS), 0, (hipStream_t)stream, src, (const Job*)jobs_dev,
                           out_h, out_w, partial);
        hipLaunchKernelGGL(image_means_kernel, dim3(n_img), dim3(IMG_THREADS), 0, (hipStream_t)stream, (const Job*)jobs_dev,
                           partial, (int)grid.x, (float)(out_h * out_w), means);
    }
    hipLaunchKernelGGL(image_frames_kernel, grid, dim3(IMG_THREADS), 0, (hipStream_t)stream, src, (const Job*)jobs_dev, out_h,
                       out_w, means, (bf16*)dst);
    return merlot_launch_status("merlot_image_frames");
}
