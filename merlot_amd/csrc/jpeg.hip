// GPU half of the JPEG decoder (host half: jpeg_host.cpp): dequantisation + inverse DCT + chroma upsampling + colour
// conversion of `tf.image.decode_jpeg(x, channels=3)` (model/dataloader.py:72-77), restated from the algorithms libjpeg
// documents and uses by default -- the "islow" integer IDCT of Loeffler, Ligtenberg and Moschytz (jidctint), the triangle
// ("fancy") h2v2 upsampler, the 16-bit fixed-point YCbCr -> RGB tables -- so that the frames equal the host library's bit for
// bit (tests/test_jpeg_gpu.py compares with PIL/libjpeg-turbo).  HBM-bound integer work: one thread per 8x8 block column /
// row pass through LDS, one thread per output pixel pair for the colour stage.
#include "common.h"

namespace {

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
              F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one 1-D pass of the LL&M inverse DCT on 8 values (in place), results descaled by `shift`
__device__ __forceinline__ void idct8(int (&d)[8], int shift) {
    int z2 = d[2], z3 = d[6];
    int z1 = (z2 + z3) * F_0_541;
    int tmp2 = z1 + z3 * (-F_1_847);
    int tmp3 = z1 + z2 * F_0_765;
    z2 = d[0]; z3 = d[4];
    int tmp0 = (z2 + z3) << CONST_BITS;
    int tmp1 = (z2 - z3) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = d[7]; tmp1 = d[5]; tmp2 = d[3]; tmp3 = d[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    d[0] = descale(tmp10 + tmp3, shift); d[7] = descale(tmp10 - tmp3, shift);
    d[1] = descale(tmp11 + tmp2, shift); d[6] = descale(tmp11 - tmp2, shift);
    d[2] = descale(tmp12 + tmp1, shift); d[5] = descale(tmp12 - tmp1, shift);
    d[3] = descale(tmp13 + tmp0, shift); d[4] = descale(tmp13 - tmp0, shift);
}

// grid.x: blocks of 32 DCT blocks (256 threads = 32 blocks x 8 columns/rows), grid.y: component, grid.z: image
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const merlot_jpeg_info_t* __restrict__ infos,
                                                        uint8_t* __restrict__ planes) {
    __shared__ int ws[32][8][9];
    const merlot_jpeg_info_t& J = infos[blockIdx.z];
    const int c = blockIdx.y;
    const int bw = J.blocks_w[c], nb = bw * J.blocks_h[c];
    const int lb = threadIdx.x >> 3, t = threadIdx.x & 7;
    const int blk = blockIdx.x * 32 + lb;
    if (blockIdx.x * 32 >= nb) return;
    const bool live = blk < nb;
    int d[8];
    if (live) {
        const int16_t* q = coef + J.coef_base + J.coef_offset[c] + (int64_t)blk * 64;
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = (int)q[r * 8 + t] * (int)J.quant[c][r * 8 + t];       // column t, dequantised
        idct8(d, CONST_BITS - PASS1_BITS);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r][t] = d[r];
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = ws[lb][t][k];                                         // row t
        idct8(d, CONST_BITS + PASS1_BITS + 3);
        // plane of component c: [blocks_h * 8][blocks_w * 8] bytes
        int64_t poff = J.plane_offset;
        for (int cc = 0; cc < c; ++cc) poff += (int64_t)J.blocks_w[cc] * J.blocks_h[cc] * 64;
        const int by = blk / bw, bx = blk - by * bw;
        uint8_t* o = planes + poff + ((int64_t)(by * 8 + t) * bw + bx) * 8;
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w0 |= (uint32_t)clamp255(d[k] + 128) << (8 * k);
            w1 |= (uint32_t)clamp255(d[4 + k] + 128) << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(o) = w0;
        *reinterpret_cast<uint32_t*>(o + 4) = w1;
    }
}

// YCbCr -> RGB, jdcolor's 16-bit fixed-point tables evaluated in place
__device__ __forceinline__ void ycc_rgb(int y, int cb, int cr, uint8_t* o) {
    const int xb = cb - 128, xr = cr - 128;
    const int r = y + ((91881 * xr + 32768) >> 16);                    // FIX(1.40200)
    const int g = y + ((-22554 * xb + 32768 - 46802 * xr) >> 16);      // Cb_g_tab carries ONE_HALF: -FIX(0.34414), -FIX(0.71414)
    const int b = y + ((116130 * xb + 32768) >> 16);                   // FIX(1.77200)
    o[0] = (uint8_t)clamp255(r);
    o[1] = (uint8_t)clamp255(g);
    o[2] = (uint8_t)clamp255(b);
}

// one thread per output pixel; 4:2:0 chroma through the h2v2 "fancy" (triangle) upsampler:
//   vertical: 3 * nearest row + 1 * next-nearest row (the rows above the first / below the last real chroma row repeat it),
//   horizontal on those sums: even x: (3 * this + left + 8) >> 4, odd x: (3 * this + right + 7) >> 4; first / last column
//   (4 * this + 8) >> 4 and (4 * this + 7) >> 4
__global__ __launch_bounds__(256) void jpeg_color_kernel(const merlot_jpeg_info_t* __restrict__ infos, const uint8_t* __restrict__ planes,
                                                         uint8_t* __restrict__ dst) {
    const merlot_jpeg_info_t& J = infos[blockIdx.z];
    const int W = J.width, H = J.height;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int yw = J.blocks_w[0] * 8;
    const uint8_t* py = planes + J.plane_offset;
    const uint8_t* pcb = py + (int64_t)J.blocks_w[0] * J.blocks_h[0] * 64;
    const uint8_t* pcr = pcb + (int64_t)J.blocks_w[1] * J.blocks_h[1] * 64;
    const int cw = J.blocks_w[1] * 8;
    const int Y = py[(int64_t)y * yw + x];
    int cb, cr;
    if (J.subsampling == 1) {
        cb = pcb[(int64_t)y * cw + x];
        cr = pcr[(int64_t)y * cw + x];
    } else {
        const int dw = (W + 1) >> 1, dh = (H + 1) >> 1;                 // downsampled_width / height: the REAL chroma samples
        const int cy = y >> 1, cx = x >> 1;
        int ny = (y & 1) ? cy + 1 : cy - 1;                             // the next-nearest chroma row
        ny = ny < 0 ? 0 : (ny > dh - 1 ? dh - 1 : ny);
        auto colsum = [&](const uint8_t* p, int xx) { return 3 * (int)p[(int64_t)cy * cw + xx] + (int)p[(int64_t)ny * cw + xx]; };
        auto up = [&](const uint8_t* p) {
            const int cur = colsum(p, cx);
            if (x & 1) {
                if (cx == dw - 1) return (cur * 4 + 7) >> 4;
                return (cur * 3 + colsum(p, cx + 1) + 7) >> 4;
            }
            if (cx == 0) return (cur * 4 + 8) >> 4;
            return (cur * 3 + colsum(p, cx - 1) + 8) >> 4;
        };
        cb = up(pcb);
        cr = up(pcr);
    }
    ycc_rgb(Y, cb, cr, dst + J.dst_offset + ((int64_t)y * W + x) * 3);
}

}  // namespace

extern "C" int64_t merlot_jpeg_plane_bytes(const merlot_jpeg_info_t* info) {
    if (!info) return 0;
    int64_t n = 0;
    for (int c = 0; c < 3; ++c) n += (int64_t)info->blocks_w[c] * info->blocks_h[c] * 64;
    return n;
}

extern "C" int merlot_jpeg_idct_rgb(const int16_t* coef, const merlot_jpeg_info_t* infos_host, const merlot_jpeg_info_t* infos_dev,
                                    int n_img, uint8_t* workspace, int64_t workspace_bytes, uint8_t* dst, int64_t dst_bytes,
                                    merlot_stream_t stream) {
    MERLOT_CHECK(coef && infos_host && infos_dev && workspace && dst && n_img > 0, MERLOT_ESHAPE, "merlot_jpeg_idct_rgb: null argument");
    int max_nb = 0, max_w = 0, max_h = 0;
    for (int i = 0; i < n_img; ++i) {
        const merlot_jpeg_info_t& J = infos_host[i];
        MERLOT_CHECK(J.width > 0 && J.height > 0 && (J.subsampling == 1 || J.subsampling == 2), MERLOT_ESHAPE,
                     "merlot_jpeg_idct_rgb: image %d: bad header", i);
        MERLOT_CHECK(J.plane_offset >= 0 && J.plane_offset % 4 == 0 && J.plane_offset + merlot_jpeg_plane_bytes(&J) <= workspace_bytes, MERLOT_ESHAPE,
                     "merlot_jpeg_idct_rgb: image %d: planes outside the workspace", i);
        MERLOT_CHECK(J.dst_offset >= 0 && J.dst_offset + (int64_t)J.width * J.height * 3 <= dst_bytes, MERLOT_ESHAPE,
                     "merlot_jpeg_idct_rgb: image %d: output outside dst", i);
        MERLOT_CHECK(J.coef_base >= 0, MERLOT_ESHAPE, "merlot_jpeg_idct_rgb: image %d: bad coefficient offset", i);
        for (int c = 0; c < 3; ++c) max_nb = J.blocks_w[c] * J.blocks_h[c] > max_nb ? J.blocks_w[c] * J.blocks_h[c] : max_nb;
        max_w = J.width > max_w ? J.width : max_w;
        max_h = J.height > max_h ? J.height : max_h;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_nb + 31) / 32, 3, n_img), dim3(256), 0, s, coef, infos_dev, workspace);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_w + 63) / 64, (max_h + 3) / 4, n_img), dim3(256), 0, s, infos_dev, workspace, dst);
    return merlot_launch_status("merlot_jpeg_idct_rgb");
}
