// Hardware-layout probes: dump what one MFMA 32x32x16 bf16 and one ds_read_b64_tr_b16 actually do, so
// tests can pin the lane maps the production kernels assume (MFMA operand/accumulator layout; LDS
// transpose-read gather pattern).
#include "common.h"
#include "../../include/merlot_probe.h"

namespace {
__global__ void probe_mfma32_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, float* __restrict__ d) {
    const int lane = threadIdx.x;
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(a + lane * 8);
    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(b + lane * 8);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[lane * 16 + r] = acc[r];
}

// tile: 256 bf16 copied lane-linearly into LDS (lane i owns bytes [8i, 8i+8)); every lane then issues one
// transpose read at ITS OWN 8-byte slot; out[lane][0..3] = what it received.
__global__ void probe_tr16_kernel(const bf16* __restrict__ tile, bf16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) bf16 lds[256];
    const int lane = threadIdx.x;
    *reinterpret_cast<bf16x4*>(lds + lane * 4) = *reinterpret_cast<const bf16x4*>(tile + lane * 4);
    __syncthreads();
    const bf16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + lane * 4));
    *reinterpret_cast<bf16x4*>(out + lane * 4) = r;
}

// the 8-bit transposing read (gemm_q8.inc): 512 bytes copied lane-linearly into LDS (lane i owns bytes [8i, 8i+8)); every lane issues one
// ds_read_b64_tr_b8 at ITS OWN 8-byte slot; out[lane][0..7] = the bytes it received.
__global__ void probe_tr8_kernel(const uint8_t* __restrict__ tile, uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds8[512];
    typedef __attribute__((ext_vector_type(2))) int i32x2_t;
    const int lane = threadIdx.x;
    *reinterpret_cast<i32x2_t*>(lds8 + lane * 8) = *reinterpret_cast<const i32x2_t*>(tile + lane * 8);
    __syncthreads();
    const i32x2_t r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_t*)(lds8 + lane * 8));
    *reinterpret_cast<i32x2_t*>(out + lane * 8) = r;
}

// what v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 do beyond the formats' ranges (saturate? NaN?): out8[i] = e4m3(in[i]), out8[n + i] = e5m2(in[i]), no clamp in front
__global__ void probe_cvt8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out8, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, w, false);
    out8[i] = (uint8_t)(w & 0xff);
    w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(in[i], 0.f, w, false);
    out8[n + i] = (uint8_t)(w & 0xff);
}

// Occupies `blocks` CUs' worth of LDS (each workgroup declares lds_bytes of dynamic LDS) for ~`cycles` shader clocks:
// a stand-in for a communication kernel running beside the GEMMs (scripts/exp_persist_dyn.py).
__global__ void probe_cu_hog_kernel(long long cycles, unsigned int* sink) {
    extern __shared__ char hog_lds[];
    hog_lds[threadIdx.x] = (char)threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    unsigned int acc = 0;
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) {
        acc += hog_lds[(threadIdx.x + acc) & 63];
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}
// Matrix-pipe ceiling under the GEMM's own instruction mix (experiments): every wave owns 8 accumulators of 32x32 and
// issues 16 MFMAs per iteration -- a K-step of the 256x256 kernel -- optionally with that K-step's 12 ds_read_b128
// fragment reads (mode & 1), its s_barrier (mode & 2) and register operands refreshed from those reads (mode & 4).
// out[block] = {s_memtime delta, s_memrealtime delta (100 MHz), 0, 0}.
__global__ __launch_bounds__(512) void probe_mfma_rate_kernel(int iters, int mode, long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char plds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float*>(plds)[i] = 0.f;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fa[2], fb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[i][e] = (bf16)0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[i][e] = (bf16)0.f;
    const int hi = lane >> 5;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (mode & 2) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (mode & 1) {
                bf16x8 ra[2], rb[4];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int row = ((wave >> 1) * 2 + f) * 32 + (lane & 31);
                    ra[f] = *reinterpret_cast<const bf16x8*>(plds + row * 64 + ((((2 * kk + hi) ^ (row >> 2)) & 3) << 4));
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int row = ((wave & 1) * 4 + f) * 32 + (lane & 31);
                    rb[f] = *reinterpret_cast<const bf16x8*>(plds + 16384 + row * 64 + ((((2 * kk + hi) ^ (row >> 2)) & 3) << 4));
                }
                if (mode & 4) {
#pragma unroll
                    for (int f = 0; f < 2; ++f) fa[f] = ra[f];
#pragma unroll
                    for (int f = 0; f < 4; ++f) fb[f] = rb[f];
                } else {
                    asm volatile("" ::"v"(ra[0]), "v"(ra[1]), "v"(rb[0]), "v"(rb[1]), "v"(rb[2]), "v"(rb[3]));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i * 4 + j], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = r1 - r0;
    }
}

// Store-rate probe (experiments): `blocks` 8-wave workgroups each write `tiles` output tiles of 256 x 256 bf16 (row stride ld
// elements) the way the GEMM epilogue does -- wave w owns rows (w >> 2) * 128 .. + 127, columns (w & 3) * 64 .. + 63, one
// 16-B store per lane and instruction -- with `rows_per_instr` rows covered by one instruction (8: 128 B per row, the
// epilogue's pattern; 4 / 2 / 1 emulate 256 / 512 / 1024 contiguous bytes per row by letting the wave own a wider strip).
// mode bit 0: s_waitcnt vmcnt(0) + s_barrier after every tile (as the epilogue does); bit 1: 160 KiB of LDS (one workgroup per CU).
__global__ __launch_bounds__(512) void probe_store_kernel(bf16* __restrict__ out, long long ld, int tiles_m, int tiles, int rows_per_instr,
                                                          int mode, long long* __restrict__ clk) {
    extern __shared__ char pst_lds[];
    if (mode & 2) pst_lds[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lanes_per_row = 64 / rows_per_instr;          // 8, 16, 32, 64 lanes x 16 B
    const int cols_per_wave = lanes_per_row * 8;            // 64 .. 512 columns
    const int strips = 256 / cols_per_wave > 0 ? 256 / cols_per_wave : 1;   // column strips per tile
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)(float)(lane + e);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile % tiles_m, tn = tile / tiles_m;
        bf16* base = out + (long long)tm * 256 * ld + (long long)tn * 256;
        // every wave writes 8192 elements = 16 instructions
        const int rows_per_wave = 8192 / (cols_per_wave < 256 ? cols_per_wave : 256);
        const int strip = wave % strips, rblk = wave / strips;
        const int wcols = cols_per_wave < 256 ? cols_per_wave : 256;
        for (int i = 0; i < 16; ++i) {
            const int idx = i * 64 + lane;                  // 16-B unit inside the wave's block, row-major
            const int units_per_row = wcols / 8;
            const int r = rblk * rows_per_wave + idx / units_per_row;
            const int c = strip * wcols + (idx % units_per_row) * 8;
            *reinterpret_cast<bf16x8*>(base + (long long)r * ld + c) = v;
        }
        if (mode & 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
}  // namespace

extern "C" int merlot_probe_mfma_rate(int blocks, int iters, int mode, void* out, void* sink, merlot_stream_t stream) {
    MERLOT_CHECK(out && sink && blocks > 0 && iters > 0, MERLOT_ESHAPE, "merlot_probe_mfma_rate: bad arguments");
    hipLaunchKernelGGL(probe_mfma_rate_kernel, dim3(blocks), dim3(512), 32768, (hipStream_t)stream, iters, mode,
                       (long long*)out, (float*)sink);
    return merlot_launch_status("merlot_probe_mfma_rate");
}

extern "C" int merlot_probe_mfma32(const void* a, const void* b, float* d, merlot_stream_t stream) {
    MERLOT_CHECK(a && b && d, MERLOT_ESHAPE, "merlot_probe_mfma32: null operand");
    hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, d);
    return merlot_launch_status("merlot_probe_mfma32");
}
extern "C" int merlot_probe_tr16(const void* tile, void* out, merlot_stream_t stream) {
    MERLOT_CHECK(tile && out, MERLOT_ESHAPE, "merlot_probe_tr16: null operand");
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)tile, (bf16*)out);
    return merlot_launch_status("merlot_probe_tr16");
}
extern "C" int merlot_probe_cvt8(const float* in, void* out8, int n, merlot_stream_t stream) {
    MERLOT_CHECK(in && out8 && n > 0, MERLOT_ESHAPE, "merlot_probe_cvt8: bad argument");
    hipLaunchKernelGGL(probe_cvt8_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, (uint8_t*)out8, n);
    return merlot_launch_status("merlot_probe_cvt8");
}
extern "C" int merlot_probe_tr8(const void* tile, void* out, merlot_stream_t stream) {
    MERLOT_CHECK(tile && out, MERLOT_ESHAPE, "merlot_probe_tr8: null argument");
    hipLaunchKernelGGL(probe_tr8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)tile, (uint8_t*)out);
    return merlot_launch_status("merlot_probe_tr8");
}
extern "C" int merlot_probe_cu_hog(int blocks, int lds_bytes, int64_t cycles, void* sink, merlot_stream_t stream) {
    MERLOT_CHECK(blocks > 0 && lds_bytes >= 64 && lds_bytes <= 160 * 1024 && sink, MERLOT_ESHAPE, "merlot_probe_cu_hog: bad arguments");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(probe_cu_hog_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(probe_cu_hog_kernel, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, (long long)cycles,
                       (unsigned int*)sink);
    return merlot_launch_status("merlot_probe_cu_hog");
}

extern "C" int merlot_probe_store(void* out, int64_t ld, int tiles_m, int blocks, int tiles, int rows_per_instr, int mode, void* clk,
                                  merlot_stream_t stream) {
    MERLOT_CHECK(out && clk && blocks > 0 && tiles > 0 && (rows_per_instr == 8 || rows_per_instr == 4 || rows_per_instr == 2 || rows_per_instr == 1),
                 MERLOT_ESHAPE, "merlot_probe_store: bad arguments");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(probe_store_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MERLOT_CHECK(e == hipSuccess, MERLOT_ELAUNCH, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(probe_store_kernel, dim3(blocks), dim3(512), (mode & 2) ? 160 * 1024 : 1024, (hipStream_t)stream, (bf16*)out,
                       (long long)ld, tiles_m, tiles, rows_per_instr, mode, (long long*)clk);
    return merlot_launch_status("merlot_probe_store");
}
