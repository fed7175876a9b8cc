// Hardware-layout probes: dump what one MFMA 32x32x16 bf16 and one ds_read_b64_tr_b16 actually do, so
// tests can pin the lane maps the production kernels assume (MFMA operand/accumulator layout; LDS
// transpose-read gather pattern).
#include "common.h"

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
}  // namespace

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
