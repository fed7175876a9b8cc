// Ring-pipelined bf16 MFMA GEMM core for gfx950 (shared by the NT and TN kernels).
//
//   * workgroup = WM x WN waves, each wave FM x FN accumulators of 32x32 (MFMA 32x32x16 bf16)
//       => block tile BM = WM*FM*32 rows x BN = WN*FN*32 columns, K-step BK (32 or 64).
//   * STAGES-deep LDS ring filled by `global_load_lds_dwordx4` (1 KiB per wave-instruction).  Loads for K-step
//     t+STAGES-1 are issued right after the barrier that opens step t and are only waited for STAGES-1 steps later
//     with a COUNTED `s_waitcnt vmcnt(N)` (never 0 in the steady state), so HBM/L2 latency (~1-2k cycles under
//     load) is covered by STAGES-1 steps of MFMA work instead of stalling every step.  Raw `s_barrier` is used:
//     `__syncthreads()` would drain vmcnt to 0 (LDS-DMA counts as a pending LDS write).
//   * one barrier per K-step: the slot refilled at step t was last read at step t-1, and every wave has finished
//     step t-1 when it arrives at the barrier of step t.
#pragma once
#include "common.h"

namespace ring {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K-contiguous tile image: [rows][BK bf16], 16-B chunks XOR-swizzled so a ds_read_b128 lane group is conflict free
template <int BK>
__device__ __forceinline__ int kc_off(int row, int chunk) {
    if (BK == 64) return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4);
    return row * 64 + (((chunk ^ (row >> 2)) & 3) << 4);
}

template <int WM_, int WN_, int FM_, int FN_, int BK_, int STAGES_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, FM = FM_, FN = FN_, BK = BK_, STAGES = STAGES_;
    static constexpr int NW = WM * WN;
    static constexpr int NT = NW * 64;
    static constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    static constexpr int ROW_BYTES = BK * 2;
    static constexpr int RP = 1024 / ROW_BYTES;                 // rows per 1 KiB LDS-DMA piece (K-contiguous image)
    static constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024 / NW;        // per wave
    static constexpr int B_PIECES = B_BYTES / 1024 / NW;
    static constexpr int LOADS = A_PIECES + B_PIECES;           // VMEM ops per wave per stage
    static constexpr int RING_BYTES = STAGES * STAGE_BYTES;
    static constexpr int EPI_BYTES = 32 * (FN * 32 * 4 + 16);     // per-wave fp32 slab of the staged epilogue
    static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES * NW ? RING_BYTES : EPI_BYTES * NW;
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "tile must split into whole pieces per wave");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS ring too large");
};

}  // namespace ring
