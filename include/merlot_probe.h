/* libmerlot_probe.so -- hardware probes and experiment helpers.  NOT part of the product library
 * (libmerlot_hip.so exports none of these): tests use the two layout probes to pin the MFMA / LDS-transpose lane
 * maps the production kernels assume; scripts/ use the rest for the measurements in profiles/.
 * merlot_probe_persist_trace exists only in the experiments build of the main library
 * (`merlot_amd/csrc/build.sh exp` -> libmerlot_hip_exp.so, compiled with -DMERLOT_EXPERIMENTS). */
#ifndef MERLOT_PROBE_H
#define MERLOT_PROBE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* merlot_stream_t;
const char* merlot_last_error(void);

/* out_* are small device buffers, see csrc/probe.hip */
int merlot_probe_mfma32(const void* a, const void* b, float* d, merlot_stream_t stream);
int merlot_probe_tr16(const void* tile, void* out, merlot_stream_t stream);
/* ds_read_b64_tr_b8: tile = 512 bytes laid lane-linearly into LDS (lane i owns bytes [8i, 8i + 8)), out[lane][0..7] = what lane received from a read at its own slot */
int merlot_probe_tr8(const void* tile, void* out, merlot_stream_t stream);
/* out8[i] = v_cvt_pk_fp8_f32(in[i]), out8[n + i] = v_cvt_pk_bf8_f32(in[i]) with NO clamp in front: what the conversions do beyond the formats' ranges */
int merlot_probe_cvt8(const float* in, void* out8, int n, merlot_stream_t stream);
/* experiment helper: `blocks` one-wave workgroups that each hold lds_bytes of LDS and spin for ~cycles shader clocks
 * (a stand-in for a communication kernel sharing the GPU with the GEMMs); sink = any 4-byte device buffer. */
int merlot_probe_cu_hog(int blocks, int lds_bytes, int64_t cycles, void* sink, merlot_stream_t stream);
/* experiment helper: matrix-pipe rate of `blocks` 8-wave workgroups issuing the 256x256 GEMM's K-step instruction mix
 * (16 MFMA 32x32x16 per wave and iteration; mode bit 1: + its 12 ds_read_b128, bit 2: + s_barrier, bit 4: MFMA operands
 * come from those reads).  out: int64 [blocks][4] = {shader-clock delta, 100 MHz wall-clock delta, 0, 0}. */
int merlot_probe_mfma_rate(int blocks, int iters, int mode, void* out, void* sink, merlot_stream_t stream);

/* experiment helper: `blocks` 8-wave workgroups each write `tiles` 256 x 256 bf16 tiles of a [tiles_m * 256, ...] matrix (row stride ld)
 * with the GEMM epilogue's store pattern; rows_per_instr 8 / 4 / 2 / 1 = 128 / 256 / 512 / 1024 contiguous bytes per row and
 * instruction; mode bit 0: vmcnt(0) + barrier per tile, bit 1: 160 KiB LDS per workgroup.  clk: int64 [blocks] shader clocks. */
int merlot_probe_store(void* out, int64_t ld, int tiles_m, int blocks, int tiles, int rows_per_instr, int mode, void* clk,
                       merlot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MERLOT_PROBE_H */
