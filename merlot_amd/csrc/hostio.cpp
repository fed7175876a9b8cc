// Host-side byte work of the checkpoint / TFRecord readers (no GPU): CRC-32C (Castagnoli), the checksum both the TF
// tensor-bundle format (`model.ckpt.index` / `.data-*`, written by the reference's Estimator, loaded through
// utils/model_utils.py:388-413) and the TFRecord framing (data/process.py:236-256) use.
#include <cstdint>
#include <cstring>

#include <nmmintrin.h>

#include "../../include/merlot_hip.h"

namespace {

struct Tables {
    uint32_t t[8][256];
    Tables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 255];
    }
};

uint32_t crc_sw(uint32_t crc, const uint8_t* p, int64_t n) {
    static const Tables T;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= crc;
        crc = T.t[7][w & 255] ^ T.t[6][(w >> 8) & 255] ^ T.t[5][(w >> 16) & 255] ^ T.t[4][(w >> 24) & 255] ^
              T.t[3][(w >> 32) & 255] ^ T.t[2][(w >> 40) & 255] ^ T.t[1][(w >> 48) & 255] ^ T.t[0][w >> 56];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 255];
    return crc;
}

__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t crc, const uint8_t* p, int64_t n) {
    uint64_t c = crc;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        c = _mm_crc32_u64(c, w);
        p += 8;
        n -= 8;
    }
    uint32_t c32 = (uint32_t)c;
    while (n-- > 0) c32 = _mm_crc32_u8(c32, *p++);
    return c32;
}

}  // namespace

// crc = 0 starts a new checksum; feed the previous return value to extend it.  `force_sw` != 0 takes the table path
// (tests compare the two).  Returns the UNMASKED CRC-32C in the low 32 bits.
extern "C" int64_t merlot_crc32c(uint64_t crc, const void* data, int64_t n, int force_sw) {
    if (n <= 0 || !data) return (int64_t)(crc & 0xffffffffu);
    const uint32_t c0 = ~(uint32_t)crc;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    const uint32_t c = (!force_sw && __builtin_cpu_supports("sse4.2")) ? crc_hw(c0, p, n) : crc_sw(c0, p, n);
    return (int64_t)(uint32_t)~c;
}
