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

// ---- tf.train.Example index (tensorflow/core/example/{example,feature}.proto) ---------------------------------------
// One pass over a serialized Example: Example{features=1: Features{feature=1: map<string, Feature>}}, Feature is a oneof
// {bytes_list=1, float_list=2, int64_list=3}, each with `repeated value = 1` (scalars packed or not).  The record parser
// of the input pipeline (model/dataloader.py:33-54 `_decode_record`) then only slices: no per-field Python.
namespace {

struct Cur {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) break;
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    // next field: number, wire type, and for length-delimited fields the payload [lo, hi)
    bool field(uint32_t& num, uint32_t& wt, uint64_t& val, const uint8_t*& lo, const uint8_t*& hi) {
        if (p >= end || !ok) return false;
        const uint64_t tag = varint();
        num = (uint32_t)(tag >> 3);
        wt = (uint32_t)(tag & 7);
        lo = hi = nullptr;
        val = 0;
        if (wt == 0) {
            val = varint();
        } else if (wt == 1) {
            if (end - p < 8) return ok = false;
            memcpy(&val, p, 8);
            p += 8;
        } else if (wt == 5) {
            if (end - p < 4) return ok = false;
            uint32_t v32;
            memcpy(&v32, p, 4);
            val = v32;
            p += 4;
        } else if (wt == 2) {
            const uint64_t n = varint();
            if (!ok || (uint64_t)(end - p) < n) return ok = false;
            lo = p;
            hi = p + n;
            p = hi;
        } else {
            return ok = false;
        }
        return ok;
    }
};

}  // namespace

// rows[i] = {key offset, key length, kind (1 bytes, 2 float, 3 int64, 0 empty feature), a, b, count}:
//   bytes: a/b = offset/length of the FIRST value, count = number of values; float / int64: a = first index into
//   fvals / ivals, count = number of values.  Returns the number of features, -1 malformed input, -2 an output is too small.
extern "C" int64_t merlot_example_index(const void* buf, int64_t n, int64_t* rows, int64_t max_rows, int64_t* ivals,
                                        int64_t max_ivals, float* fvals, int64_t max_fvals) {
    if (!buf || n < 0 || !rows) return -1;
    const uint8_t* base = static_cast<const uint8_t*>(buf);
    int64_t nrows = 0, ni = 0, nf = 0;
    Cur ex{base, base + n};
    uint32_t num, wt;
    uint64_t val;
    const uint8_t *lo, *hi;
    while (ex.field(num, wt, val, lo, hi)) {
        if (num != 1 || wt != 2) continue;
        Cur feats{lo, hi};
        const uint8_t *elo, *ehi;
        while (feats.field(num, wt, val, elo, ehi)) {
            if (num != 1 || wt != 2) continue;
            Cur entry{elo, ehi};
            const uint8_t *klo = nullptr, *khi = nullptr, *flo = nullptr, *fhi = nullptr;
            const uint8_t *a, *b;
            while (entry.field(num, wt, val, a, b)) {
                if (num == 1 && wt == 2) { klo = a; khi = b; }
                else if (num == 2 && wt == 2) { flo = a; fhi = b; }
            }
            if (!entry.ok) return -1;
            if (!klo) continue;
            if (nrows >= max_rows) return -2;
            int64_t* r = rows + 6 * nrows++;
            r[0] = klo - base; r[1] = khi - klo; r[2] = 0; r[3] = 0; r[4] = 0; r[5] = 0;
            if (!flo) continue;
            Cur feat{flo, fhi};
            const uint8_t *llo, *lhi;
            while (feat.field(num, wt, val, llo, lhi)) {
                if (wt != 2 || num < 1 || num > 3) continue;
                r[2] = num; r[3] = 0; r[4] = 0; r[5] = 0;
                Cur lst{llo, lhi};
                const uint8_t *vlo, *vhi;
                if (num == 2) r[3] = nf;
                if (num == 3) r[3] = ni;
                while (lst.field(num, wt, val, vlo, vhi)) {
                    if (num != 1) continue;
                    if (r[2] == 1) {
                        if (wt != 2) return -1;
                        if (r[5] == 0) { r[3] = vlo - base; r[4] = vhi - vlo; }
                        ++r[5];
                    } else if (r[2] == 2) {
                        if (wt == 2) {                    // packed
                            const int64_t c = (vhi - vlo) / 4;
                            if (nf + c > max_fvals) return -2;
                            memcpy(fvals + nf, vlo, (size_t)c * 4);
                            nf += c; r[5] += c;
                        } else if (wt == 5) {
                            if (nf + 1 > max_fvals) return -2;
                            const uint32_t v32 = (uint32_t)val;
                            memcpy(fvals + nf, &v32, 4);
                            ++nf; ++r[5];
                        } else return -1;
                    } else {
                        if (wt == 2) {                    // packed varints
                            Cur pk{vlo, vhi};
                            while (pk.p < pk.end) {
                                const uint64_t v = pk.varint();
                                if (!pk.ok) return -1;
                                if (ni + 1 > max_ivals) return -2;
                                ivals[ni++] = (int64_t)v; ++r[5];
                            }
                        } else if (wt == 0) {
                            if (ni + 1 > max_ivals) return -2;
                            ivals[ni++] = (int64_t)val; ++r[5];
                        } else return -1;
                    }
                }
                if (!lst.ok) return -1;
            }
            if (!feat.ok) return -1;
        }
        if (!feats.ok) return -1;
    }
    return ex.ok ? nrows : -1;
}
