// Host half of the GPU JPEG decoder (`tf.image.decode_jpeg`, model/dataloader.py:72-77): marker parsing and the inherently
// sequential Huffman decode of a baseline (SOF0) 8-bit JPEG, YCbCr 4:4:4 or 4:2:0.  Output: the QUANTISED DCT coefficients
// of every block in natural (de-zigzagged) order + a small header; dequantisation, the inverse DCT, chroma upsampling and
// colour conversion run on the GPU (csrc/jpeg.hip), bit for bit what libjpeg(-turbo)'s default decoder computes.
// Anything else (progressive, grayscale, CMYK, 4:2:2, 12-bit, arithmetic coding) returns MERLOT_JPEG_UNSUPPORTED and the
// caller decodes that frame with the host library.
#include <cstdint>
#include <cstring>

#include "../../include/merlot_hip.h"

namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    // canonical decode tables (ITU T.81 F.2.2.3): for code length l, codes first[l] .. first[l] + count[l] - 1
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    uint16_t lut[512];                                    // 9-bit lookahead: (code length << 8) | symbol, 0 = longer than 9 bits
    bool present = false;
};

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool marker = false;                                  // ran into a marker (only RSTn is legal inside the scan)
    inline void fill() {
        while (n <= 24) {
            uint32_t b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else {
                        marker = true;                    // leave p on the marker; feed zeros
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            acc |= b << (24 - n);
            n += 8;
        }
    }
    inline int get(int k) {                               // k <= 16
        if (k == 0) return 0;
        if (n < k) fill();
        const int v = (int)(acc >> (32 - k));
        acc <<= k;
        n -= k;
        return v;
    }
    inline int bit() { return get(1); }
    void reset() { acc = 0; n = 0; marker = false; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

inline int decode_sym(Bits& b, const Huff& h) {
    if (b.n < 16) b.fill();
    const uint16_t e = h.lut[b.acc >> 23];
    if (e) {                                              // the common case: one table look-up
        const int l = e >> 8;
        b.acc <<= l;
        b.n -= l;
        return e & 255;
    }
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | b.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

}  // namespace

extern "C" int merlot_jpeg_entropy_decode(const uint8_t* data, int64_t n, merlot_jpeg_info_t* info, int16_t* coef,
                                          int64_t coef_capacity) {
    if (!data || !info || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return MERLOT_JPEG_MALFORMED;
    memset(info, 0, sizeof(*info));
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0}, comp_h[3] = {0, 0, 0}, comp_v[3] = {0, 0, 0};
    int restart_interval = 0;
    bool have_sof = false;
    int64_t pos = 2;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) return MERLOT_JPEG_MALFORMED;
        const int m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }               // fill bytes
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return MERLOT_JPEG_MALFORMED;      // EOI before a scan
        if (pos + 2 > n) return MERLOT_JPEG_MALFORMED;
        const int len = rd16(data + pos);
        if (len < 2 || pos + len > n) return MERLOT_JPEG_MALFORMED;
        const uint8_t* seg = data + pos + 2;
        const int slen = len - 2;
        if (m == 0xDB) {                                  // DQT
            int o = 0;
            while (o < slen) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (tq > 3) return MERLOT_JPEG_MALFORMED;
                if (pq != 0) return MERLOT_JPEG_UNSUPPORTED;   // 16-bit tables: 12-bit JPEG
                if (o + 65 > slen) return MERLOT_JPEG_MALFORMED;
                for (int k = 0; k < 64; ++k) qt[tq][ZIGZAG[k]] = seg[o + 1 + k];
                qt_present[tq] = true;
                o += 65;
            }
        } else if (m == 0xC4) {                           // DHT
            int o = 0;
            while (o < slen) {
                if (o + 17 > slen) return MERLOT_JPEG_MALFORMED;
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3) return MERLOT_JPEG_MALFORMED;
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0, code = 0, k = 0;
                for (int l = 1; l <= 16; ++l) {
                    const int c = seg[o + l];
                    h.valptr[l] = k;
                    h.mincode[l] = code;
                    h.maxcode[l] = c ? code + c - 1 : -1;
                    code = (code + c) << 1;
                    k += c;
                    total += c;
                }
                if (total > 256 || o + 17 + total > slen) return MERLOT_JPEG_MALFORMED;
                memcpy(h.vals, seg + o + 17, total);
                memset(h.lut, 0, sizeof(h.lut));
                for (int l = 1; l <= 9; ++l)
                    for (int cde = h.mincode[l]; h.maxcode[l] >= 0 && cde <= h.maxcode[l]; ++cde) {
                        const int sym = h.vals[h.valptr[l] + cde - h.mincode[l]];
                        if ((cde << (9 - l)) + (1 << (9 - l)) > 512) return MERLOT_JPEG_MALFORMED;       // over-subscribed table
                        for (int f = 0; f < (1 << (9 - l)); ++f) h.lut[(cde << (9 - l)) + f] = (uint16_t)((l << 8) | sym);
                    }
                h.present = true;
                o += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {              // SOF0 / SOF1 (extended sequential, Huffman): same decode
            if (slen < 6) return MERLOT_JPEG_MALFORMED;
            if (seg[0] != 8) return MERLOT_JPEG_UNSUPPORTED;
            info->height = rd16(seg + 1);
            info->width = rd16(seg + 3);
            const int nc = seg[5];
            if (nc != 3) return MERLOT_JPEG_UNSUPPORTED;  // grayscale / CMYK: host library
            if (slen < 6 + 3 * nc || info->height <= 0 || info->width <= 0) return MERLOT_JPEG_MALFORMED;
            for (int c = 0; c < 3; ++c) {
                comp_id[c] = seg[6 + 3 * c];
                comp_h[c] = seg[7 + 3 * c] >> 4;
                comp_v[c] = seg[7 + 3 * c] & 15;
                comp_tq[c] = seg[8 + 3 * c];
                if (comp_tq[c] > 3) return MERLOT_JPEG_MALFORMED;
            }
            const bool s444 = comp_h[0] == 1 && comp_v[0] == 1, s420 = comp_h[0] == 2 && comp_v[0] == 2;
            if (!(s444 || s420) || comp_h[1] != 1 || comp_v[1] != 1 || comp_h[2] != 1 || comp_v[2] != 1) return MERLOT_JPEG_UNSUPPORTED;
            info->subsampling = s420 ? 2 : 1;
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return MERLOT_JPEG_UNSUPPORTED;               // progressive, lossless, arithmetic
        } else if (m == 0xDD) {
            if (slen < 2) return MERLOT_JPEG_MALFORMED;
            restart_interval = rd16(seg);
        } else if (m == 0xEE) {                           // Adobe APP14: a transform flag other than YCbCr means RGB / CMYK data
            if (slen >= 12 && !memcmp(seg, "Adobe", 5) && seg[11] != 1) return MERLOT_JPEG_UNSUPPORTED;
        } else if (m == 0xDA) {                           // SOS: the (single, interleaved) scan
            if (!have_sof) return MERLOT_JPEG_MALFORMED;
            if (slen < 1 + 2 * 3 + 3 || seg[0] != 3) return MERLOT_JPEG_UNSUPPORTED;   // non-interleaved scans: host library
            int td[3], ta[3];
            for (int c = 0; c < 3; ++c) {
                if (seg[1 + 2 * c] != comp_id[c]) return MERLOT_JPEG_UNSUPPORTED;
                td[c] = seg[2 + 2 * c] >> 4;
                ta[c] = seg[2 + 2 * c] & 15;
                if (td[c] > 3 || ta[c] > 3 || !dc[td[c]].present || !ac[ta[c]].present || !qt_present[comp_tq[c]]) return MERLOT_JPEG_MALFORMED;
            }
            const int s = info->subsampling;              // MCU = 8s x 8s pixels
            const int mcux = (info->width + 8 * s - 1) / (8 * s), mcuy = (info->height + 8 * s - 1) / (8 * s);
            info->blocks_w[0] = mcux * s; info->blocks_h[0] = mcuy * s;
            info->blocks_w[1] = info->blocks_w[2] = mcux;
            info->blocks_h[1] = info->blocks_h[2] = mcuy;
            int64_t off = 0;
            for (int c = 0; c < 3; ++c) {
                info->coef_offset[c] = off;
                off += (int64_t)info->blocks_w[c] * info->blocks_h[c] * 64;
                for (int k = 0; k < 64; ++k) info->quant[c][k] = qt[comp_tq[c]][k];
            }
            info->coef_count = off;
            if (!coef) return MERLOT_OK;                  // size query
            if (coef_capacity < off) return MERLOT_JPEG_CAPACITY;
            memset(coef, 0, (size_t)off * sizeof(int16_t));
            Bits b;
            b.p = data + pos + len;
            b.end = data + n;
            int pred[3] = {0, 0, 0};
            int until_restart = restart_interval;
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    if (restart_interval && until_restart == 0) {
                        // byte-align, expect RSTn
                        b.reset();
                        while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) ++b.p;
                        if (b.p + 2 > b.end || b.p[0] != 0xFF || b.p[1] < 0xD0 || b.p[1] > 0xD7) return MERLOT_JPEG_MALFORMED;
                        b.p += 2;
                        pred[0] = pred[1] = pred[2] = 0;
                        until_restart = restart_interval;
                    }
                    for (int c = 0; c < 3; ++c) {
                        const int hs = c == 0 ? s : 1, vs = c == 0 ? s : 1;
                        for (int by = 0; by < vs; ++by)
                            for (int bx = 0; bx < hs; ++bx) {
                                int16_t* blk = coef + info->coef_offset[c] +
                                               ((int64_t)(my * vs + by) * info->blocks_w[c] + (mx * hs + bx)) * 64;
                                int t = decode_sym(b, dc[td[c]]);
                                if (t < 0 || t > 11) return MERLOT_JPEG_MALFORMED;
                                int diff = t ? extend(b.get(t), t) : 0;
                                pred[c] += diff;
                                blk[0] = (int16_t)pred[c];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_sym(b, ac[ta[c]]);
                                    if (rs < 0) return MERLOT_JPEG_MALFORMED;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r == 15) { k += 16; continue; }
                                        break;            // EOB
                                    }
                                    k += r;
                                    if (k > 63) return MERLOT_JPEG_MALFORMED;
                                    blk[ZIGZAG[k]] = (int16_t)extend(b.get(sz), sz);
                                    ++k;
                                }
                            }
                    }
                    if (restart_interval) --until_restart;
                }
            return MERLOT_OK;
        }
        pos += len;
    }
    return MERLOT_JPEG_MALFORMED;
}
