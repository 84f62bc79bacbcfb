// JPEG -> RGBA8 decode (the reference loads textures with stb_image, Texture.cpp:9, which is not vendored in its tree:
// PARITY UNPINNED for this file — see tests/test_host_textures.py).  Baseline and progressive Huffman JPEG (ITU T.81),
// 8-bit, 1 or 3 components, any sampling factors, restart intervals; written from the standard, with the fixed-point
// arithmetic stb_image documents for the three places where decoders differ in the last bit:
//   * inverse DCT: the 12-bit-constant "islow" butterfly, column pass kept at 2 extra bits, rows rounded at bit 17;
//   * chroma upsampling: (3 near + far + 2) >> 2 along one axis, (9, 3, 3, 1) / 16 with + 8 for 2 x 2, replication otherwise;
//   * YCbCr -> RGB: 20-bit fixed point with the constants rounded at 12 bits, the Cb term of green truncated to 16 bits.
// The two JPEGs the reference's models carry (sphere: progressive 4:2:0, burger: baseline 4:2:0) are within 2 / 255 of
// libjpeg-turbo's decode (PIL), which has its own last-bit choices.  Arithmetic-coded, 12-bit, lossless and CMYK files are
// refused.  Output as png.h: 3 components -> RGB + A = 255, 1 component -> R only (GL_RED upload, Texture.cpp:25-26).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "png.h"

namespace rt_host {
namespace jpeg_detail {

static const uint8_t kZigzag[64 + 15] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};      // a corrupt run past the end lands on the last coefficient

struct Huff {
    bool set = false;
    uint8_t values[256];
    int maxcode[18];       // largest code of each length, left-aligned to 16 bits, + 1
    int delta[17];         // value index = code - delta[length]
    int mincode[17];
    void build(const uint8_t* counts, const uint8_t* vals, int n) {
        std::memcpy(values, vals, (size_t)n);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            delta[len] = k - code;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            if (counts[len - 1] && code - 1 >= (1 << len)) throw std::runtime_error("jpeg: bad Huffman code lengths");
            maxcode[len] = code;       // one past the last code of this length
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        set = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, dcPred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;          // size in pixels; padded size (whole MCUs)
    int coeffW = 0, coeffH = 0;                // blocks per row / column of the padded plane
    std::vector<uint8_t> data;                 // decoded plane, w2 x h2
    std::vector<int16_t> coeff;                // progressive: all coefficients, 64 per block
};

struct Decoder {
    const uint8_t* p; const uint8_t* end;
    uint16_t dequant[4][64];
    Huff hdc[4], hac[4];
    Component comp[3];
    int nComp = 0, width = 0, height = 0, hMax = 1, vMax = 1, mcuW = 0, mcuH = 0, mcuX = 0, mcuY = 0;
    bool progressive = false, adobeRgb = false, sawSof = false;
    int restartInterval = 0;
    // scan state
    int scanN = 0, order[3] = {0, 0, 0};
    int specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0, todo = 0;
    // bit reader
    uint32_t bitBuf = 0; int bitCnt = 0; bool noMore = false; int marker = -1;

    Decoder(const uint8_t* b, const uint8_t* e) : p(b), end(e) { std::memset(dequant, 0, sizeof dequant); }

    int u8() { if (p >= end) throw std::runtime_error("jpeg: truncated file"); return *p++; }
    int u16() { int a = u8(); return (a << 8) | u8(); }

    void resetBits() { bitBuf = 0; bitCnt = 0; noMore = false; marker = -1; eobRun = 0; for (Component& c : comp) c.dcPred = 0; todo = restartInterval ? restartInterval : 0x7fffffff; }
    void fill() {
        while (bitCnt <= 24) {
            int b = 0;
            if (!noMore) {
                b = p < end ? *p++ : 0;
                if (b == 0xff) {
                    int c = p < end ? *p++ : 0xd9;
                    while (c == 0xff) c = p < end ? *p++ : 0xd9;      // fill bytes
                    if (c != 0) { marker = c; noMore = true; b = 0; }
                }
            }
            bitBuf |= (uint32_t)b << (24 - bitCnt);
            bitCnt += 8;
        }
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (bitCnt < n) fill();
        const int v = (int)(bitBuf >> (32 - n));
        bitBuf <<= n; bitCnt -= n;
        return v;
    }
    int bit() { return bits(1); }
    int decode(const Huff& h) {
        if (!h.set) throw std::runtime_error("jpeg: scan uses a Huffman table that was not defined");
        if (bitCnt < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (int)(bitBuf >> (32 - len));
            if (code < h.maxcode[len] && code >= h.mincode[len]) {
                bitBuf <<= len; bitCnt -= len;
                return h.values[(code + h.delta[len]) & 255];
            }
        }
        throw std::runtime_error("jpeg: bad Huffman code");
    }
    // T.81 F.2.2.1 EXTEND
    int receiveExtend(int n) {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

    // coefficient arithmetic that wraps instead of overflowing (a valid file stays far inside 16 bits)
    static inline int wrapAdd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
    static inline int16_t wrapMul(int a, int b) { return (int16_t)(uint16_t)((uint32_t)a * (uint32_t)b); }

    // ---- entropy-coded blocks ----
    void blockBaseline(int16_t* d, Component& c) {
        const Huff& dc = hdc[c.td]; const Huff& ac = hac[c.ta];
        const uint16_t* q = dequant[c.tq];
        std::memset(d, 0, 64 * sizeof(int16_t));
        const int t = decode(dc);
        if (t > 15) throw std::runtime_error("jpeg: bad DC size");
        c.dcPred = wrapAdd(c.dcPred, receiveExtend(t));
        d[0] = wrapMul(c.dcPred, q[0]);
        int k = 1;
        do {
            const int rs = decode(ac), s = rs & 15, r = rs >> 4;
            if (s == 0) { if (rs != 0xf0) break; k += 16; }
            else { k += r; const int z = kZigzag[k++]; d[z] = wrapMul(receiveExtend(s), q[z]); }
        } while (k < 64);
    }
    void blockProgDc(int16_t* d, Component& c) {
        if (specEnd != 0) throw std::runtime_error("jpeg: DC scan with AC coefficients");
        if (succHigh == 0) {
            const int t = decode(hdc[c.td]);
            if (t > 15) throw std::runtime_error("jpeg: bad DC size");
            c.dcPred = wrapAdd(c.dcPred, receiveExtend(t));
            d[0] = wrapMul(c.dcPred, 1 << succLow);
        } else if (bit()) d[0] = (int16_t)(d[0] + (1 << succLow));
    }
    void blockProgAc(int16_t* d, Component& c) {
        if (specStart == 0) throw std::runtime_error("jpeg: AC scan that starts at the DC coefficient");
        const Huff& ac = hac[c.ta];
        if (succHigh == 0) {
            if (eobRun) { --eobRun; return; }
            int k = specStart;
            do {
                const int rs = decode(ac), s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) { eobRun = 1 << r; if (r) eobRun += bits(r); --eobRun; break; }
                    k += 16;
                } else { k += r; const int z = kZigzag[k++]; d[z] = wrapMul(receiveExtend(s), 1 << succLow); }
            } while (k <= specEnd);
        } else {
            const int16_t one = (int16_t)(1 << succLow);
            auto refine = [&](int16_t* q) { if (bit() && (*q & one) == 0) *q = (int16_t)(*q > 0 ? *q + one : *q - one); };
            if (eobRun) {
                --eobRun;
                for (int k = specStart; k <= specEnd; k++) { int16_t* q = &d[kZigzag[k]]; if (*q != 0) refine(q); }
            } else {
                int k = specStart;
                do {
                    const int rs = decode(ac);
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) { eobRun = (1 << r) - 1; if (r) eobRun += bits(r); r = 64; }      // the rest of the band only gets refined
                    } else {
                        if (s != 1) throw std::runtime_error("jpeg: bad refinement code");
                        s = bit() ? one : -one;
                    }
                    while (k <= specEnd) {
                        int16_t* q = &d[kZigzag[k++]];
                        if (*q != 0) refine(q);
                        else { if (r == 0) { *q = (int16_t)s; break; } --r; }
                    }
                } while (k <= specEnd);
            }
        }
    }

    // ---- inverse DCT (see the header) ----
    static inline uint8_t clamp8(int64_t x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }
    static inline int f2f(float x) { return (int)(x * 4096 + 0.5); }
    static void idct(uint8_t* out, int stride, const int16_t* data) {
        int val[64], *v = val;
        const int16_t* d = data;
#define RT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                      \
        int64_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;      /* 64 bits: a damaged file cannot overflow them */ \
        p2 = s2; p3 = s6;                                                               \
        p1 = (p2 + p3) * f2f(0.5411961f);                                               \
        t2 = p1 + p3 * f2f(-1.847759065f);                                              \
        t3 = p1 + p2 * f2f(0.765366865f);                                               \
        p2 = s0; p3 = s4;                                                               \
        t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                   \
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                         \
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                             \
        p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                         \
        p5 = (p3 + p4) * f2f(1.175875602f);                                             \
        t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f);                       \
        t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);                       \
        p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);           \
        p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);                     \
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        for (int i = 0; i < 8; ++i, ++d, ++v) {
            if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
                const int dc = d[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
            } else {
                RT_IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (int)((x0 + t3) >> 10); v[56] = (int)((x0 - t3) >> 10);
                v[8] = (int)((x1 + t2) >> 10); v[48] = (int)((x1 - t2) >> 10);
                v[16] = (int)((x2 + t1) >> 10); v[40] = (int)((x2 - t1) >> 10);
                v[24] = (int)((x3 + t0) >> 10); v[32] = (int)((x3 - t0) >> 10);
            }
        }
        v = val;
        uint8_t* o = out;
        for (int i = 0; i < 8; ++i, v += 8, o += stride) {
            RT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
            x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
#undef RT_IDCT_1D
    }

    // ---- markers ----
    void readDqt(int len) {
        while (len > 0) {
            const int q = u8(), prec = q >> 4, t = q & 15;
            if (prec > 1 || t > 3) throw std::runtime_error("jpeg: bad quantisation table");
            for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = (uint16_t)(prec ? u16() : u8());
            len -= prec ? 129 : 65;
        }
        if (len != 0) throw std::runtime_error("jpeg: bad DQT length");
    }
    void readDht(int len) {
        while (len > 0) {
            const int q = u8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad Huffman table header");
            uint8_t counts[16], vals[256];
            int n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)u8(); n += counts[i]; }
            if (n > 256) throw std::runtime_error("jpeg: bad Huffman table size");
            for (int i = 0; i < n; i++) vals[i] = (uint8_t)u8();
            (tc ? hac[th] : hdc[th]).build(counts, vals, n);
            len -= 17 + n;
        }
        if (len != 0) throw std::runtime_error("jpeg: bad DHT length");
    }
    void readSof(int len, bool prog) {
        if (sawSof) throw std::runtime_error("jpeg: more than one frame header");
        if (len < 9) throw std::runtime_error("jpeg: bad SOF length");
        if (u8() != 8) throw std::runtime_error("jpeg: only 8-bit samples are supported");
        height = u16(); width = u16(); nComp = u8();
        if (height == 0 || width == 0 || (uint64_t)width * (uint64_t)height > (1ull << 28)) throw std::runtime_error("jpeg: image size out of range (1 .. 2^28 pixels)");
        if (nComp != 1 && nComp != 3) throw std::runtime_error("jpeg: only 1- and 3-component images are supported (no CMYK)");
        if (len != 6 + 3 * nComp) throw std::runtime_error("jpeg: bad SOF length");
        for (int i = 0; i < nComp; i++) {
            comp[i].id = u8();
            const int q = u8();
            comp[i].h = q >> 4; comp[i].v = q & 15; comp[i].tq = u8();
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw std::runtime_error("jpeg: bad component header");
            hMax = std::max(hMax, comp[i].h); vMax = std::max(vMax, comp[i].v);
        }
        for (int i = 0; i < nComp; i++) if (hMax % comp[i].h || vMax % comp[i].v) throw std::runtime_error("jpeg: fractional sampling ratios are not supported");
        progressive = prog;
        mcuW = hMax * 8; mcuH = vMax * 8;
        mcuX = (width + mcuW - 1) / mcuW; mcuY = (height + mcuH - 1) / mcuH;
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            c.x = (width * c.h + hMax - 1) / hMax; c.y = (height * c.v + vMax - 1) / vMax;
            c.w2 = mcuX * c.h * 8; c.h2 = mcuY * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            if (prog) { c.coeffW = c.w2 / 8; c.coeffH = c.h2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
        }
        sawSof = true;
    }
    void readSos(int len) {
        if (!sawSof) throw std::runtime_error("jpeg: scan before the frame header");
        scanN = u8();
        if (scanN < 1 || scanN > nComp || len != 4 + 2 * scanN) throw std::runtime_error("jpeg: bad SOS header");
        for (int i = 0; i < scanN; i++) {
            const int id = u8(), q = u8();
            int which = -1;
            for (int k = 0; k < nComp; k++) if (comp[k].id == id) which = k;
            if (which < 0 || (q >> 4) > 3 || (q & 15) > 3) throw std::runtime_error("jpeg: bad SOS component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            order[i] = which;
        }
        specStart = u8(); specEnd = u8();
        const int a = u8();
        succHigh = a >> 4; succLow = a & 15;
        if (progressive) {
            if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) throw std::runtime_error("jpeg: bad progressive scan parameters");
        } else {
            if (specStart != 0 || succHigh != 0 || succLow != 0) throw std::runtime_error("jpeg: bad baseline scan parameters");
            specEnd = 63;
        }
    }

    void restartIfDue() {
        if (--todo > 0) return;
        if (bitCnt < 24) fill();
        // an RSTn marker must follow; anything else ends the scan's data (the remaining blocks stay as they are)
        if (marker >= 0xd0 && marker <= 0xd7) resetBits();
        else { todo = 0x7fffffff; finished = true; }
    }
    bool finished = false;

    void decodeScan() {
        resetBits();
        finished = false;
        auto one = [&](Component& c, int bx, int by) {
            if (!progressive) {
                int16_t d[64];
                blockBaseline(d, c);
                idct(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, d);
            } else {
                int16_t* d = &c.coeff[64 * ((size_t)bx + (size_t)by * c.coeffW)];
                if (specStart == 0) blockProgDc(d, c); else blockProgAc(d, c);
            }
        };
        if (scanN == 1) {
            Component& c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;       // non-interleaved: the component's own blocks, row by row
            for (int j = 0; j < h && !finished; j++)
                for (int i = 0; i < w && !finished; i++) { one(c, i, j); if (restartInterval) restartIfDue(); }
        } else {
            if (progressive && specStart != 0) throw std::runtime_error("jpeg: interleaved AC scan");
            for (int j = 0; j < mcuY && !finished; j++)
                for (int i = 0; i < mcuX && !finished; i++) {
                    for (int k = 0; k < scanN; k++) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) one(c, i * c.h + x, j * c.v + y);
                    }
                    if (restartInterval) restartIfDue();
                }
        }
    }

    void finishProgressive() {
        for (int n = 0; n < nComp; n++) {
            Component& c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeffW)];
                    for (int k = 0; k < 64; k++) d[k] = wrapMul(d[k], dequant[c.tq][k]);
                    idct(&c.data[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, d);
                }
        }
    }

    // next marker in the header stream (skips fill bytes)
    int nextMarker() {
        if (marker >= 0) { const int m = marker; marker = -1; return m; }
        int c = u8();
        while (c != 0xff) { if (p >= end) return 0xd9; c = u8(); }      // junk between segments
        while (c == 0xff) c = u8();
        return c;
    }

    void run() {
        if (u8() != 0xff || u8() != 0xd8) throw std::runtime_error("jpeg: no SOI marker");
        for (;;) {
            const int m = nextMarker();
            if (m == 0xd9) break;
            if (m == 0xda) {
                readSos(u16() - 2);
                decodeScan();
                // the entropy-coded data ends at the next marker: find it if the bit reader has not
                if (marker < 0) {
                    while (p < end) {
                        if (*p++ != 0xff) continue;
                        while (p < end && *p == 0xff) p++;
                        if (p >= end) break;
                        const int c = *p++;
                        if (c != 0 && !(c >= 0xd0 && c <= 0xd7)) { marker = c; break; }
                    }
                    if (marker < 0) break;
                } else if (marker >= 0xd0 && marker <= 0xd7) marker = -1;
                continue;
            }
            if (m >= 0xd0 && m <= 0xd7) continue;
            if (m == 0x01) continue;
            const int len = u16() - 2;
            if (len < 0 || p + len > end) throw std::runtime_error("jpeg: bad segment length");
            const uint8_t* segEnd = p + len;
            switch (m) {
                case 0xc0: case 0xc1: readSof(len, false); break;
                case 0xc2: readSof(len, true); break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    throw std::runtime_error("jpeg: lossless, hierarchical and arithmetic-coded files are not supported");
                case 0xc4: readDht(len); break;
                case 0xdb: readDqt(len); break;
                case 0xdd: if (len != 2) throw std::runtime_error("jpeg: bad DRI length"); restartInterval = u16(); break;
                case 0xee:      // APP14 "Adobe": transform flag 0 on a 3-component image = RGB, not YCbCr
                    if (len >= 12 && std::memcmp(p, "Adobe", 5) == 0) adobeRgb = p[11] == 0;
                    break;
                default: break;      // APPn, COM, anything else: skipped
            }
            p = segEnd;
        }
        if (!sawSof) throw std::runtime_error("jpeg: no frame header");
        if (progressive) finishProgressive();
    }
};

// one output row of a component at full resolution (see the header for the three filters)
inline const uint8_t* upsampleRow(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return nearRow;
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2); return out; }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = nearRow;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            const int n = 3 * in[i] + 2;
            out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        // (the last-but-one output weighs its FAR neighbour 3:1 — stb_image's h2v1 filter as it is written, kept so that
        //  4:2:2 textures match the reference's loader in that column too; every other decoder differs there)
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2); return out; }
        int t1 = 3 * nearRow[0] + farRow[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * nearRow[i] + farRow[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = nearRow[i];      // other ratios: nearest
    return out;
}

}  // namespace jpeg_detail

inline ImageRGBA8 decode_jpeg(const std::vector<uint8_t>& f) {
    using namespace jpeg_detail;
    Decoder D(f.data(), f.data() + f.size());
    D.run();
    const int W = D.width, H = D.height, n = D.nComp;
    ImageRGBA8 im; im.width = (uint32_t)W; im.height = (uint32_t)H; im.channels = n;
    im.rgba.assign((size_t)W * H * 4, 0);
    struct Res { int hs, vs, ystep, wLores, ypos; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; } res[3];
    for (int k = 0; k < n; k++) {
        Component& c = D.comp[k];
        res[k].hs = D.hMax / c.h; res[k].vs = D.vMax / c.v;
        res[k].ystep = res[k].vs >> 1;
        res[k].wLores = (W + res[k].hs - 1) / res[k].hs;
        res[k].ypos = 0;
        res[k].line0 = res[k].line1 = c.data.data();
        res[k].buf.assign((size_t)W + 2 * 8 * 4 + 3, 0);
    }
    auto f2x = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
    const int kCrR = f2x(1.40200f), kCrG = -f2x(0.71414f), kCbG = -f2x(0.34414f), kCbB = f2x(1.77200f);
    for (int j = 0; j < H; j++) {
        const uint8_t* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < n; k++) {
            Res& r = res[k];
            const bool bot = r.ystep >= (r.vs >> 1);
            row[k] = upsampleRow(r.buf.data(), bot ? r.line1 : r.line0, bot ? r.line0 : r.line1, r.wLores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < D.comp[k].y) r.line1 += D.comp[k].w2;
            }
        }
        uint8_t* q = &im.rgba[(size_t)j * W * 4];
        if (n == 1) for (int i = 0; i < W; i++) { q[4 * i] = row[0][i]; q[4 * i + 3] = 255; }               // GL_RED upload
        else if (D.adobeRgb || (D.comp[0].id == 'R' && D.comp[1].id == 'G' && D.comp[2].id == 'B'))
            for (int i = 0; i < W; i++) { q[4 * i] = row[0][i]; q[4 * i + 1] = row[1][i]; q[4 * i + 2] = row[2][i]; q[4 * i + 3] = 255; }
        else
            for (int i = 0; i < W; i++) {
                const int yf = (row[0][i] << 20) + (1 << 19);
                const int cb = row[1][i] - 128, cr = row[2][i] - 128;
                int r = yf + cr * kCrR;
                int g = yf + cr * kCrG + (int)(((unsigned)(cb * kCbG)) & 0xffff0000u);
                int b = yf + cb * kCbB;
                r >>= 20; g >>= 20; b >>= 20;
                q[4 * i] = Decoder::clamp8(r); q[4 * i + 1] = Decoder::clamp8(g); q[4 * i + 2] = Decoder::clamp8(b); q[4 * i + 3] = 255;
            }
    }
    return im;
}

// stbi_load's format sniffing for the two formats glTF allows (image/png, image/jpeg)
inline ImageRGBA8 decode_image(const std::vector<uint8_t>& f) {
    if (f.size() >= 8 && f[0] == 137 && f[1] == 'P' && f[2] == 'N' && f[3] == 'G') return decode_png(f);
    if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8) return decode_jpeg(f);
    throw std::runtime_error("texture: neither a PNG nor a JPEG file");
}

}  // namespace rt_host
