// PNG -> RGBA8 decode with zlib (the reference uses stb_image, Texture.cpp:9, which is not vendored in its tree and not in
// this image).  Every colour type and bit depth of the PNG specification, Adam7 interlacing included, reduced to 8 bits
// the way stb_image's 8-bit API does it: 1/2/4-bit grey scaled by 255/(2^d - 1), 16-bit samples cut to their high byte, a
// tRNS colour key turned into an alpha channel, palette entries looked up (alpha from tRNS).  Output follows
// glTexImage2D(GL_RGBA, ..., format-by-channel-count) of Texture.cpp:21-28: 4ch -> RGBA, 3ch -> RGB + A=255, 1ch -> R
// only (G=B=0, A=255), 2ch -> the constructor's own exception; row 0 = first row in file.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt_host {

struct ImageRGBA8 { uint32_t width = 0, height = 0; int channels = 0; std::vector<uint8_t> rgba; };

namespace png_detail {

// One (sub-)image of w x h pixels: undo the row filters in place.  rowBytes excludes the filter byte; bpp = bytes per
// whole pixel, at least 1 (PNG specification 9.2).
inline void unfilter(uint8_t* raw, uint32_t h, size_t rowBytes, size_t bpp) {
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* cur = raw + (rowBytes + 1) * y + 1;
        const uint8_t* up = y ? cur - (rowBytes + 1) : nullptr;
        const uint8_t ft = cur[-1];
        if (ft > 4) throw std::runtime_error("png: bad filter");
        for (size_t x = 0; x < rowBytes; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = cur[x];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: break;
            }
            cur[x] = (uint8_t)v;
        }
    }
}

}  // namespace png_detail

inline ImageRGBA8 decode_png(const std::vector<uint8_t>& f) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) throw std::runtime_error("png: bad signature");
    auto be32 = [&](size_t o) { return ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]; };
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t o = 8;
    while (o + 12 <= f.size()) {
        uint32_t len = be32(o);
        std::string type((const char*)&f[o + 4], 4);
        if (o + 12 + (size_t)len > f.size()) throw std::runtime_error("png: truncated chunk");
        const uint8_t* d = &f[o + 8];
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("png: IHDR chunk is not 13 bytes");
            W = be32(o + 8); H = be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12];
        }
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "tRNS") trns.assign(d, d + len);
        else if (type == "IEND") break;
        o += 12 + (size_t)len;
    }
    // a header that is missing, empty or absurd (the pixel buffers below are sized from it) is refused before anything is allocated
    if (W == 0 || H == 0 || (uint64_t)W * H > (1ull << 28)) throw std::runtime_error("png: missing IHDR or image size out of range (1 .. 2^28 pixels)");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error("png: bad colour type");
    const bool depthOk = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                       : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depthOk || interlace > 1) throw std::runtime_error("png: bad bit depth or interlace method");
    const size_t bitsPerPixel = (size_t)ch * depth;
    const size_t filterBpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    auto rowBytesOf = [&](uint32_t w) { return ((size_t)w * bitsPerPixel + 7) / 8; };

    // the (sub-)images of the stream: one, or the seven passes of Adam7
    struct Pass { uint32_t x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, W, H});
    else {
        static const uint32_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int p = 0; p < 7; p++) {
            const uint32_t w = (W - X0[p] + DX[p] - 1) / DX[p], h = (H - Y0[p] + DY[p] - 1) / DY[p];
            if (W > X0[p] && H > Y0[p] && w && h) passes.push_back({X0[p], Y0[p], DX[p], DY[p], w, h});
        }
    }
    size_t rawSize = 0;
    for (const Pass& p : passes) rawSize += (rowBytesOf(p.w) + 1) * p.h;
    std::vector<uint8_t> raw(rawSize);
    uLongf outLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outLen, idat.data(), (uLong)idat.size()) != Z_OK || outLen != raw.size())
        throw std::runtime_error("png: inflate failed");

    // samples as 16-bit values, ch per pixel, in image order
    std::vector<uint16_t> smp((size_t)W * H * ch);
    size_t off = 0;
    for (const Pass& p : passes) {
        const size_t rb = rowBytesOf(p.w);
        png_detail::unfilter(raw.data() + off, p.h, rb, filterBpp);
        for (uint32_t y = 0; y < p.h; y++) {
            const uint8_t* row = raw.data() + off + (rb + 1) * y + 1;
            for (uint32_t x = 0; x < p.w; x++) {
                uint16_t* dst = &smp[((size_t)(p.y0 + y * p.dy) * W + (p.x0 + x * p.dx)) * ch];
                for (int k = 0; k < ch; k++) {
                    const size_t s = (size_t)x * ch + k;          // sample index in the row
                    if (depth == 16) dst[k] = (uint16_t)((row[2 * s] << 8) | row[2 * s + 1]);
                    else if (depth == 8) dst[k] = row[s];
                    else { const size_t bit = s * depth; dst[k] = (uint16_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                }
            }
        }
        off += (rb + 1) * p.h;
    }

    // reduce to what stbi_load(..., 0) returns: 8-bit samples and its channel count
    static const int scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};          // grey of 1 / 2 / 4 bits spread over 0..255
    const bool keyed = !trns.empty() && (ctype == 0 || ctype == 2) && trns.size() >= (size_t)ch * 2;
    uint16_t key[3] = {0, 0, 0};
    if (keyed) for (int k = 0; k < ch; k++) key[k] = (uint16_t)((trns[2 * k] << 8) | trns[2 * k + 1]);
    ImageRGBA8 im; im.width = W; im.height = H;
    im.channels = ctype == 3 ? (trns.empty() ? 3 : 4) : ch + (keyed ? 1 : 0);
    if (im.channels == 2) throw std::runtime_error("png: 2-channel images: Texture.cpp:27 throws 'Automatic Texture type recognition failed'");
    im.rgba.assign((size_t)W * H * 4, 0);
    auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (ctype == 0 && depth < 8) ? (uint8_t)(v * scale[depth]) : (uint8_t)v; };
    for (size_t i = 0; i < (size_t)W * H; i++) {
        uint8_t* q = &im.rgba[i * 4];
        const uint16_t* p = &smp[i * ch];
        if (ctype == 3) {
            const size_t k = p[0];
            const bool ok = k * 3 + 2 < plte.size();
            q[0] = ok ? plte[k * 3] : 0; q[1] = ok ? plte[k * 3 + 1] : 0; q[2] = ok ? plte[k * 3 + 2] : 0;
            q[3] = k < trns.size() ? trns[k] : 255;
            continue;
        }
        bool transparent = keyed;
        if (keyed) for (int k = 0; k < ch; k++) if ((depth == 16 ? p[k] : (uint16_t)(p[k] & 255)) != (depth == 16 ? key[k] : (uint16_t)(key[k] & 255))) transparent = false;
        if (ctype == 6) { q[0] = to8(p[0]); q[1] = to8(p[1]); q[2] = to8(p[2]); q[3] = to8(p[3]); }
        else if (ctype == 2) { q[0] = to8(p[0]); q[1] = to8(p[1]); q[2] = to8(p[2]); q[3] = transparent ? 0 : 255; }
        else { q[0] = to8(p[0]); q[1] = 0; q[2] = 0; q[3] = 255; }                    // GL_RED upload (grey + key is 2 channels: refused above)
    }
    return im;
}

inline std::vector<uint8_t> read_file_bytes(const std::string& path) {
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("cannot open " + path);   // get_file_contents throws errno, shaderClass.cpp:7-19
    std::vector<uint8_t> out;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(fp);
    return out;
}

}  // namespace rt_host
