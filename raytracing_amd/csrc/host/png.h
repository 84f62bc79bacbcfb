// PNG -> RGBA8 decode with zlib (the reference uses stb_image, Texture.cpp:9, not vendored and not in
// this image).  Supports 8-bit gray / gray+alpha / RGB / RGBA / palette, non-interlaced — what the
// bundled models use.  Output follows glTexImage2D(GL_RGBA, ..., format-by-channel-count) of
// Texture.cpp:21-28: 4ch -> RGBA, 3ch -> RGB + A=255, 1ch -> R only (G=B=0, A=255); row 0 = first row in file.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt_host {

struct ImageRGBA8 { uint32_t width = 0, height = 0; int channels = 0; std::vector<uint8_t> rgba; };

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
        if (o + 12 + len > f.size()) throw std::runtime_error("png: truncated chunk");
        const uint8_t* d = &f[o + 8];
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("png: IHDR chunk is not 13 bytes");
            W = be32(o + 8); H = be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12];
        }
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "tRNS") trns.assign(d, d + len);
        else if (type == "IEND") break;
        o += 12 + len;
    }
    // a header that is missing, empty or absurd (the pixel buffers below are sized from it) is refused before anything is allocated
    if (W == 0 || H == 0 || (uint64_t)W * H > (1ull << 28)) throw std::runtime_error("png: missing IHDR or image size out of range (1 .. 2^28 pixels)");
    if (depth != 8 || interlace != 0) throw std::runtime_error("png: only 8-bit non-interlaced images are supported");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error("png: bad colour type");
    size_t stride = (size_t)W * ch;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf outLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outLen, idat.data(), (uLong)idat.size()) != Z_OK || outLen != raw.size())
        throw std::runtime_error("png: inflate failed");
    std::vector<uint8_t> pix(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t* in = &raw[(stride + 1) * y];
        uint8_t ft = in[0]; in++;
        uint8_t* cur = &pix[stride * y];
        const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int v = in[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error("png: bad filter");
            }
            cur[x] = (uint8_t)v;
        }
    }
    ImageRGBA8 im; im.width = W; im.height = H;
    im.rgba.assign((size_t)W * H * 4, 0);
    // stbi_load(..., 0) channel count: gray 1, gray+alpha 2, rgb 3, rgba 4; palette expands to 3 (4 with tRNS)
    im.channels = ctype == 3 ? (trns.empty() ? 3 : 4) : ch;
    for (size_t i = 0; i < (size_t)W * H; i++) {
        uint8_t* q = &im.rgba[i * 4];
        const uint8_t* p = &pix[i * ch];
        if (ctype == 6) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3]; }
        else if (ctype == 2) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = 255; }
        else if (ctype == 0) { q[0] = p[0]; q[1] = 0; q[2] = 0; q[3] = 255; }           // GL_RED upload
        else if (ctype == 3) {
            size_t k = p[0];
            bool ok = k * 3 + 2 < plte.size();
            q[0] = ok ? plte[k * 3] : 0; q[1] = ok ? plte[k * 3 + 1] : 0; q[2] = ok ? plte[k * 3 + 2] : 0;
            q[3] = k < trns.size() ? trns[k] : 255;
        }
        else throw std::runtime_error("png: 2-channel images: Texture.cpp:27 throws 'Automatic Texture type recognition failed'");
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
