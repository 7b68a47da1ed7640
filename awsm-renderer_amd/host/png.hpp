// png.hpp — PNG -> RGBA8 for the glTF reader's images (the reference hands image bytes to the browser's createImageBitmap;
// renderer-core/src/image.rs).  Colour types 0/2/3/4/6, bit depths 1..16, tRNS; non-interlaced.  Inflate is zlib's.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace awsm_png {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool is_png(const uint8_t* d, size_t n) { static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A}; return n >= 8 && memcmp(d, sig, 8) == 0; }

inline bool decode(const uint8_t* data, size_t len, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height, std::string& err) {
    if (!is_png(data, len)) { err = "not a PNG stream"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, have_end = false;
    while (pos + 12 <= len && !have_end) {
        const uint32_t clen = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if ((size_t)clen > len - pos - 12) { err = "truncated PNG chunk"; return false; }
        if (memcmp(type, "IHDR", 4) == 0) {
            if (clen < 13) { err = "bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) { err = "unknown PNG compression / filter method"; return false; }
            have_ihdr = true;
        } else if (memcmp(type, "PLTE", 4) == 0) plte.assign(body, body + clen);
        else if (memcmp(type, "tRNS", 4) == 0) trns.assign(body, body + clen);
        else if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + clen);
        else if (memcmp(type, "IEND", 4) == 0) have_end = true;
        pos += 12 + (size_t)clen;
    }
    if (!have_ihdr || idat.empty()) { err = "PNG without IHDR or IDAT"; return false; }
    if (w == 0 || h == 0 || w > 16384 || h > 16384) { err = "PNG size out of range"; return false; }
    if (interlace != 0) { err = "interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
        default: err = "bad PNG colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16)) { err = "bad PNG bit depth"; return false; }
    const size_t bits_pp = (size_t)channels * depth, stride = (w * bits_pp + 7) / 8, bpp = (bits_pp + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    const int zrc = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || out_len != raw.size()) { err = "PNG inflate failed"; return false; }
    // unfilter in place (each scanline: filter byte + stride bytes)
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* line = raw.data() + (size_t)y * (stride + 1);
        const uint8_t ft = line[0];
        uint8_t* cur = line + 1;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: err = "bad PNG filter type"; return false;
            }
            cur[i] = (uint8_t)v;
        }
        memcpy(prev.data(), cur, stride);
    }
    rgba.assign((size_t)w * h * 4, 255);
    const int maxv = (1 << (depth > 8 ? 8 : depth)) - 1;
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* cur = raw.data() + (size_t)y * (stride + 1) + 1;
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* o = rgba.data() + ((size_t)y * w + x) * 4;
            auto sample = [&](int ch) -> int {   // channel value; 16-bit keeps the high byte, sub-byte depths stay unscaled
                if (depth == 16) return cur[((size_t)x * channels + ch) * 2];
                if (depth == 8) return cur[(size_t)x * channels + ch];
                const size_t bit = (size_t)x * depth;
                return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
            };
            auto raw16 = [&](int ch) -> int { return depth == 16 ? (cur[((size_t)x * channels + ch) * 2] << 8) | cur[((size_t)x * channels + ch) * 2 + 1] : sample(ch); };
            if (ctype == 3) {
                const int idx = sample(0);
                if ((size_t)idx * 3 + 2 < plte.size()) { o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; } else { o[0] = o[1] = o[2] = 0; }
                o[3] = (size_t)idx < trns.size() ? trns[idx] : 255;
            } else if (ctype == 0 || ctype == 4) {
                const int g = sample(0);
                const uint8_t g8 = depth < 8 ? (uint8_t)(g * 255 / maxv) : (uint8_t)g;
                o[0] = o[1] = o[2] = g8;
                if (ctype == 4) o[3] = (uint8_t)sample(1);
                else if (trns.size() >= 2 && raw16(0) == ((trns[0] << 8) | trns[1])) o[3] = 0;
            } else {
                o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2);
                if (ctype == 6) o[3] = (uint8_t)sample(3);
                else if (trns.size() >= 6 && raw16(0) == ((trns[0] << 8) | trns[1]) && raw16(1) == ((trns[2] << 8) | trns[3]) && raw16(2) == ((trns[4] << 8) | trns[5])) o[3] = 0;
            }
        }
    }
    width = w; height = h;
    return true;
}

}  // namespace awsm_png
