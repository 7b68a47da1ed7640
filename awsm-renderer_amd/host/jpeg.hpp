// jpeg.hpp — baseline / extended-sequential / progressive JPEG (SOF0, SOF1, SOF2; 8-bit; Huffman) -> RGBA8 for the glTF reader's
// images (the reference hands image bytes to the browser's createImageBitmap, renderer-core/src/image.rs).  Grayscale and YCbCr
// (JFIF) with any sampling factors up to 4x4, interleaved and non-interleaved scans, spectral selection and successive approximation,
// restart intervals, Adobe RGB marker.  Arithmetic coding, 12-bit and CMYK streams are refused.  Chroma is upsampled by replication;
// the inverse DCT is the separable float one.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace awsm_jpeg {

inline bool is_jpeg(const uint8_t* d, size_t n) { return n > 3 && d[0] == 0xFF && d[1] == 0xD8 && d[2] == 0xFF; }

struct Huff { uint8_t bits[17] = {}; uint8_t vals[256] = {}; int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {}; bool present = false; };

inline void huff_build(Huff& h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h.valptr[l] = k; h.mincode[l] = code;
        code += h.bits[l]; k += h.bits[l];
        h.maxcode[l] = h.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7FFFFFFF;
    h.present = true;
}

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0; bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    const int b2 = p < end ? *p : 0xD9;
                    if (b2 == 0) p++;                       // stuffed zero
                    else { hit_marker = true; p--; b = 0; } // a marker: feed zeros, leave it for the caller
                }
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int bit() { if (n == 0) fill(); const int v = (int)(acc >> 31); acc <<= 1; n--; return v; }
    int bits(int c) { if (c == 0) return 0; if (n < c) fill(); const int v = (int)(acc >> (32 - c)); acc <<= c; n -= c; return v; }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int huff_decode(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}
inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

inline void idct8x8(const float* in, uint8_t* out, int stride) {
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { float s = 0; for (int u = 0; u < 8; u++) s += c[x][u] * in[y * 8 + u]; tmp[y * 8 + x] = s; }
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
        float s = 0;
        for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
        const int o = (int)std::lrintf(s + 128.0f);
        out[y * stride + x] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
}

// One scan of the entropy-coded data into the coefficient arrays (T.81 F.2 sequential, G.1 progressive: DC / AC, first / refining).
struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int bw = 0, bh = 0;            // blocks per row / column of the coefficient array (padded to whole MCUs)
    int cw = 0, ch = 0;            // blocks that carry picture (non-interleaved scans cover exactly these)
    std::vector<int16_t> coef;     // bw * bh * 64, natural (row-major) order inside a block
    std::vector<uint8_t> plane;
};

inline bool decode_block(BitReader& br, int16_t* blk, Comp& c, const Huff& hdc, const Huff& hac, bool progressive, int Ss, int Se, int Ah, int Al,
                         int& eobrun, const uint8_t* zigzag, std::string& err) {
    if (!progressive) {
        const int t = huff_decode(br, hdc);
        if (t < 0 || t > 11) { err = "corrupt JPEG data (DC)"; return false; }
        c.pred += extend(br.bits(t), t);
        blk[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = huff_decode(br, hac);
            if (rs < 0) { err = "corrupt JPEG data (AC)"; return false; }
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) { err = "corrupt JPEG data (run)"; return false; }
            blk[zigzag[k]] = (int16_t)extend(br.bits(sz), sz);
            k++;
        }
        return true;
    }
    if (Ss == 0) {                                       // DC scan
        if (Ah == 0) {
            const int t = huff_decode(br, hdc);
            if (t < 0 || t > 11) { err = "corrupt JPEG data (DC)"; return false; }
            c.pred += extend(br.bits(t), t);
            blk[0] = (int16_t)(c.pred * (1 << Al));
        } else if (br.bit()) blk[0] = (int16_t)(blk[0] | (1 << Al));
        return true;
    }
    const int p1 = 1 << Al, m1 = -(1 << Al);
    if (Ah == 0) {                                       // AC, first pass over a band
        if (eobrun > 0) { eobrun--; return true; }
        for (int k = Ss; k <= Se;) {
            const int rs = huff_decode(br, hac);
            if (rs < 0) { err = "corrupt JPEG data (AC)"; return false; }
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
                k += 16;
                continue;
            }
            k += r;
            if (k > Se) { err = "corrupt JPEG data (run)"; return false; }
            blk[zigzag[k]] = (int16_t)(extend(br.bits(sz), sz) * p1);
            k++;
        }
        return true;
    }
    int k = Ss;                                          // AC, refining pass (G.1.2.3)
    if (eobrun == 0) {
        for (; k <= Se; k++) {
            const int rs = huff_decode(br, hac);
            if (rs < 0) { err = "corrupt JPEG data (AC)"; return false; }
            int r = rs >> 4, sv = rs & 15;
            if (sv == 0) {
                if (r < 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }
            } else {
                if (sv != 1) { err = "corrupt JPEG data (refinement)"; return false; }
                sv = br.bit() ? p1 : m1;
            }
            for (; k <= Se; k++) {                       // past r still-zero coefficients, correcting the non-zero ones on the way
                int16_t& co = blk[zigzag[k]];
                if (co != 0) { if (br.bit() && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1)); }
                else { if (r == 0) { if (sv) co = (int16_t)sv; break; } r--; }
            }
        }
    }
    if (eobrun > 0) {
        for (; k <= Se; k++) {
            int16_t& co = blk[zigzag[k]];
            if (co != 0 && br.bit() && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
        }
        eobrun--;
    }
    return true;
}

inline bool decode(const uint8_t* data, size_t len, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height, std::string& err) {
    static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (!is_jpeg(data, len)) { err = "not a JPEG stream"; return false; }
    float qt[4][64] = {};
    bool qt_present[4] = {};
    Huff dc[4], ac[4];
    Comp comps[4];
    int ncomp = 0, W = 0, H = 0, restart = 0, adobe_transform = -1, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    size_t pos = 2;
    bool have_sof = false, progressive = false, have_scan = false;
    while (pos + 4 <= len) {
        if (data[pos] != 0xFF) { pos++; continue; }
        const int m = data[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { pos += 2; continue; }
        if (m == 0xD9) break;
        const size_t seg = ((size_t)data[pos + 2] << 8) | data[pos + 3];
        if (seg < 2 || pos + 2 + seg > len) { err = "truncated JPEG segment"; return false; }
        const uint8_t* s = data + pos + 4;
        const size_t sl = seg - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                i++;
                if (tq > 3 || i + (pq ? 128 : 64) > sl) { err = "bad DQT"; return false; }
                for (int k = 0; k < 64; k++) { qt[tq][zigzag[k]] = pq ? (float)((s[i] << 8) | s[i + 1]) : (float)s[i]; i += pq ? 2 : 1; }
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i + 17 <= sl) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) { err = "bad DHT"; return false; }
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[i + l]; total += h.bits[l]; }
                i += 17;
                if (total > 256 || i + total > sl) { err = "bad DHT"; return false; }
                memcpy(h.vals, s + i, total);
                i += total;
                huff_build(h);
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (have_sof) { err = "JPEG with more than one frame"; return false; }
            if (sl < 6 || s[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
            progressive = m == 0xC2;
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4]; ncomp = s[5];
            if (W <= 0 || H <= 0 || W > 16384 || H > 16384) { err = "JPEG size out of range"; return false; }
            if (!(ncomp == 1 || ncomp == 3) || sl < 6 + (size_t)ncomp * 3) { err = ncomp == 4 ? "CMYK JPEG is not supported" : "bad JPEG component count"; return false; }
            for (int c = 0; c < ncomp; c++) {
                comps[c].id = s[6 + c * 3]; comps[c].h = s[7 + c * 3] >> 4; comps[c].v = s[7 + c * 3] & 15; comps[c].tq = s[8 + c * 3];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) { err = "bad JPEG sampling factors"; return false; }
                hmax = hmax > comps[c].h ? hmax : comps[c].h; vmax = vmax > comps[c].v ? vmax : comps[c].v;
            }
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int c = 0; c < ncomp; c++) {
                Comp& k = comps[c];
                k.bw = mcux * k.h; k.bh = mcuy * k.v;
                k.cw = ((W * k.h + hmax - 1) / hmax + 7) / 8; k.ch = ((H * k.v + vmax - 1) / vmax + 7) / 8;
                k.coef.assign((size_t)k.bw * k.bh * 64, 0);
            }
            have_sof = true;
        }
        else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) { err = "this JPEG coding process is not supported"; return false; }
        else if (m == 0xDD) { if (sl >= 2) restart = (s[0] << 8) | s[1]; }
        else if (m == 0xEE) { if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11]; }
        else if (m == 0xDA) {
            if (!have_sof) { err = "SOS before SOF"; return false; }
            const int ns = s[0];
            if (ns < 1 || ns > ncomp || sl < 1 + (size_t)ns * 2 + 3) { err = "bad SOS"; return false; }
            int scan[4];
            for (int k = 0; k < ns; k++) {
                int ci = -1;
                for (int c = 0; c < ncomp; c++) if (comps[c].id == s[1 + k * 2]) ci = c;
                if (ci < 0) { err = "bad SOS component"; return false; }
                comps[ci].td = s[2 + k * 2] >> 4; comps[ci].ta = s[2 + k * 2] & 15;
                if (comps[ci].td > 3 || comps[ci].ta > 3) { err = "bad SOS table index"; return false; }
                scan[k] = ci;
            }
            const int Ss = progressive ? s[1 + ns * 2] : 0, Se = progressive ? s[2 + ns * 2] : 63, Ah = progressive ? s[3 + ns * 2] >> 4 : 0, Al = progressive ? s[3 + ns * 2] & 15 : 0;
            if (Ss > Se || Se > 63 || Al > 13 || (progressive && Ss == 0 && Se != 0) || (progressive && Ss > 0 && ns != 1)) { err = "bad progressive scan parameters"; return false; }
            for (int k = 0; k < ns; k++) {
                const Comp& c = comps[scan[k]];
                const bool need_dc = !progressive || (Ss == 0 && Ah == 0), need_ac = !progressive || Ss > 0;
                if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) { err = "JPEG scan refers to a missing table"; return false; }
            }
            BitReader br{data + pos + 2 + seg, data + len};
            int until_restart = restart, eobrun = 0;
            for (int c = 0; c < ncomp; c++) comps[c].pred = 0;
            auto on_restart = [&]() {        // RSTn: byte align, skip the marker, reset the predictors and the end-of-band run
                br.reset();
                while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
                if (br.p + 1 < br.end) br.p += 2;
                for (int c = 0; c < ncomp; c++) comps[c].pred = 0;
                eobrun = 0;
                until_restart = restart;
            };
            if (ns == 1) {                   // non-interleaved: the component's own block grid, one block per "MCU"
                Comp& c = comps[scan[0]];
                for (int by = 0; by < c.ch; by++) for (int bx = 0; bx < c.cw; bx++) {
                    if (restart && until_restart == 0) on_restart();
                    if (!decode_block(br, c.coef.data() + ((size_t)by * c.bw + bx) * 64, c, dc[c.td], ac[c.ta], progressive, Ss, Se, Ah, Al, eobrun, zigzag, err)) return false;
                    if (restart) until_restart--;
                }
            } else {
                for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++) {
                    if (restart && until_restart == 0) on_restart();
                    for (int k = 0; k < ns; k++) {
                        Comp& c = comps[scan[k]];
                        for (int by = 0; by < c.v; by++) for (int bx = 0; bx < c.h; bx++)
                            if (!decode_block(br, c.coef.data() + ((size_t)(my * c.v + by) * c.bw + (size_t)(mx * c.h + bx)) * 64, c, dc[c.td], ac[c.ta], progressive, Ss, Se, Ah, Al, eobrun, zigzag, err)) return false;
                    }
                    if (restart) until_restart--;
                }
            }
            have_scan = true;
            // the next marker that is neither a stuffed 0xFF00 nor a restart marker ends the scan's data
            size_t q = pos + 2 + seg;
            while (q + 1 < len && !(data[q] == 0xFF && data[q + 1] != 0x00 && data[q + 1] != 0xFF && !(data[q + 1] >= 0xD0 && data[q + 1] <= 0xD7))) q++;
            pos = q;
            continue;
        }
        pos += 2 + seg;
    }
    if (!have_scan) { err = "JPEG without a scan"; return false; }
    // dequantise + inverse DCT, then colour conversion with replicated chroma
    for (int c = 0; c < ncomp; c++) {
        Comp& k = comps[c];
        if (!qt_present[k.tq]) { err = "JPEG component refers to a missing quantisation table"; return false; }
        k.plane.assign((size_t)k.bw * 8 * k.bh * 8, 0);
        for (int by = 0; by < k.bh; by++) for (int bx = 0; bx < k.bw; bx++) {
            const int16_t* co = k.coef.data() + ((size_t)by * k.bw + bx) * 64;
            float blk[64];
            for (int i = 0; i < 64; i++) blk[i] = (float)co[i] * qt[k.tq][i];
            idct8x8(blk, k.plane.data() + (size_t)(by * 8) * (k.bw * 8) + (size_t)bx * 8, k.bw * 8);
        }
    }
    rgba.assign((size_t)W * H * 4, 255);
    const bool ycc = ncomp == 3 && adobe_transform != 0;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        uint8_t* o = rgba.data() + ((size_t)y * W + x) * 4;
        int v[3] = {0, 0, 0};
        for (int c = 0; c < ncomp; c++) v[c] = comps[c].plane[(size_t)(y * comps[c].v / vmax) * (comps[c].bw * 8) + (size_t)(x * comps[c].h / hmax)];
        if (ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)v[0]; }
        else if (!ycc) { o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2]; }
        else {
            const float Y = (float)v[0], cb = (float)v[1] - 128.0f, cr = (float)v[2] - 128.0f;
            auto cl = [](float f) { const int i = (int)std::lrintf(f); return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i)); };
            o[0] = cl(Y + 1.402f * cr); o[1] = cl(Y - 0.344136f * cb - 0.714136f * cr); o[2] = cl(Y + 1.772f * cb);
        }
    }
    width = (uint32_t)W; height = (uint32_t)H;
    return true;
}

}  // namespace awsm_jpeg
