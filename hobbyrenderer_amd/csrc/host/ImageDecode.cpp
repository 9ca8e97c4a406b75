#include "ImageDecode.h"

#include <cstdio>
#include <cstring>

namespace hobbyrt {

// ------------------------------------------------------------------ inflate
namespace {
struct Bits {
    const uint8_t* p; const uint8_t* end; uint64_t acc = 0; int cnt = 0; bool overrun = false;
    void need(int n) { while (cnt < n) { uint64_t b = 0; if (p < end) b = *p++; else overrun = true; acc |= b << cnt; cnt += 8; } }
    uint32_t get(int n) { if (n == 0) return 0; need(n); uint32_t v = (uint32_t)(acc & ((1ull << n) - 1)); acc >>= n; cnt -= n; return v; }
    void align() { int r = cnt & 7; acc >>= r; cnt -= r; }
};
struct Huff {       // canonical code, decoded bit by bit over per-length counts (inputs are small: textures of a scene load)
    uint16_t count[16]; uint16_t symbol[288];
    bool build(const uint8_t* lengths, int n)
    {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(Bits& b) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)b.get(1);
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};
const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
}

bool Inflate(const uint8_t* data, size_t n, std::vector<uint8_t>& out, std::string& err)
{
    out.clear();
    if (n < 6) { err = "zlib stream too short"; return false; }
    if ((data[0] & 0x0F) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) { err = "bad zlib header"; return false; }
    Bits b{ data + 2, data + n };
    for (bool last = false; !last;) {
        last = b.get(1) != 0;
        uint32_t type = b.get(2);
        if (type == 0) {
            b.align();
            uint32_t len = b.get(16), nlen = b.get(16);
            if ((len ^ 0xFFFFu) != nlen) { err = "stored block length mismatch"; return false; }
            for (uint32_t i = 0; i < len; ++i) out.push_back((uint8_t)b.get(8));
        } else if (type == 1 || type == 2) {
            Huff lit, dist; uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lengths[i] = 8;
                for (int i = 144; i < 256; ++i) lengths[i] = 9;
                for (int i = 256; i < 280; ++i) lengths[i] = 7;
                for (int i = 280; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
                static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint8_t cl[19] = {};
                for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)b.get(3);
                Huff clh;
                if (hlit > 286 || hdist > 30 || !clh.build(cl, 19)) { err = "bad dynamic block header"; return false; }
                int i = 0;
                while (i < hlit + hdist) {
                    int sym = clh.decode(b);
                    if (sym < 0) { err = "bad code length code"; return false; }
                    if (sym < 16) lengths[i++] = (uint8_t)sym;
                    else {
                        int rep; uint8_t val = 0;
                        if (sym == 16) { if (i == 0) { err = "repeat without previous length"; return false; } val = lengths[i - 1]; rep = 3 + (int)b.get(2); }
                        else if (sym == 17) rep = 3 + (int)b.get(3);
                        else rep = 11 + (int)b.get(7);
                        if (i + rep > hlit + hdist) { err = "code length repeat overflows"; return false; }
                        while (rep--) lengths[i++] = val;
                    }
                }
                if (!lit.build(lengths, hlit) || !dist.build(lengths + hlit, hdist)) { err = "over-subscribed Huffman code"; return false; }
            }
            for (;;) {
                int sym = lit.decode(b);
                if (sym < 0 || b.overrun) { err = "bad literal/length code or truncated stream"; return false; }
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) { err = "bad length symbol"; return false; }
                    uint32_t len = kLenBase[sym] + b.get(kLenExtra[sym]);
                    int ds = dist.decode(b);
                    if (ds < 0 || ds >= 30) { err = "bad distance symbol"; return false; }
                    uint32_t d = kDistBase[ds] + b.get(kDistExtra[ds]);
                    if (d > out.size()) { err = "distance beyond the window"; return false; }
                    size_t from = out.size() - d;
                    for (uint32_t k = 0; k < len; ++k) out.push_back(out[from + k]);
                }
            }
        } else { err = "reserved block type"; return false; }
        if (b.overrun) { err = "truncated deflate stream"; return false; }
    }
    b.align();
    uint32_t want = 0;
    for (int i = 0; i < 4; ++i) want = (want << 8) | b.get(8);
    if (b.overrun) { err = "missing adler32"; return false; }
    uint32_t s1 = 1, s2 = 0;
    for (uint8_t c : out) { s1 = (s1 + c) % 65521u; s2 = (s2 + s1) % 65521u; }
    if (((s2 << 16) | s1) != want) { err = "adler32 mismatch"; return false; }
    return true;
}

// ------------------------------------------------------------------ PNG
namespace {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) { int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p; return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

// reverses the scanline filters of one (sub)image in place; rows are `stride` bytes after the filter byte
bool unfilter(uint8_t* d, size_t rows, size_t stride, size_t bpp, std::string& err)
{
    for (size_t y = 0; y < rows; ++y) {
        uint8_t* cur = d + y * (stride + 1);
        const uint8_t* up = y ? cur - (stride + 1) + 1 : nullptr;
        uint8_t f = cur[0]; uint8_t* x = cur + 1;
        switch (f) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; ++i) x[i] = (uint8_t)(x[i] + x[i - bpp]); break;
        case 2: if (up) for (size_t i = 0; i < stride; ++i) x[i] = (uint8_t)(x[i] + up[i]); break;
        case 3: for (size_t i = 0; i < stride; ++i) { int a = i >= bpp ? x[i - bpp] : 0, b = up ? up[i] : 0; x[i] = (uint8_t)(x[i] + ((a + b) >> 1)); } break;
        case 4: for (size_t i = 0; i < stride; ++i) { int a = i >= bpp ? x[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0; x[i] = (uint8_t)(x[i] + paeth(a, b, c)); } break;
        default: err = "unknown PNG filter type"; return false;
        }
    }
    return true;
}
}

bool DecodePNG(const uint8_t* data, size_t n, Image& out, std::string& err)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (n < 8 || std::memcmp(data, sig, 8) != 0) { err = "not a PNG"; return false; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    size_t off = 8; bool sawEnd = false;
    while (off + 12 <= n && !sawEnd) {
        uint32_t len = be32(data + off); const uint8_t* type = data + off + 4; const uint8_t* body = data + off + 8;
        if (len > n - off - 12) { err = "PNG chunk overruns the file"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) { err = "unsupported PNG compression/filter/interlace method"; return false; }
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) sawEnd = true;
        off += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > 32768 || h > 32768) { err = "missing or bad IHDR"; return false; }
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: err = "bad PNG colour type"; return false; }
    bool depthOk = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                   ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depthOk) { err = "bad PNG bit depth for the colour type"; return false; }
    if (ctype == 3 && (palette.empty() || palette.size() % 3)) { err = "palette PNG without PLTE"; return false; }
    std::vector<uint8_t> raw;
    if (!Inflate(idat.data(), idat.size(), raw, err)) { err = "PNG IDAT: " + err; return false; }

    const size_t bitsPerPixel = (size_t)channels * depth, bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    // samples[y][x][c] as 16-bit values in file precision
    std::vector<uint16_t> samples((size_t)w * h * channels);
    auto unpack = [&](const uint8_t* rows, size_t pw, size_t ph, size_t x0, size_t y0, size_t dx, size_t dy) {
        size_t stride = (pw * bitsPerPixel + 7) / 8;
        for (size_t y = 0; y < ph; ++y) {
            const uint8_t* r = rows + y * (stride + 1) + 1;
            for (size_t x = 0; x < pw; ++x) for (int c = 0; c < channels; ++c) {
                size_t si = x * channels + c; uint16_t v;
                if (depth == 8) v = r[si];
                else if (depth == 16) v = (uint16_t)((r[2 * si] << 8) | r[2 * si + 1]);
                else { size_t bit = si * depth; v = (uint16_t)((r[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                samples[(((y0 + y * dy) * w) + (x0 + x * dx)) * channels + c] = v;
            }
        }
    };
    if (!interlace) {
        size_t stride = ((size_t)w * bitsPerPixel + 7) / 8;
        if (raw.size() < (stride + 1) * h) { err = "PNG image data too short"; return false; }
        if (!unfilter(raw.data(), h, stride, bpp, err)) return false;
        unpack(raw.data(), w, h, 0, 0, 1, 1);
    } else {
        static const int xs[7] = { 0, 4, 0, 2, 0, 1, 0 }, ys[7] = { 0, 0, 4, 0, 2, 0, 1 }, dxs[7] = { 8, 8, 4, 4, 2, 2, 1 }, dys[7] = { 8, 8, 8, 4, 4, 2, 2 };
        size_t pos = 0;
        for (int p = 0; p < 7; ++p) {
            size_t pw = (w > (uint32_t)xs[p]) ? (w - xs[p] + dxs[p] - 1) / dxs[p] : 0, ph = (h > (uint32_t)ys[p]) ? (h - ys[p] + dys[p] - 1) / dys[p] : 0;
            if (!pw || !ph) continue;
            size_t stride = (pw * bitsPerPixel + 7) / 8;
            if (raw.size() < pos + (stride + 1) * ph) { err = "PNG interlaced data too short"; return false; }
            if (!unfilter(raw.data() + pos, ph, stride, bpp, err)) return false;
            unpack(raw.data() + pos, pw, ph, xs[p], ys[p], dxs[p], dys[p]);
            pos += (stride + 1) * ph;
        }
    }
    // to RGBA8 with stb_image's conventions: 16-bit keeps the high byte, 1/2/4-bit grey is scaled to 0..255, tRNS colour key -> alpha 0
    out.width = w; out.height = h; out.rgba.assign((size_t)w * h * 4, 255);
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (uint8_t)v; };
    uint16_t key[3] = { 0, 0, 0 }; bool haveKey = false;
    if (ctype == 0 && trns.size() >= 2) { key[0] = (uint16_t)((trns[0] << 8) | trns[1]); haveKey = true; }
    if (ctype == 2 && trns.size() >= 6) { for (int c = 0; c < 3; ++c) key[c] = (uint16_t)((trns[2 * c] << 8) | trns[2 * c + 1]); haveKey = true; }
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint16_t* s = &samples[i * channels]; uint8_t* o = &out.rgba[i * 4];
        switch (ctype) {
        case 0: o[0] = o[1] = o[2] = (uint8_t)(depth < 8 ? s[0] * scale : to8(s[0])); if (haveKey && s[0] == key[0]) o[3] = 0; break;
        case 2: o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]); if (haveKey && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) o[3] = 0; break;
        case 3: {
            size_t pi = s[0];
            if (pi * 3 + 2 >= palette.size()) { err = "PNG palette index out of range"; return false; }
            o[0] = palette[pi * 3]; o[1] = palette[pi * 3 + 1]; o[2] = palette[pi * 3 + 2]; o[3] = pi < trns.size() ? trns[pi] : 255; break;
        }
        case 4: o[0] = o[1] = o[2] = to8(s[0]); o[3] = to8(s[1]); break;
        case 6: o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]); o[3] = to8(s[3]); break;
        }
    }
    return true;
}

// ------------------------------------------------------------------ DDS (header walk of src/TextureLoader.cpp:136-213; block decode per the D3D10+ BC specification)
namespace {
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void rgb565(uint16_t c, uint8_t* o) { uint32_t r = (c >> 11) & 31, g = (c >> 5) & 63, b = c & 31; o[0] = (uint8_t)((r << 3) | (r >> 2)); o[1] = (uint8_t)((g << 2) | (g >> 4)); o[2] = (uint8_t)((b << 3) | (b >> 2)); }
void bc1_colors(const uint8_t* blk, uint8_t pal[4][4], bool allowPunchThrough)
{
    uint16_t c0 = (uint16_t)(blk[0] | (blk[1] << 8)), c1 = (uint16_t)(blk[2] | (blk[3] << 8));
    rgb565(c0, pal[0]); rgb565(c1, pal[1]); pal[0][3] = pal[1][3] = 255;
    if (c0 > c1 || !allowPunchThrough) {
        for (int k = 0; k < 3; ++k) { pal[2][k] = (uint8_t)((2 * pal[0][k] + pal[1][k] + 1) / 3); pal[3][k] = (uint8_t)((pal[0][k] + 2 * pal[1][k] + 1) / 3); }
        pal[2][3] = pal[3][3] = 255;
    } else {
        for (int k = 0; k < 3; ++k) { pal[2][k] = (uint8_t)((pal[0][k] + pal[1][k]) / 2); pal[3][k] = 0; }
        pal[2][3] = 255; pal[3][3] = 0;
    }
}
void bc_alpha_block(const uint8_t* blk, uint8_t a[16])
{   // BC3 alpha / BC4 / one BC5 channel: two endpoints + 3-bit indices
    uint8_t e[8]; e[0] = blk[0]; e[1] = blk[1];
    if (e[0] > e[1]) for (int i = 1; i < 7; ++i) e[1 + i] = (uint8_t)(((7 - i) * e[0] + i * e[1] + 3) / 7);
    else { for (int i = 1; i < 5; ++i) e[1 + i] = (uint8_t)(((5 - i) * e[0] + i * e[1] + 2) / 5); e[6] = 0; e[7] = 255; }
    uint64_t bits = 0; for (int i = 0; i < 6; ++i) bits |= (uint64_t)blk[2 + i] << (8 * i);
    for (int i = 0; i < 16; ++i) a[i] = e[(bits >> (3 * i)) & 7];
}
}

bool DecodeDDS(const uint8_t* d, size_t n, Image& out, std::string& err)
{
    if (n < 128 || std::memcmp(d, "DDS ", 4) != 0 || le32(d + 4) != 124) { err = "not a DDS file"; return false; }
    uint32_t h = le32(d + 12), w = le32(d + 16), pfFlags = le32(d + 80), fourCC = le32(d + 84), bitCount = le32(d + 88);
    uint32_t rm = le32(d + 92), gm = le32(d + 96), bm = le32(d + 100), am = le32(d + 104);
    size_t off = 128; uint32_t dxgi = 0;
    const bool hasFourCC = (pfFlags & 0x4) != 0;
    if (hasFourCC && fourCC == 0x30315844u /* "DX10" */) { if (n < 148) { err = "DDS DX10 header truncated"; return false; } dxgi = le32(d + 128); off = 148; }
    if (w == 0 || h == 0 || w > 32768 || h > 32768) { err = "bad DDS dimensions"; return false; }
    enum { RGBA8, BGRA8, BC1, BC2, BC3, BC4, BC5 } fmt;
    if (dxgi) {
        switch (dxgi) { case 28: case 29: fmt = RGBA8; break; case 71: case 72: fmt = BC1; break; case 74: case 75: fmt = BC2; break; case 77: case 78: fmt = BC3; break;
                        case 80: fmt = BC4; break; case 83: fmt = BC5; break; default: err = "unsupported DXGI format " + std::to_string(dxgi) + " (BC6H/BC7 and float formats are not decoded on the host)"; return false; }
    } else if (hasFourCC) {
        if (fourCC == 0x31545844u) fmt = BC1; else if (fourCC == 0x33545844u) fmt = BC2; else if (fourCC == 0x35545844u) fmt = BC3;
        else if (fourCC == 0x31495441u) fmt = BC4; else if (fourCC == 0x32495441u) fmt = BC5; else { err = "unsupported DDS FourCC"; return false; }
    } else if ((pfFlags & 0x40) && bitCount == 32 && rm == 0x00ff0000u && gm == 0x0000ff00u && bm == 0x000000ffu && am == 0xff000000u) fmt = BGRA8;   // the reference labels this mask set RGBA8_UNORM (:118-121); the bytes in memory are B,G,R,A
    else if ((pfFlags & 0x40) && bitCount == 32 && rm == 0x000000ffu && gm == 0x0000ff00u && bm == 0x00ff0000u && am == 0xff000000u) fmt = RGBA8;
    else { err = "unsupported DDS pixel format"; return false; }
    out.width = w; out.height = h; out.rgba.assign((size_t)w * h * 4, 255);
    const uint8_t* p = d + off; size_t left = n - off;
    if (fmt == RGBA8 || fmt == BGRA8) {
        if (left < (size_t)w * h * 4) { err = "DDS pixel data truncated"; return false; }
        for (size_t i = 0; i < (size_t)w * h; ++i) { const uint8_t* s = p + 4 * i; uint8_t* o = &out.rgba[4 * i]; if (fmt == RGBA8) std::memcpy(o, s, 4); else { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = s[3]; } }
        return true;
    }
    const size_t bw = (w + 3) / 4, bh = (h + 3) / 4, blockBytes = (fmt == BC1 || fmt == BC4) ? 8 : 16;
    if (left < bw * bh * blockBytes) { err = "DDS block data truncated"; return false; }
    for (size_t by = 0; by < bh; ++by) for (size_t bx = 0; bx < bw; ++bx) {
        const uint8_t* blk = p + (by * bw + bx) * blockBytes;
        uint8_t px[16][4];
        if (fmt == BC1 || fmt == BC2 || fmt == BC3) {
            const uint8_t* cblk = fmt == BC1 ? blk : blk + 8;
            uint8_t pal[4][4]; bc1_colors(cblk, pal, fmt == BC1);
            uint32_t idx = le32(cblk + 4);
            for (int i = 0; i < 16; ++i) std::memcpy(px[i], pal[(idx >> (2 * i)) & 3], 4);
            if (fmt == BC2) for (int i = 0; i < 16; ++i) { uint32_t a4 = (blk[i >> 1] >> ((i & 1) * 4)) & 15; px[i][3] = (uint8_t)(a4 * 17); }
            if (fmt == BC3) { uint8_t a[16]; bc_alpha_block(blk, a); for (int i = 0; i < 16; ++i) px[i][3] = a[i]; }
        } else {
            uint8_t r[16], g[16]; bc_alpha_block(blk, r);
            if (fmt == BC5) bc_alpha_block(blk + 8, g);
            for (int i = 0; i < 16; ++i) { px[i][0] = r[i]; px[i][1] = fmt == BC5 ? g[i] : 0; px[i][2] = 0; px[i][3] = 255; }
        }
        for (int i = 0; i < 16; ++i) { size_t x = bx * 4 + (i & 3), y = by * 4 + (i >> 2); if (x < w && y < h) std::memcpy(&out.rgba[(y * w + x) * 4], px[i], 4); }
    }
    return true;
}

bool DecodeImage(const uint8_t* data, size_t n, Image& out, std::string& err)
{
    if (n >= 8 && data[0] == 0x89 && data[1] == 'P') return DecodePNG(data, n, out, err);
    if (n >= 4 && !std::memcmp(data, "DDS ", 4)) return DecodeDDS(data, n, out, err);
    if (n >= 3 && data[0] == 0xFF && data[1] == 0xD8) { err = "JPEG images are not decoded (convert to PNG or DDS)"; return false; }
    if (n >= 12 && !std::memcmp(data + 1, "KTX 20", 6)) { err = "KTX2 images are not decoded"; return false; }
    err = "unrecognised image format";
    return false;
}

bool LoadImageFile(const std::string& path, Image& out, std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> bytes; uint8_t buf[65536]; size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + k);
    std::fclose(f);
    if (!DecodeImage(bytes.data(), bytes.size(), out, err)) { err = path + ": " + err; return false; }
    return true;
}

} // namespace hobbyrt
