#include "ImageDecode.h"

#include <algorithm>
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

bool Inflate(const uint8_t* data, size_t n, std::vector<uint8_t>& out, std::string& err, size_t maxOut)
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
            if (out.size() + len > maxOut) { err = "more data than the container announced"; return false; }
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
                if (sym < 256) { if (out.size() >= maxOut) { err = "more data than the container announced"; return false; } out.push_back((uint8_t)sym); }
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
                    if (out.size() + len > maxOut) { err = "more data than the container announced"; return false; }
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
    const size_t bitsPerPixel = (size_t)channels * depth, bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    // The header bounds the size of the filtered image data ((row bytes + 1 filter byte) per row; the seven Adam7 passes add at most one byte of
    // rounding and one filter byte per pass row): the stream may not inflate past it (a few KB of IDAT can otherwise ask for gigabytes), and
    // nothing of the image's size is allocated before the data turned out to be there.
    const size_t rowBytes = ((size_t)w * bitsPerPixel + 7) / 8;
    const size_t maxRaw = (rowBytes + 1) * h + (interlace ? 2 * ((size_t)h + 7) * 7 : 0);
    std::vector<uint8_t> raw;
    if (!Inflate(idat.data(), idat.size(), raw, err, maxRaw)) { err = "PNG IDAT: " + err; return false; }
    if (raw.size() < rowBytes * (size_t)h / (interlace ? 2 : 1)) { err = "PNG image data too short"; return false; }

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

// ---- BC7 (D3D11 functional spec 19.5.x / the BPTC specification): 8 modes, 1-3 subsets, 64 partitions, optional p-bits, channel rotation
namespace {
const uint8_t kBc7Partition2[64][16] = {
    {0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1},{0,0,0,1,0,0,0,1,0,0,0,1,0,0,0,1},{0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1},{0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1},
    {0,0,0,0,0,0,0,1,0,0,0,1,0,0,1,1},{0,0,1,1,0,1,1,1,0,1,1,1,1,1,1,1},{0,0,0,1,0,0,1,1,0,1,1,1,1,1,1,1},{0,0,0,0,0,0,0,1,0,0,1,1,0,1,1,1},
    {0,0,0,0,0,0,0,0,0,0,0,1,0,0,1,1},{0,0,1,1,0,1,1,1,1,1,1,1,1,1,1,1},{0,0,0,0,0,0,0,1,0,1,1,1,1,1,1,1},{0,0,0,0,0,0,0,0,0,0,0,1,0,1,1,1},
    {0,0,0,1,0,1,1,1,1,1,1,1,1,1,1,1},{0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1},{0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1},{0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1},
    {0,0,0,0,1,0,0,0,1,1,1,0,1,1,1,1},{0,1,1,1,0,0,0,1,0,0,0,0,0,0,0,0},{0,0,0,0,0,0,0,0,1,0,0,0,1,1,1,0},{0,1,1,1,0,0,1,1,0,0,0,1,0,0,0,0},
    {0,0,1,1,0,0,0,1,0,0,0,0,0,0,0,0},{0,0,0,0,1,0,0,0,1,1,0,0,1,1,1,0},{0,0,0,0,0,0,0,0,1,0,0,0,1,1,0,0},{0,1,1,1,0,0,1,1,0,0,1,1,0,0,0,1},
    {0,0,1,1,0,0,0,1,0,0,0,1,0,0,0,0},{0,0,0,0,1,0,0,0,1,0,0,0,1,1,0,0},{0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0},{0,0,1,1,0,1,1,0,0,1,1,0,1,1,0,0},
    {0,0,0,1,0,1,1,1,1,1,1,0,1,0,0,0},{0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0},{0,1,1,1,0,0,0,1,1,0,0,0,1,1,1,0},{0,0,1,1,1,0,0,1,1,0,0,1,1,1,0,0},
    {0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1},{0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1},{0,1,0,1,1,0,1,0,0,1,0,1,1,0,1,0},{0,0,1,1,0,0,1,1,1,1,0,0,1,1,0,0},
    {0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,0},{0,1,0,1,0,1,0,1,1,0,1,0,1,0,1,0},{0,1,1,0,1,0,0,1,0,1,1,0,1,0,0,1},{0,1,0,1,1,0,1,0,1,0,1,0,0,1,0,1},
    {0,1,1,1,0,0,1,1,1,1,0,0,1,1,1,0},{0,0,0,1,0,0,1,1,1,1,0,0,1,0,0,0},{0,0,1,1,0,0,1,0,0,1,0,0,1,1,0,0},{0,0,1,1,1,0,1,1,1,1,0,1,1,1,0,0},
    {0,1,1,0,1,0,0,1,1,0,0,1,0,1,1,0},{0,0,1,1,1,1,0,0,1,1,0,0,0,0,1,1},{0,1,1,0,0,1,1,0,1,0,0,1,1,0,0,1},{0,0,0,0,0,1,1,0,0,1,1,0,0,0,0,0},
    {0,1,0,0,1,1,1,0,0,1,0,0,0,0,0,0},{0,0,1,0,0,1,1,1,0,0,1,0,0,0,0,0},{0,0,0,0,0,0,1,0,0,1,1,1,0,0,1,0},{0,0,0,0,0,1,0,0,1,1,1,0,0,1,0,0},
    {0,1,1,0,1,1,0,0,1,0,0,1,0,0,1,1},{0,0,1,1,0,1,1,0,1,1,0,0,1,0,0,1},{0,1,1,0,0,0,1,1,1,0,0,1,1,1,0,0},{0,0,1,1,1,0,0,1,1,1,0,0,0,1,1,0},
    {0,1,1,0,1,1,0,0,1,1,0,0,1,0,0,1},{0,1,1,0,0,0,1,1,0,0,1,1,1,0,0,1},{0,1,1,1,1,1,1,0,1,0,0,0,0,0,0,1},{0,0,0,1,1,0,0,0,1,1,1,0,0,1,1,1},
    {0,0,0,0,1,1,1,1,0,0,1,1,0,0,1,1},{0,0,1,1,0,0,1,1,1,1,1,1,0,0,0,0},{0,0,1,0,0,0,1,0,1,1,1,0,1,1,1,0},{0,1,0,0,0,1,0,0,0,1,1,1,0,1,1,1} };
const uint8_t kBc7Partition3[64][16] = {
    {0,0,1,1,0,0,1,1,0,2,2,1,2,2,2,2},{0,0,0,1,0,0,1,1,2,2,1,1,2,2,2,1},{0,0,0,0,2,0,0,1,2,2,1,1,2,2,1,1},{0,2,2,2,0,0,2,2,0,0,1,1,0,1,1,1},
    {0,0,0,0,0,0,0,0,1,1,2,2,1,1,2,2},{0,0,1,1,0,0,1,1,0,0,2,2,0,0,2,2},{0,0,2,2,0,0,2,2,1,1,1,1,1,1,1,1},{0,0,1,1,0,0,1,1,2,2,1,1,2,2,1,1},
    {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2},{0,0,0,0,1,1,1,1,1,1,1,1,2,2,2,2},{0,0,0,0,1,1,1,1,2,2,2,2,2,2,2,2},{0,0,1,2,0,0,1,2,0,0,1,2,0,0,1,2},
    {0,1,1,2,0,1,1,2,0,1,1,2,0,1,1,2},{0,1,2,2,0,1,2,2,0,1,2,2,0,1,2,2},{0,0,1,1,0,1,1,2,1,1,2,2,1,2,2,2},{0,0,1,1,2,0,0,1,2,2,0,0,2,2,2,0},
    {0,0,0,1,0,0,1,1,0,1,1,2,1,1,2,2},{0,1,1,1,0,0,1,1,2,0,0,1,2,2,0,0},{0,0,0,0,1,1,2,2,1,1,2,2,1,1,2,2},{0,0,2,2,0,0,2,2,0,0,2,2,1,1,1,1},
    {0,1,1,1,0,1,1,1,0,2,2,2,0,2,2,2},{0,0,0,1,0,0,0,1,2,2,2,1,2,2,2,1},{0,0,0,0,0,0,1,1,0,1,2,2,0,1,2,2},{0,0,0,0,1,1,0,0,2,2,1,0,2,2,1,0},
    {0,1,2,2,0,1,2,2,0,0,1,1,0,0,0,0},{0,0,1,2,0,0,1,2,1,1,2,2,2,2,2,2},{0,1,1,0,1,2,2,1,1,2,2,1,0,1,1,0},{0,0,0,0,0,1,1,0,1,2,2,1,1,2,2,1},
    {0,0,2,2,1,1,0,2,1,1,0,2,0,0,2,2},{0,1,1,0,0,1,1,0,2,0,0,2,2,2,2,2},{0,0,1,1,0,1,2,2,0,1,2,2,0,0,1,1},{0,0,0,0,2,0,0,0,2,2,1,1,2,2,2,1},
    {0,0,0,0,0,0,0,2,1,1,2,2,1,2,2,2},{0,2,2,2,0,0,2,2,0,0,1,2,0,0,1,1},{0,0,1,1,0,0,1,2,0,0,2,2,0,2,2,2},{0,1,2,0,0,1,2,0,0,1,2,0,0,1,2,0},
    {0,0,0,0,1,1,1,1,2,2,2,2,0,0,0,0},{0,1,2,0,1,2,0,1,2,0,1,2,0,1,2,0},{0,1,2,0,2,0,1,2,1,2,0,1,0,1,2,0},{0,0,1,1,2,2,0,0,1,1,2,2,0,0,1,1},
    {0,0,1,1,1,1,2,2,2,2,0,0,0,0,1,1},{0,1,0,1,0,1,0,1,2,2,2,2,2,2,2,2},{0,0,0,0,0,0,0,0,2,1,2,1,2,1,2,1},{0,0,2,2,1,1,2,2,0,0,2,2,1,1,2,2},
    {0,0,2,2,0,0,1,1,0,0,2,2,0,0,1,1},{0,2,2,0,1,2,2,1,0,2,2,0,1,2,2,1},{0,1,0,1,2,2,2,2,2,2,2,2,0,1,0,1},{0,0,0,0,2,1,2,1,2,1,2,1,2,1,2,1},
    {0,1,0,1,0,1,0,1,0,1,0,1,2,2,2,2},{0,2,2,2,0,1,1,1,0,2,2,2,0,1,1,1},{0,0,0,2,1,1,1,2,0,0,0,2,1,1,1,2},{0,0,0,0,2,1,1,2,2,1,1,2,2,1,1,2},
    {0,2,2,2,0,1,1,1,0,1,1,1,0,2,2,2},{0,0,0,2,1,1,1,2,1,1,1,2,0,0,0,2},{0,1,1,0,0,1,1,0,0,1,1,0,2,2,2,2},{0,0,0,0,0,0,0,0,2,1,1,2,2,1,1,2},
    {0,1,1,0,0,1,1,0,2,2,2,2,2,2,2,2},{0,0,2,2,0,0,1,1,0,0,1,1,0,0,2,2},{0,0,2,2,1,1,2,2,1,1,2,2,0,0,2,2},{0,0,0,0,0,0,0,0,0,0,0,0,2,1,1,2},
    {0,0,0,2,0,0,0,1,0,0,0,2,0,0,0,1},{0,2,2,2,1,2,2,2,0,2,2,2,1,2,2,2},{0,1,0,1,2,2,2,2,2,2,2,2,2,2,2,2},{0,1,1,1,2,0,1,1,2,2,0,1,2,2,2,0} };
const uint8_t kBc7Anchor2[64] = { 15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15, 15,2,8,2,2,8,8,15, 2,8,2,2,8,8,2,2, 15,15,6,8,2,8,15,15, 2,8,2,2,2,15,15,6, 6,2,6,8,15,15,2,2, 15,15,15,15,15,2,2,15 };
const uint8_t kBc7Anchor3a[64] = { 3,3,15,15,8,3,15,15, 8,8,6,6,6,5,3,3, 3,3,8,15,3,3,6,10, 5,8,8,6,8,5,15,15, 8,15,3,5,6,10,8,15, 15,3,15,5,15,15,15,15, 3,15,5,5,5,8,5,10, 5,10,8,13,15,12,3,3 };
const uint8_t kBc7Anchor3b[64] = { 15,8,8,3,15,15,3,8, 15,15,15,15,15,15,15,8, 15,8,15,3,15,8,15,8, 3,15,6,10,15,15,10,8, 15,3,15,10,10,8,9,10, 6,15,8,15,3,6,6,8, 15,3,15,15,15,15,15,15, 15,15,15,15,3,15,15,8 };
const uint8_t kBcWeights2[4] = { 0, 21, 43, 64 }, kBcWeights3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 }, kBcWeights4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
struct Bc7Mode { uint8_t ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; };
const Bc7Mode kBc7Modes[8] = { {3,4,0,0,4,0,1,0,3,0}, {2,6,0,0,6,0,0,1,3,0}, {3,6,0,0,5,0,0,0,2,0}, {2,6,0,0,7,0,1,0,2,0},
                               {1,0,2,1,5,6,0,0,2,3}, {1,0,2,0,7,8,0,0,2,2}, {1,0,0,0,7,7,1,0,4,0}, {2,6,0,0,5,5,1,0,2,0} };
struct BitReader {
    const uint8_t* p; uint32_t pos = 0;
    uint32_t get(uint32_t n) { uint32_t v = 0; for (uint32_t i = 0; i < n; ++i, ++pos) v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1u) << i; return v; }
};
inline const uint8_t* bc_weights(uint32_t bits) { return bits == 2 ? kBcWeights2 : (bits == 3 ? kBcWeights3 : kBcWeights4); }
void bc7_block(const uint8_t* blk, uint8_t px[16][4])
{
    uint32_t mode = 0; while (mode < 8 && !((blk[0] >> mode) & 1)) ++mode;
    if (mode == 8) { std::memset(px, 0, 64); return; }        // reserved mode: transparent black (spec)
    const Bc7Mode& m = kBc7Modes[mode];
    BitReader br{ blk, mode + 1 };
    const uint32_t partition = br.get(m.pb), rotation = br.get(m.rb), indexSel = br.get(m.isb);
    const uint32_t ne = m.ns * 2u;
    uint32_t e[6][4];                                         // endpoints [subset * 2 + {0, 1}][r, g, b, a]
    for (int c = 0; c < 3; ++c) for (uint32_t i = 0; i < ne; ++i) e[i][c] = br.get(m.cb);
    for (uint32_t i = 0; i < ne; ++i) e[i][3] = m.ab ? br.get(m.ab) : 255u;
    uint32_t cbits = m.cb, abits = m.ab;
    if (m.epb) { for (uint32_t i = 0; i < ne; ++i) { const uint32_t p = br.get(1); for (int c = 0; c < 3; ++c) e[i][c] = (e[i][c] << 1) | p; if (m.ab) e[i][3] = (e[i][3] << 1) | p; } ++cbits; if (m.ab) ++abits; }
    if (m.spb) { for (uint32_t sub = 0; sub < m.ns; ++sub) { const uint32_t p = br.get(1); for (uint32_t k = 0; k < 2; ++k) for (int c = 0; c < 3; ++c) e[sub * 2 + k][c] = (e[sub * 2 + k][c] << 1) | p; } ++cbits; }
    for (uint32_t i = 0; i < ne; ++i) {
        for (int c = 0; c < 3; ++c) { const uint32_t v = e[i][c] << (8 - cbits); e[i][c] = v | (v >> cbits); }
        if (m.ab) { const uint32_t v = e[i][3] << (8 - abits); e[i][3] = v | (v >> abits); }
    }
    uint8_t subsetOf[16];
    for (int i = 0; i < 16; ++i) subsetOf[i] = m.ns == 1 ? 0 : (m.ns == 2 ? kBc7Partition2[partition][i] : kBc7Partition3[partition][i]);
    uint32_t anchor[3] = { 0, 0, 0 };
    if (m.ns == 2) anchor[1] = kBc7Anchor2[partition];
    if (m.ns == 3) { anchor[1] = kBc7Anchor3a[partition]; anchor[2] = kBc7Anchor3b[partition]; }
    uint32_t idx1[16], idx2[16];
    for (uint32_t i = 0; i < 16; ++i) { const bool isAnchor = i == anchor[subsetOf[i]]; idx1[i] = br.get(isAnchor ? m.ib - 1u : m.ib); }
    for (uint32_t i = 0; i < 16; ++i) idx2[i] = m.ib2 ? br.get(i == 0 ? m.ib2 - 1u : m.ib2) : 0;
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t* e0 = e[subsetOf[i] * 2], * e1 = e[subsetOf[i] * 2 + 1];
        uint32_t ci = idx1[i], cbw = m.ib, ai = idx1[i], abw = m.ib;
        if (m.ib2) { if (indexSel) { ci = idx2[i]; cbw = m.ib2; } else { ai = idx2[i]; abw = m.ib2; } }
        const uint32_t wc = bc_weights(cbw)[ci], wa = bc_weights(abw)[ai];
        uint32_t c[4];
        for (int k = 0; k < 3; ++k) c[k] = ((64 - wc) * e0[k] + wc * e1[k] + 32) >> 6;
        c[3] = ((64 - wa) * e0[3] + wa * e1[3] + 32) >> 6;
        if (rotation == 1) std::swap(c[3], c[0]); else if (rotation == 2) std::swap(c[3], c[1]); else if (rotation == 3) std::swap(c[3], c[2]);
        for (int k = 0; k < 4; ++k) px[i][k] = (uint8_t)c[k];
    }
}
// ---- BC6H (D3D11 functional spec 19.5.x / BPTC float): 14 modes, 1-2 subsets (the first 32 two-subset partitions of BC7), endpoints stored with
// 6..16 bits and -- in the "transformed" modes -- as deltas against the first one, 3- or 4-bit indices; decodes to binary16 RGB.
// The header bit order of every mode as a string of fields: <endpoint letter r/g/b><endpoint 0..3>:<first bit>[-<last bit>] ("d" = partition);
// a range written high-to-low stores the most significant bit first (modes 13 and 14).
struct Bc6Mode { uint8_t transformed, subsets, wBits, dBits[3]; const char* layout; };
const Bc6Mode kBc6Modes[14] = {
    { 1, 2, 10, { 5, 5, 5 }, "g2:4 b2:4 b3:4 r0:0-9 g0:0-9 b0:0-9 r1:0-4 g3:4 g2:0-3 g1:0-4 b3:0 g3:0-3 b1:0-4 b3:1 b2:0-3 r2:0-4 b3:2 r3:0-4 b3:3 d:0-4" },
    { 1, 2, 7, { 6, 6, 6 }, "g2:5 g3:4 g3:5 r0:0-6 b3:0 b3:1 b2:4 g0:0-6 b2:5 b3:2 g2:4 b0:0-6 b3:3 b3:5 b3:4 r1:0-5 g2:0-3 g1:0-5 g3:0-3 b1:0-5 b2:0-3 r2:0-5 r3:0-5 d:0-4" },
    { 1, 2, 11, { 5, 4, 4 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-4 r0:10 g2:0-3 g1:0-3 g0:10 b3:0 g3:0-3 b1:0-3 b0:10 b3:1 b2:0-3 r2:0-4 b3:2 r3:0-4 b3:3 d:0-4" },
    { 1, 2, 11, { 4, 5, 4 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-3 r0:10 g3:4 g2:0-3 g1:0-4 g0:10 g3:0-3 b1:0-3 b0:10 b3:1 b2:0-3 r2:0-3 b3:0 b3:2 r3:0-3 g2:4 b3:3 d:0-4" },
    { 1, 2, 11, { 4, 4, 5 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-3 r0:10 b2:4 g2:0-3 g1:0-3 g0:10 b3:0 g3:0-3 b1:0-4 b0:10 b2:0-3 r2:0-3 b3:1 b3:2 r3:0-3 b3:4 b3:3 d:0-4" },
    { 1, 2, 9, { 5, 5, 5 }, "r0:0-8 b2:4 g0:0-8 g2:4 b0:0-8 b3:4 r1:0-4 g3:4 g2:0-3 g1:0-4 b3:0 g3:0-3 b1:0-4 b3:1 b2:0-3 r2:0-4 b3:2 r3:0-4 b3:3 d:0-4" },
    { 1, 2, 8, { 6, 5, 5 }, "r0:0-7 g3:4 b2:4 g0:0-7 b3:2 g2:4 b0:0-7 b3:3 b3:4 r1:0-5 g2:0-3 g1:0-4 b3:0 g3:0-3 b1:0-4 b3:1 b2:0-3 r2:0-5 r3:0-5 d:0-4" },
    { 1, 2, 8, { 5, 6, 5 }, "r0:0-7 b3:0 b2:4 g0:0-7 g2:5 g2:4 b0:0-7 g3:5 b3:4 r1:0-4 g3:4 g2:0-3 g1:0-5 g3:0-3 b1:0-4 b3:1 b2:0-3 r2:0-4 b3:2 r3:0-4 b3:3 d:0-4" },
    { 1, 2, 8, { 5, 5, 6 }, "r0:0-7 b3:1 b2:4 g0:0-7 b2:5 g2:4 b0:0-7 b3:5 b3:4 r1:0-4 g3:4 g2:0-3 g1:0-4 b3:0 g3:0-3 b1:0-5 b2:0-3 r2:0-4 b3:2 r3:0-4 b3:3 d:0-4" },
    { 0, 2, 6, { 6, 6, 6 }, "r0:0-5 g3:4 b3:0 b3:1 b2:4 g0:0-5 g2:5 b2:5 b3:2 g2:4 b0:0-5 g3:5 b3:3 b3:5 b3:4 r1:0-5 g2:0-3 g1:0-5 g3:0-3 b1:0-5 b2:0-3 r2:0-5 r3:0-5 d:0-4" },
    { 0, 1, 10, { 10, 10, 10 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-9 g1:0-9 b1:0-9" },
    { 1, 1, 11, { 9, 9, 9 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-8 r0:10 g1:0-8 g0:10 b1:0-8 b0:10" },
    { 1, 1, 12, { 8, 8, 8 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-7 r0:11-10 g1:0-7 g0:11-10 b1:0-7 b0:11-10" },
    { 1, 1, 16, { 4, 4, 4 }, "r0:0-9 g0:0-9 b0:0-9 r1:0-3 r0:15-10 g1:0-3 g0:15-10 b1:0-3 b0:15-10" },
};
// mode number (0-based index into kBc6Modes) from the low bits: two 2-bit codes, ten 5-bit codes; -1 = reserved
int bc6_mode(const uint8_t* blk, uint32_t& modeBits)
{
    const uint32_t lo2 = blk[0] & 3u, lo5 = blk[0] & 31u;
    if (lo2 == 0) { modeBits = 2; return 0; }
    if (lo2 == 1) { modeBits = 2; return 1; }
    modeBits = 5;
    switch (lo5) {
        case 2: return 2; case 6: return 3; case 10: return 4; case 14: return 5; case 18: return 6; case 22: return 7; case 26: return 8; case 30: return 9;
        case 3: return 10; case 7: return 11; case 11: return 12; case 15: return 13;
        default: return -1;
    }
}
inline int32_t bc6_sign_extend(uint32_t v, uint32_t bits) { const uint32_t m = 1u << (bits - 1); return (int32_t)((v ^ m) - m); }
int32_t bc6_unquantize(int32_t x, uint32_t bits, bool isSigned)
{
    if (!isSigned) {
        if (bits >= 15) return x;
        if (x == 0) return 0;
        if (x == (int32_t)((1u << bits) - 1u)) return 0xFFFF;
        return (int32_t)((((uint32_t)x << 15) + 0x4000u) >> (bits - 1));
    }
    if (bits >= 16) return x;
    const bool neg = x < 0; uint32_t a = (uint32_t)(neg ? -x : x), u;
    if (a == 0) u = 0; else if (a >= (1u << (bits - 1)) - 1u) u = 0x7FFF; else u = ((a << 15) + 0x4000u) >> (bits - 1);
    return neg ? -(int32_t)u : (int32_t)u;
}
uint16_t bc6_finish(int32_t c, bool isSigned)
{
    if (!isSigned) return (uint16_t)((c * 31) >> 6);
    return c < 0 ? (uint16_t)((((-c) * 31) >> 5) | 0x8000) : (uint16_t)((c * 31) >> 5);
}
void bc6h_block(const uint8_t* blk, bool isSigned, uint16_t px[16][4])
{
    uint32_t modeBits = 0;
    const int mi = bc6_mode(blk, modeBits);
    for (int i = 0; i < 16; ++i) { px[i][0] = px[i][1] = px[i][2] = 0; px[i][3] = 0x3C00u; }
    if (mi < 0) return;                                         // reserved mode: black (spec)
    const Bc6Mode& m = kBc6Modes[mi];
    BitReader br{ blk, modeBits };
    uint32_t e[4][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, partition = 0;
    for (const char* p = m.layout; *p;) {                       // walk the layout string
        while (*p == ' ') ++p;
        if (!*p) break;
        const char field = *p++;
        uint32_t* dst; 
        if (field == 'd') { dst = &partition; }
        else { const int ch = field == 'r' ? 0 : (field == 'g' ? 1 : 2); dst = &e[*p++ - '0'][ch]; }
        ++p;                                                    // ':'
        int first = 0; while (*p >= '0' && *p <= '9') first = first * 10 + (*p++ - '0');
        int last = first;
        if (*p == '-') { ++p; last = 0; while (*p >= '0' && *p <= '9') last = last * 10 + (*p++ - '0'); }
        const int step = last >= first ? 1 : -1;
        for (int b = first;; b += step) { *dst |= br.get(1) << b; if (b == last) break; }
    }
    const uint32_t ne = m.subsets * 2u;
    int32_t ep[4][3];
    for (int c = 0; c < 3; ++c) {
        int32_t base = (int32_t)e[0][c];
        if (isSigned) base = bc6_sign_extend(e[0][c], m.wBits);
        ep[0][c] = base;
        for (uint32_t i = 1; i < ne; ++i) {
            if (m.transformed) {
                const int32_t delta = bc6_sign_extend(e[i][c], m.dBits[c]);
                const uint32_t sum = ((uint32_t)e[0][c] + (uint32_t)delta) & ((1u << m.wBits) - 1u);
                ep[i][c] = isSigned ? bc6_sign_extend(sum, m.wBits) : (int32_t)sum;
            } else ep[i][c] = isSigned ? bc6_sign_extend(e[i][c], m.dBits[c]) : (int32_t)e[i][c];
        }
    }
    for (uint32_t i = 0; i < ne; ++i) for (int c = 0; c < 3; ++c) ep[i][c] = bc6_unquantize(ep[i][c], m.wBits, isSigned);
    const uint32_t ib = m.subsets == 2 ? 3u : 4u;
    const uint32_t anchor1 = m.subsets == 2 ? kBc7Anchor2[partition] : 0u;
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t sub = m.subsets == 2 ? kBc7Partition2[partition][i] : 0u;
        const bool isAnchor = i == 0 || (m.subsets == 2 && i == anchor1);
        const uint32_t idx = br.get(isAnchor ? ib - 1u : ib), w = bc_weights(ib)[idx];
        for (int c = 0; c < 3; ++c) {
            const int32_t v = (ep[sub * 2][c] * (int32_t)(64 - w) + ep[sub * 2 + 1][c] * (int32_t)w + 32) >> 6;
            px[i][c] = bc6_finish(v, isSigned);
        }
    }
}
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline void put_f32(std::vector<uint8_t>& v, size_t texel, const float rgba[4]) { std::memcpy(&v[texel * 16], rgba, 16); }
// one channel of a BC4_SNORM / BC5_SNORM block: signed endpoints, palette interpolated in float (D3D11 functional spec 19.5.7)
void bc_alpha_block_snorm(const uint8_t* blk, float a[16])
{
    const int r0 = (int8_t)blk[0], r1 = (int8_t)blk[1];
    const float f0 = r0 == -128 ? -1.0f : (float)r0 / 127.0f, f1 = r1 == -128 ? -1.0f : (float)r1 / 127.0f;
    float e[8]; e[0] = f0; e[1] = f1;
    if (r0 > r1) for (int i = 1; i < 7; ++i) e[1 + i] = ((float)(7 - i) * f0 + (float)i * f1) / 7.0f;
    else { for (int i = 1; i < 5; ++i) e[1 + i] = ((float)(5 - i) * f0 + (float)i * f1) / 5.0f; e[6] = -1.0f; e[7] = 1.0f; }
    uint64_t bits = 0; for (int i = 0; i < 6; ++i) bits |= (uint64_t)blk[2 + i] << (8 * i);
    for (int i = 0; i < 16; ++i) a[i] = e[(bits >> (3 * i)) & 7];
}
}
bool Bc7TablesConsistent()
{   // every anchor index must lie in the subset it anchors (a transcription check of the partition / anchor tables against each other)
    for (int p = 0; p < 64; ++p) {
        if (kBc7Partition2[p][0] != 0 || kBc7Partition2[p][kBc7Anchor2[p]] != 1) return false;
        if (kBc7Partition3[p][0] != 0 || kBc7Partition3[p][kBc7Anchor3a[p]] != 1 || kBc7Partition3[p][kBc7Anchor3b[p]] != 2) return false;
    }
    return true;
}

// DDS: the header walk and format map of src/TextureLoader.cpp:66-213 (GetFormatFromDDS); every level of the file is decoded.
bool DecodeDDS(const uint8_t* d, size_t n, Image& out, std::string& err)
{
    if (n < 128 || std::memcmp(d, "DDS ", 4) != 0 || le32(d + 4) != 124) { err = "not a DDS file"; return false; }
    uint32_t h = le32(d + 12), w = le32(d + 16), mipMapCount = le32(d + 28), pfFlags = le32(d + 80), fourCC = le32(d + 84), bitCount = le32(d + 88);
    uint32_t rm = le32(d + 92), gm = le32(d + 96), bm = le32(d + 100), am = le32(d + 104);
    size_t off = 128; uint32_t dxgi = 0;
    const bool hasFourCC = (pfFlags & 0x4) != 0;
    if (hasFourCC && fourCC == 0x30315844u /* "DX10" */) { if (n < 148) { err = "DDS DX10 header truncated"; return false; } dxgi = le32(d + 128); off = 148; }
    if (w == 0 || h == 0 || w > 32768 || h > 32768) { err = "bad DDS dimensions"; return false; }
    enum Fmt { RGBA8, BGRA8, BC1, BC2, BC3, BC4U, BC4S, BC5U, BC5S, BC6U, BC6S, BC7, R16F, RG16F, RGBA16F, R32F, RG32F, RGBA32F, RG16U, RGBA16U } fmt;
    bool srgb = false;
    if (dxgi) {
        switch (dxgi) {
            case 28: fmt = RGBA8; break; case 29: fmt = RGBA8; srgb = true; break; case 34: fmt = RG16F; break; case 35: fmt = RG16U; break;
            case 71: fmt = BC1; break; case 72: fmt = BC1; srgb = true; break; case 74: fmt = BC2; break; case 75: fmt = BC2; srgb = true; break;
            case 77: fmt = BC3; break; case 78: fmt = BC3; srgb = true; break; case 80: fmt = BC4U; break; case 81: fmt = BC4S; break;
            case 83: fmt = BC5U; break; case 84: fmt = BC5S; break; case 98: fmt = BC7; break; case 99: fmt = BC7; srgb = true; break;
            case 95: fmt = BC6U; break; case 96: fmt = BC6S; break;
            default: err = "unsupported DXGI format " + std::to_string(dxgi); return false;
        }
    } else if (hasFourCC) {
        if (fourCC == 0x31545844u) fmt = BC1; else if (fourCC == 0x33545844u) fmt = BC2; else if (fourCC == 0x35545844u) fmt = BC3;
        else if (fourCC == 0x31495441u) fmt = BC4U; else if (fourCC == 0x32495441u) fmt = BC5U;
        else if (fourCC == 34) fmt = RG16U; else if (fourCC == 36) fmt = RGBA16U; else if (fourCC == 111) fmt = R16F; else if (fourCC == 112) fmt = RG16F;
        else if (fourCC == 113) fmt = RGBA16F; else if (fourCC == 114) fmt = R32F; else if (fourCC == 115) fmt = RG32F; else if (fourCC == 116) fmt = RGBA32F;
        else { err = "unsupported DDS FourCC"; return false; }
    } else if ((pfFlags & 0x40) && bitCount == 32 && rm == 0x00ff0000u && gm == 0x0000ff00u && bm == 0x000000ffu && am == 0xff000000u) fmt = BGRA8;   // the reference labels this mask set RGBA8_UNORM (:118-121); the bytes in memory are B,G,R,A
    else if ((pfFlags & 0x40) && bitCount == 32 && rm == 0x000000ffu && gm == 0x0000ff00u && bm == 0x00ff0000u && am == 0xff000000u) fmt = RGBA8;
    else { err = "unsupported DDS pixel format"; return false; }
    uint32_t levels = mipMapCount ? mipMapCount : 1u;            // desc.mipLevels = dwMipMapCount ? dwMipMapCount : 1 (:202)
    uint32_t maxLevels = 1; while ((w >> maxLevels) || (h >> maxLevels)) ++maxLevels;
    if (levels > maxLevels || levels > 16u) { err = "DDS mip count exceeds the chain of a " + std::to_string(w) + "x" + std::to_string(h) + " texture"; return false; }
    const bool blockFmt = fmt >= BC1 && fmt <= BC7;
    const bool toF16 = fmt == R16F || fmt == RG16F || fmt == RGBA16F || fmt == BC6U || fmt == BC6S;
    const bool toF32 = fmt == R32F || fmt == RG32F || fmt == RGBA32F || fmt == RG16U || fmt == RGBA16U || fmt == BC4S || fmt == BC5S;
    const size_t texelBytes = toF32 ? 16 : (toF16 ? 8 : 4);
    out.format = toF32 ? 3u : (toF16 ? 2u : (srgb ? 1u : 0u));
    // payload size first: nothing is allocated for a truncated file
    size_t need = 0, totalTexels = 0;
    for (uint32_t l = 0; l < levels; ++l) {
        const size_t lw = std::max(1u, w >> l), lh = std::max(1u, h >> l);
        totalTexels += lw * lh;
        if (blockFmt) need += ((lw + 3) / 4) * ((lh + 3) / 4) * ((fmt == BC1 || fmt == BC4U || fmt == BC4S) ? 8 : 16);
        else { const size_t srcBytes = fmt == RGBA8 || fmt == BGRA8 ? 4 : (fmt == R16F ? 2 : (fmt == RG16F || fmt == R32F || fmt == RG16U ? 4 : (fmt == RGBA16F || fmt == RG32F || fmt == RGBA16U ? 8 : 16))); need += lw * lh * srcBytes; }
    }
    if (n - off < need) { err = "DDS pixel data truncated"; return false; }
    out.width = w; out.height = h; out.mipCount = levels; out.rgba.assign(totalTexels * texelBytes, 0);
    const uint8_t* p = d + off; size_t texelBase = 0;
    const uint16_t halfOne = 0x3C00u;
    for (uint32_t l = 0; l < levels; ++l) {
        const size_t lw = std::max(1u, w >> l), lh = std::max(1u, h >> l);
        if (!blockFmt) {
            for (size_t i = 0; i < lw * lh; ++i) {
                const size_t t = texelBase + i;
                if (fmt == RGBA8) { std::memcpy(&out.rgba[t * 4], p, 4); p += 4; }
                else if (fmt == BGRA8) { uint8_t* o = &out.rgba[t * 4]; o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = p[3]; p += 4; }
                else if (toF16) {
                    uint16_t v[4] = { 0, 0, 0, halfOne }; const int nc = fmt == R16F ? 1 : (fmt == RG16F ? 2 : 4);
                    for (int c = 0; c < nc; ++c) v[c] = le16(p + 2 * c);
                    std::memcpy(&out.rgba[t * 8], v, 8); p += 2 * nc;
                } else if (fmt == RG16U || fmt == RGBA16U) {
                    float v[4] = { 0.0f, 0.0f, 0.0f, 1.0f }; const int nc = fmt == RG16U ? 2 : 4;
                    for (int c = 0; c < nc; ++c) v[c] = (float)le16(p + 2 * c) / 65535.0f;
                    put_f32(out.rgba, t, v); p += 2 * nc;
                } else {
                    float v[4] = { 0.0f, 0.0f, 0.0f, 1.0f }; const int nc = fmt == R32F ? 1 : (fmt == RG32F ? 2 : 4);
                    std::memcpy(v, p, 4 * (size_t)nc); put_f32(out.rgba, t, v); p += 4 * nc;
                }
            }
        } else {
            const size_t bw = (lw + 3) / 4, bh = (lh + 3) / 4, blockBytes = (fmt == BC1 || fmt == BC4U || fmt == BC4S) ? 8 : 16;
            for (size_t by = 0; by < bh; ++by) for (size_t bx = 0; bx < bw; ++bx) {
                const uint8_t* blk = p + (by * bw + bx) * blockBytes;
                uint8_t px[16][4]; float pf[16][4]; uint16_t ph[16][4];
                if (fmt == BC1 || fmt == BC2 || fmt == BC3) {
                    const uint8_t* cblk = fmt == BC1 ? blk : blk + 8;
                    uint8_t pal[4][4]; bc1_colors(cblk, pal, fmt == BC1);
                    uint32_t idx = le32(cblk + 4);
                    for (int i = 0; i < 16; ++i) std::memcpy(px[i], pal[(idx >> (2 * i)) & 3], 4);
                    if (fmt == BC2) for (int i = 0; i < 16; ++i) { uint32_t a4 = (blk[i >> 1] >> ((i & 1) * 4)) & 15; px[i][3] = (uint8_t)(a4 * 17); }
                    if (fmt == BC3) { uint8_t a[16]; bc_alpha_block(blk, a); for (int i = 0; i < 16; ++i) px[i][3] = a[i]; }
                } else if (fmt == BC7) bc7_block(blk, px);
                else if (fmt == BC6U || fmt == BC6S) bc6h_block(blk, fmt == BC6S, ph);
                else if (fmt == BC4U || fmt == BC5U) {
                    uint8_t r[16], g[16]; bc_alpha_block(blk, r);
                    if (fmt == BC5U) bc_alpha_block(blk + 8, g);
                    for (int i = 0; i < 16; ++i) { px[i][0] = r[i]; px[i][1] = fmt == BC5U ? g[i] : 0; px[i][2] = 0; px[i][3] = 255; }
                } else {
                    float r[16], g[16]; bc_alpha_block_snorm(blk, r);
                    if (fmt == BC5S) bc_alpha_block_snorm(blk + 8, g);
                    for (int i = 0; i < 16; ++i) { pf[i][0] = r[i]; pf[i][1] = fmt == BC5S ? g[i] : 0.0f; pf[i][2] = 0.0f; pf[i][3] = 1.0f; }
                }
                for (int i = 0; i < 16; ++i) {
                    const size_t x = bx * 4 + (i & 3), y = by * 4 + (i >> 2);
                    if (x >= lw || y >= lh) continue;
                    if (toF32) put_f32(out.rgba, texelBase + y * lw + x, pf[i]);
                    else if (toF16) std::memcpy(&out.rgba[(texelBase + y * lw + x) * 8], ph[i], 8);
                    else std::memcpy(&out.rgba[(texelBase + y * lw + x) * 4], px[i], 4);
                }
            }
            p += bw * bh * blockBytes;
        }
        texelBase += lw * lh;
    }
    return true;
}


// ------------------------------------------------------------------ JPEG (baseline / extended sequential / progressive Huffman, 8 bit)
// Integer pipeline restated from the decoder the reference vendors and decodes through, /root/reference/external/stb_image.h (public
// domain; src/TextureLoader.cpp:225-257 calls stbi_load_from_memory(..., 4)): byte-identical texels need ITS arithmetic, so the 12-bit
// fixed-point inverse DCT below follows stb_image.h:2430-2465 (STBI__IDCT_1D, the jidctint family) term by term, the chroma upsampling its
// triangle filters for 2x1 / 1x2 / 2x2 (:3413-3540) and the colour conversion its 20-bit fixed-point YCbCr -> RGB (:3542-3570).
// Pinned byte for byte against that header compiled as test infrastructure: tests/test_decoders_vs_stb.py (oracle/_ref/libstb_ref.so).
// Progressive files keep the coefficients of every block over their scans (spectral selection + successive approximation, T.81 annex G)
// and are transformed at the end. Arithmetic-coded files, 12-bit samples and CMYK are reported as unsupported.
namespace {
const uint8_t kDezigzag[64 + 15] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                     29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };
struct JHuff { uint8_t size[257]; uint16_t code[257]; uint8_t value[256]; int maxcode[18]; int delta[17]; int count = 0; bool ok = false; };
bool jhuff_build(JHuff& h, const uint8_t* counts, const uint8_t* vals, int nvals)
{
    int k = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < counts[i]; ++j) { if (k >= 256) return false; h.size[k++] = (uint8_t)(i + 1); }
    if (k != nvals) return false;
    h.size[k] = 0; h.count = k;
    int code = 0; k = 0;
    for (int j = 1; j <= 16; ++j) {
        h.delta[j] = k - code;
        if (h.size[k] == j) { while (h.size[k] == j) h.code[k++] = (uint16_t)(code++); if (code - 1 >= (1 << j)) return false; }
        h.maxcode[j] = code << (16 - j);
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    std::memcpy(h.value, vals, (size_t)nvals);
    h.ok = true;
    return true;
}
struct JBits {
    const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int cnt = 0; int marker = -1; bool overrun = false;
    void grow()
    {
        while (cnt <= 24) {
            int b = 0;
            if (marker < 0 && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p : 0xD9;
                    while (c == 0xFF && p + 1 < end) { ++p; c = *p; }
                    if (c == 0) ++p;                    // stuffed zero
                    else { marker = c; ++p; b = 0; }   // a marker ends the entropy-coded segment: feed zeros from here on
                }
            } else if (marker < 0) overrun = true;
            acc |= (uint32_t)b << (24 - cnt); cnt += 8;
        }
    }
    int get(int n) { if (n == 0) return 0; if (cnt < n) grow(); uint32_t v = acc >> (32 - n); acc <<= n; cnt -= n; return (int)v; }
    int decode(const JHuff& h)
    {
        if (cnt < 16) grow();
        uint32_t top = acc >> 16; int len = 1;
        while (len <= 16 && (int)top >= h.maxcode[len]) ++len;
        if (len > 16) return -1;
        int idx = (int)(acc >> (32 - len)) + h.delta[len];
        if (idx < 0 || idx >= h.count) return -1;
        acc <<= len; cnt -= len;
        return h.value[idx];
    }
    static int extend(int v, int bits) { return v < (1 << (bits - 1)) ? v - (1 << bits) + 1 : v; }
    void reset() { acc = 0; cnt = 0; marker = -1; }
};
inline uint8_t jclamp(long long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
// 64-bit temporaries: identical to the 32-bit original on every valid stream, and no signed overflow on corrupted coefficients
#define JF2F(x) ((long long)(((x) * 4096 + 0.5)))
#define JIDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                        \
    long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                        \
    p2 = s2; p3 = s6; p1 = (p2 + p3) * JF2F(0.5411961f);                                                 \
    t2 = p1 + p3 * JF2F(-1.847759065f); t3 = p1 + p2 * JF2F(0.765366865f);                               \
    p2 = s0; p3 = s4; t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                      \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                              \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                  \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2; p5 = (p3 + p4) * JF2F(1.175875602f);         \
    t0 = t0 * JF2F(0.298631336f); t1 = t1 * JF2F(2.053119869f); t2 = t2 * JF2F(3.072711026f); t3 = t3 * JF2F(1.501321110f); \
    p1 = p5 + p1 * JF2F(-0.899976223f); p2 = p5 + p2 * JF2F(-2.562915447f);                              \
    p3 = p3 * JF2F(-1.961570560f); p4 = p4 * JF2F(-0.390180644f);                                        \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
void jidct_block(uint8_t* out, size_t stride, const short d[64])
{
    long long val[64];
    for (int i = 0; i < 8; ++i) {
        const short* c = d + i; long long* v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            long long dc = c[0] * 4; v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            JIDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {
        const long long* v = val + 8 * i; uint8_t* o = out + stride * (size_t)i;
        JIDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        const long long bias = 65536 + (128 << 17);
        x0 += bias; x1 += bias; x2 += bias; x3 += bias;
        o[0] = jclamp((x0 + t3) >> 17); o[7] = jclamp((x0 - t3) >> 17); o[1] = jclamp((x1 + t2) >> 17); o[6] = jclamp((x1 - t2) >> 17);
        o[2] = jclamp((x2 + t1) >> 17); o[5] = jclamp((x2 - t1) >> 17); o[3] = jclamp((x3 + t0) >> 17); o[4] = jclamp((x3 - t0) >> 17);
    }
}
struct JComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, dcPred = 0; int x = 0, y = 0, w2 = 0, h2 = 0; std::vector<uint8_t> data;
               std::vector<short> coeff; };      // progressive: the coefficients of every block (64 each, natural order), refined scan by scan
// one output row of a component at full resolution
void jresample(const JComp& c, int hs, int vs, const uint8_t* nearRow, const uint8_t* farRow, int wLores, uint8_t* out)
{
    if (hs == 1 && vs == 1) { std::memcpy(out, nearRow, (size_t)wLores); return; }
    if (hs == 1 && vs == 2) { for (int i = 0; i < wLores; ++i) out[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2); return; }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = nearRow; int w = wLores;
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) { int n = 3 * in[i] + 2; out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2); }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2); out[i * 2 + 1] = in[w - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        int w = wLores;
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2); return; }
        int t1 = 3 * nearRow[0] + farRow[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) { int t0 = t1; t1 = 3 * nearRow[i] + farRow[i]; out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4); }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return;
    }
    for (int i = 0; i < wLores; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];     // other ratios: nearest
    (void)c;
}
} // namespace

bool DecodeJPEG(const uint8_t* data, size_t n, Image& out, std::string& err)
{
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) { err = "not a JPEG"; return false; }
    uint16_t quant[4][64]; bool haveQuant[4] = { false, false, false, false };
    JHuff dcTab[4], acTab[4];
    std::vector<JComp> comps; int width = 0, height = 0, hmax = 1, vmax = 1, restart = 0, adobeTransform = -1; bool sawSOF = false, decoded = false, progressive = false;
    size_t pos = 2;
    auto u16 = [&](size_t o) { return (int)((data[o] << 8) | data[o + 1]); };
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) { err = "JPEG marker expected"; return false; }
        int m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) { err = "JPEG truncated"; return false; }
        int len = u16(pos);
        if (len < 2 || pos + (size_t)len > n) { err = "JPEG segment overruns the file"; return false; }
        const uint8_t* seg = data + pos + 2; int sl = len - 2;
        if (m == 0xDB) {
            while (sl > 0) {
                int pq = seg[0] >> 4, tq = seg[0] & 15;
                if (tq > 3 || pq > 1 || sl < 1 + 64 * (pq + 1)) { err = "bad DQT"; return false; }
                for (int i = 0; i < 64; ++i) quant[tq][kDezigzag[i]] = (uint16_t)(pq ? ((seg[1 + 2 * i] << 8) | seg[2 + 2 * i]) : seg[1 + i]);
                haveQuant[tq] = true;
                seg += 1 + 64 * (pq + 1); sl -= 1 + 64 * (pq + 1);
            }
        } else if (m == 0xC4) {
            while (sl > 0) {
                if (sl < 17) { err = "bad DHT"; return false; }
                int tc = seg[0] >> 4, th = seg[0] & 15, total = 0;
                for (int i = 0; i < 16; ++i) total += seg[1 + i];
                if (tc > 1 || th > 3 || total > 256 || sl < 17 + total) { err = "bad DHT"; return false; }
                if (!jhuff_build(tc ? acTab[th] : dcTab[th], seg + 1, seg + 17, total)) { err = "bad Huffman table"; return false; }
                seg += 17 + total; sl -= 17 + total;
            }
        } else if (m == 0xDD) { if (sl < 2) { err = "bad DRI"; return false; } restart = (seg[0] << 8) | seg[1]; }
        else if (m == 0xEE && sl >= 12 && !std::memcmp(seg, "Adobe", 5)) adobeTransform = seg[11];
        else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (sawSOF) { err = "JPEG with several frames"; return false; }
            progressive = m == 0xC2;
            if (sl < 6 || seg[0] != 8) { err = "only 8-bit JPEG samples are supported"; return false; }
            height = (seg[1] << 8) | seg[2]; width = (seg[3] << 8) | seg[4]; int nc = seg[5];
            if (width <= 0 || height <= 0 || width > 32768 || height > 32768) { err = "bad JPEG dimensions"; return false; }
            if (nc != 1 && nc != 3) { err = "JPEG with " + std::to_string(nc) + " components is not supported (CMYK / YCCK)"; return false; }
            if (sl < 6 + 3 * nc) { err = "bad SOF"; return false; }
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; ++i) {
                JComp& c = comps[(size_t)i]; c.id = seg[6 + 3 * i]; c.h = seg[7 + 3 * i] >> 4; c.v = seg[7 + 3 * i] & 15; c.tq = seg[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { err = "bad JPEG sampling factors"; return false; }
                hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
            }
            for (JComp& c : comps) if (hmax % c.h || vmax % c.v) { err = "unsupported JPEG sampling ratio"; return false; }
            int mcuX = (width + 8 * hmax - 1) / (8 * hmax), mcuY = (height + 8 * vmax - 1) / (8 * vmax);
            {   // the planes are sized from this header alone: refuse a frame whose entropy-coded data cannot exist in the bytes that are left
                // (every 8 x 8 block costs at least one DC code bit in its first scan), before allocating anything for it
                size_t blocks = 0; for (const JComp& c : comps) blocks += (size_t)mcuX * c.h * (size_t)mcuY * c.v;
                if (blocks / 8 > n - pos) { err = "JPEG frame of " + std::to_string(width) + " x " + std::to_string(height) + " announced by a file of " + std::to_string(n) + " bytes"; return false; }
            }
            for (JComp& c : comps) {
                c.x = (width * c.h + hmax - 1) / hmax; c.y = (height * c.v + vmax - 1) / vmax; c.w2 = mcuX * c.h * 8; c.h2 = mcuY * c.v * 8;
                c.data.assign((size_t)c.w2 * c.h2 + 15, 0);
                if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
            sawSOF = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) { err = "unsupported JPEG coding process (lossless, hierarchical or arithmetic)"; return false; }
        else if (m == 0xDA) {
            if (!sawSOF || sl < 1) { err = "SOS before SOF"; return false; }
            int ns = seg[0];
            if (ns < 1 || ns > (int)comps.size() || sl < 1 + 2 * ns + 3) { err = "bad SOS"; return false; }
            std::vector<JComp*> scan;
            for (int i = 0; i < ns; ++i) {
                JComp* c = nullptr; for (JComp& k : comps) if (k.id == seg[1 + 2 * i]) c = &k;
                if (!c) { err = "SOS names an unknown component"; return false; }
                c->td = seg[2 + 2 * i] >> 4; c->ta = seg[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !haveQuant[c->tq]) { err = "scan uses an undefined table"; return false; }
                scan.push_back(c);
            }
            const int specStart = seg[1 + 2 * ns], specEnd = seg[2 + 2 * ns], succHigh = seg[3 + 2 * ns] >> 4, succLow = seg[3 + 2 * ns] & 15;
            if (!progressive) {
                if (specStart != 0 || specEnd != 63) { err = "spectral selection in a sequential JPEG"; return false; }
                for (JComp* c : scan) if (!dcTab[c->td].ok || !acTab[c->ta].ok) { err = "scan uses an undefined table"; return false; }
            } else {
                if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13 || (specStart == 0 && specEnd != 0) || (specStart > 0 && ns != 1)) { err = "bad progressive scan parameters"; return false; }
                for (JComp* c : scan) if (specStart == 0 ? (succHigh == 0 && !dcTab[c->td].ok) : !acTab[c->ta].ok) { err = "scan uses an undefined table"; return false; }
            }
            JBits bits{ data + pos + (size_t)len, data + n };
            for (JComp& c : comps) c.dcPred = 0;
            int eobRun = 0;
            // one block of a progressive scan (ITU T.81 annex G; the decode order of stb_image's stbi__jpeg_decode_block_prog_dc / _ac)
            auto block_prog = [&](JComp& c, int bx, int by) -> bool {
                short* dat = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (size_t)(c.w2 >> 3));
                if (specStart == 0) {
                    if (succHigh == 0) {
                        int t = bits.decode(dcTab[c.td]);
                        if (t < 0 || t > 15) return false;
                        int diff = t ? JBits::extend(bits.get(t), t) : 0;
                        c.dcPred += diff; dat[0] = (short)(c.dcPred * (1 << succLow));
                    } else if (bits.get(1)) dat[0] = (short)(dat[0] + (short)(1 << succLow));
                    return !bits.overrun;
                }
                if (succHigh == 0) {
                    if (eobRun) { --eobRun; return true; }
                    int k = specStart;
                    do {
                        int rs = bits.decode(acTab[c.ta]);
                        if (rs < 0) return false;
                        int sz = rs & 15, r = rs >> 4;
                        if (sz == 0) {
                            if (r < 15) { eobRun = 1 << r; if (r) eobRun += bits.get(r); --eobRun; break; }
                            k += 16;
                        } else { k += r; if (k > 63) return false; int zig = kDezigzag[k++]; dat[zig] = (short)(JBits::extend(bits.get(sz), sz) * (1 << succLow)); }
                    } while (k <= specEnd);
                } else {
                    const short bit = (short)(1 << succLow);
                    auto refine = [&](short& p) { if (bits.get(1) && (p & bit) == 0) p = (short)(p > 0 ? p + bit : p - bit); };
                    if (eobRun) {
                        --eobRun;
                        for (int k = specStart; k <= specEnd; ++k) { short& p = dat[kDezigzag[k]]; if (p != 0) refine(p); }
                    } else {
                        int k = specStart;
                        do {
                            int rs = bits.decode(acTab[c.ta]);
                            if (rs < 0) return false;
                            int sz = rs & 15, r = rs >> 4; short nv = 0;
                            if (sz == 0) {
                                if (r < 15) { eobRun = (1 << r) - 1; if (r) eobRun += bits.get(r); r = 64; }   // r = 64: run to the end of the band
                            } else { if (sz != 1) return false; nv = bits.get(1) ? bit : (short)-bit; }
                            while (k <= specEnd) {
                                short& p = dat[kDezigzag[k++]];
                                if (p != 0) refine(p);
                                else { if (r == 0) { p = nv; break; } --r; }
                            }
                        } while (k <= specEnd);
                    }
                }
                return !bits.overrun;
            };
            auto block = [&](JComp& c, int bx, int by) -> bool {
                short coef[64]; std::memset(coef, 0, sizeof coef);
                int t = bits.decode(dcTab[c.td]);
                if (t < 0 || t > 15) return false;
                int diff = t ? JBits::extend(bits.get(t), t) : 0;
                c.dcPred += diff; coef[0] = (short)(c.dcPred * quant[c.tq][0]);
                for (int k = 1; k < 64;) {
                    int rs = bits.decode(acTab[c.ta]);
                    if (rs < 0) return false;
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) { if (rs != 0xF0) break; k += 16; }
                    else { k += r; int zig = kDezigzag[k++]; coef[zig] = (short)(JBits::extend(bits.get(s), s) * quant[c.tq][zig]); }
                }
                jidct_block(c.data.data() + (size_t)c.w2 * (size_t)by * 8 + (size_t)bx * 8, (size_t)c.w2, coef);
                return !bits.overrun;
            };
            int todo = restart ? restart : 0x7fffffff;
            auto after_mcu = [&]() -> bool {
                if (--todo <= 0) {
                    if (bits.cnt < 24) bits.grow();
                    if (bits.marker < 0xD0 || bits.marker > 0xD7) return true;      // no restart marker here: the scan data ended (or is damaged); decode what is left as zeros
                    bits.reset(); for (JComp& c : comps) c.dcPred = 0; eobRun = 0; todo = restart;
                }
                return true;
            };
            bool ok = true;
            auto one_block = [&](JComp& c, int bx, int by) -> bool { return progressive ? block_prog(c, bx, by) : block(c, bx, by); };
            if (ns == 1) {
                JComp& c = *scan[0]; int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h && ok; ++j) for (int i = 0; i < w && ok; ++i) { ok = one_block(c, i, j) && after_mcu(); }
            } else {
                int mcuX = (width + 8 * hmax - 1) / (8 * hmax), mcuY = (height + 8 * vmax - 1) / (8 * vmax);
            {   // the planes are sized from this header alone: refuse a frame whose entropy-coded data cannot exist in the bytes that are left
                // (every 8 x 8 block costs at least one DC code bit in its first scan), before allocating anything for it
                size_t blocks = 0; for (const JComp& c : comps) blocks += (size_t)mcuX * c.h * (size_t)mcuY * c.v;
                if (blocks / 8 > n - pos) { err = "JPEG frame of " + std::to_string(width) + " x " + std::to_string(height) + " announced by a file of " + std::to_string(n) + " bytes"; return false; }
            }
                for (int j = 0; j < mcuY && ok; ++j) for (int i = 0; i < mcuX && ok; ++i) {
                    for (JComp* c : scan) for (int y = 0; y < c->v && ok; ++y) for (int x = 0; x < c->h && ok; ++x) ok = one_block(*c, i * c->h + x, j * c->v + y);
                    ok = ok && after_mcu();
                }
            }
            if (!ok) { err = "corrupt JPEG entropy-coded data"; return false; }
            decoded = true;
            // continue after the entropy-coded segment: at the marker the bit reader stopped on, or scan forward for one
            size_t q = (size_t)(bits.p - data);
            if (bits.marker >= 0) { pos = q - 2; continue; }
            while (q + 1 < n && !(data[q] == 0xFF && data[q + 1] != 0 && !(data[q + 1] >= 0xD0 && data[q + 1] <= 0xD7))) ++q;
            pos = q; continue;
        }
        pos += (size_t)len;
    }
    if (!sawSOF || !decoded) { err = "JPEG without image data"; return false; }
    if (progressive) {      // all scans are in: dequantise and transform every block (stbi__jpeg_finish)
        for (JComp& c : comps) {
            if (!haveQuant[c.tq]) { err = "scan uses an undefined table"; return false; }
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
                short* dat = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (size_t)(c.w2 >> 3));
                for (int k = 0; k < 64; ++k) dat[k] = (short)(dat[k] * quant[c.tq][k]);
                jidct_block(c.data.data() + (size_t)c.w2 * (size_t)j * 8 + (size_t)i * 8, (size_t)c.w2, dat);
            }
        }
    }
    out.width = (uint32_t)width; out.height = (uint32_t)height; out.rgba.assign((size_t)width * height * 4, 255);
    const size_t nc = comps.size();
    std::vector<std::vector<uint8_t>> line(nc, std::vector<uint8_t>((size_t)width + 8 * 4 + 16));
    struct Res { int hs, vs, ystep, wLores, ypos; const uint8_t* line0; const uint8_t* line1; };
    std::vector<Res> res(nc);
    for (size_t k = 0; k < nc; ++k) { const JComp& c = comps[k]; res[k] = { hmax / c.h, vmax / c.v, (vmax / c.v) >> 1, (width + hmax / c.h - 1) / (hmax / c.h), 0, c.data.data(), c.data.data() }; }
    for (int j = 0; j < height; ++j) {
        for (size_t k = 0; k < nc; ++k) {
            Res& r = res[k]; const JComp& c = comps[k];
            bool bot = r.ystep >= (r.vs >> 1);
            jresample(c, r.hs, r.vs, bot ? r.line1 : r.line0, bot ? r.line0 : r.line1, r.wLores, line[k].data());
            if (++r.ystep >= r.vs) { r.ystep = 0; r.line0 = r.line1; if (++r.ypos < c.y) r.line1 += c.w2; }
        }
        uint8_t* o = &out.rgba[(size_t)j * width * 4];
        if (nc == 1) for (int i = 0; i < width; ++i) { o[4 * i] = o[4 * i + 1] = o[4 * i + 2] = line[0][(size_t)i]; }
        else if (adobeTransform == 0) for (int i = 0; i < width; ++i) { o[4 * i] = line[0][(size_t)i]; o[4 * i + 1] = line[1][(size_t)i]; o[4 * i + 2] = line[2][(size_t)i]; }
        else for (int i = 0; i < width; ++i) {
            #define JFIX(x) (((int)((x) * 4096.0f + 0.5f)) << 8)
            int yf = (line[0][(size_t)i] << 20) + (1 << 19), cr = line[2][(size_t)i] - 128, cb = line[1][(size_t)i] - 128;
            int r = yf + cr * JFIX(1.40200f);
            int g = yf + (cr * -JFIX(0.71414f)) + (int)(((unsigned)(cb * -JFIX(0.34414f))) & 0xffff0000u);
            int b = yf + cb * JFIX(1.77200f);
            o[4 * i] = jclamp(r >> 20); o[4 * i + 1] = jclamp(g >> 20); o[4 * i + 2] = jclamp(b >> 20);
            #undef JFIX
        }
    }
    return true;
}

bool DecodeImage(const uint8_t* data, size_t n, Image& out, std::string& err)
{
    if (n >= 8 && data[0] == 0x89 && data[1] == 'P') return DecodePNG(data, n, out, err);
    if (n >= 4 && !std::memcmp(data, "DDS ", 4)) return DecodeDDS(data, n, out, err);
    if (n >= 3 && data[0] == 0xFF && data[1] == 0xD8) return DecodeJPEG(data, n, out, err);
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
