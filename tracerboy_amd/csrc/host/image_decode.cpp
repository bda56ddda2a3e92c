/* image_decode.cpp -- PNG and TGA decoders for image textures.
 *
 * The reference hands these files to DirectXTex (TracerBoy.cpp:2188-2232: LoadFromTGAFile / LoadFromWICFile); what reaches
 * the shaders is the DXGI format's typed load, which the build reproduces as RGBA32F texels:
 *   8-bit channels  -> v / 255          16-bit channels -> v / 65535            (UNORM: IsNormalizedFormat -> gamma flag)
 *   grey without alpha -> (g, 0, 0, 1)  (WIC 8bppGray -> DXGI_FORMAT_R8_UNORM / R16_UNORM: the typed load fills g,b with 0)
 *   grey + alpha, palette, RGB, RGBA -> R8G8B8A8 / R16G16B16A16 with alpha 1 where the file has none
 * hasAlpha <-> !ScratchImage::IsAlphaAllOpaque() (drives NO_ALPHA_MATERIAL_FLAG / the any-hit geometry flag).
 * No third-party code: inflate (RFC 1950/1951), PNG unfiltering (all five filters, bit depths 1-16, colour types
 * 0/2/3/4/6, Adam7 interlace) and TGA types 1/2/3/9/10/11 are written out here. */
#include "host_scene.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include <sys/stat.h>

namespace tbhost {
namespace {

bool readFile(const std::string& file, std::vector<uint8_t>& out, std::string& err)
{
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) { err = "cannot open image '" + file + "'"; return false; }
    struct stat st;
    /* fopen succeeds on a directory and ftell then reports LONG_MAX */
    if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode)) { fclose(f); err = "'" + file + "' is not a regular file"; return false; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    bool ok = out.empty() || fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short read from '" + file + "'";
    return ok;
}

/* ---- inflate ------------------------------------------------------------------------------------------------------ */
struct BitReader {
    const uint8_t* p; size_t n, at = 0; uint32_t acc = 0; int bits = 0;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    uint32_t get(int k)
    {
        while (bits < k) { if (at >= n) throw std::runtime_error("inflate: out of data"); acc |= (uint32_t)p[at++] << bits; bits += 8; }
        uint32_t v = k ? acc & ((1u << k) - 1) : 0; acc = k >= 32 ? 0 : acc >> k; bits -= k; return v;
    }
    void alignByte() { acc = 0; bits = 0; }
};

struct Huffman {
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const uint8_t* lengths, int n)
    {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)br.get(1);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw std::runtime_error("inflate: bad Huffman code");
    }
};

void inflateRaw(BitReader& br, std::vector<uint8_t>& out)
{
    static const uint16_t lenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t distBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
        8193, 12289, 16385, 24577};
    static const uint8_t distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const uint32_t final = br.get(1), type = br.get(2);
        if (type == 0) {
            br.alignByte();
            if (br.at + 4 > br.n) throw std::runtime_error("inflate: truncated stored block");
            uint32_t len = br.p[br.at] | (br.p[br.at + 1] << 8), nlen = br.p[br.at + 2] | (br.p[br.at + 3] << 8);
            br.at += 4;
            if ((len ^ 0xffffu) != nlen || br.at + len > br.n) throw std::runtime_error("inflate: bad stored block");
            out.insert(out.end(), br.p + br.at, br.p + br.at + len); br.at += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lengths[i] = 8; for (int i = 144; i < 256; i++) lengths[i] = 9;
                for (int i = 256; i < 280; i++) lengths[i] = 7; for (int i = 280; i < 288; i++) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; i++) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
                if (nlen > 286 || ndist > 30) throw std::runtime_error("inflate: bad dynamic header");
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.get(3);
                Huffman clh; clh.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    int sym = clh.decode(br);
                    if (sym < 16) lengths[i++] = (uint8_t)sym;
                    else {
                        uint8_t prev = 0; int rep;
                        if (sym == 16) { if (i == 0) throw std::runtime_error("inflate: repeat without previous length"); prev = lengths[i - 1];
                            rep = 3 + (int)br.get(2); }
                        else if (sym == 17) rep = 3 + (int)br.get(3);
                        else rep = 11 + (int)br.get(7);
                        if (i + rep > nlen + ndist) throw std::runtime_error("inflate: repeat overruns the code lengths");
                        while (rep--) lengths[i++] = prev;
                    }
                }
                lit.build(lengths, nlen); dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) throw std::runtime_error("inflate: bad length symbol");
                    const uint32_t len = lenBase[sym] + br.get(lenExtra[sym]);
                    const int ds = dist.decode(br);
                    if (ds >= 30) throw std::runtime_error("inflate: bad distance symbol");
                    const uint32_t d = distBase[ds] + br.get(distExtra[ds]);
                    if (d > out.size()) throw std::runtime_error("inflate: distance beyond the start of the output");
                    const size_t from = out.size() - d;
                    for (uint32_t k = 0; k < len; k++) out.push_back(out[from + k]);
                }
            }
        } else throw std::runtime_error("inflate: reserved block type");
        if (final) break;
    }
}

void zlibInflate(const std::vector<uint8_t>& z, std::vector<uint8_t>& out)
{
    if (z.size() < 6 || (z[0] & 0x0f) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) throw std::runtime_error("zlib: bad stream header");
    BitReader br(z.data() + 2, z.size() - 2);
    inflateRaw(br, out);
    uint32_t a = 1, b = 0;
    for (uint8_t v : out) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
    br.alignByte();
    if (br.at + 4 <= br.n) {
        uint32_t want = ((uint32_t)br.p[br.at] << 24) | ((uint32_t)br.p[br.at + 1] << 16) | ((uint32_t)br.p[br.at + 2] << 8) | br.p[br.at + 3];
        if (want != ((b << 16) | a)) throw std::runtime_error("zlib: Adler-32 mismatch");
    }
}

/* ---- PNG ---------------------------------------------------------------------------------------------------------- */
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

/* unfilters `rows` scanlines of `rowBytes` bytes (each preceded by its filter byte) in place; returns the pixel bytes */
void unfilter(const uint8_t* src, uint32_t rows, size_t rowBytes, size_t bpp, std::vector<uint8_t>& dst)
{
    dst.assign((size_t)rows * rowBytes, 0);
    for (uint32_t y = 0; y < rows; y++) {
        const uint8_t ft = src[(size_t)y * (rowBytes + 1)];
        const uint8_t* in = src + (size_t)y * (rowBytes + 1) + 1;
        uint8_t* cur = dst.data() + (size_t)y * rowBytes;
        const uint8_t* up = y ? cur - rowBytes : nullptr;
        for (size_t i = 0; i < rowBytes; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = in[i];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break;
                          default: throw std::runtime_error("png: unknown filter type"); }
            cur[i] = (uint8_t)v;
        }
    }
}

bool decodePng(const std::vector<uint8_t>& d, DecodedImage& img, std::string& err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || memcmp(d.data(), sig, 8)) { err = "not a PNG file"; return false; }
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t at = 8; at + 12 <= d.size();) {
        const uint32_t n = be32(&d[at]); const uint8_t* type = &d[at + 4];
        if (at + 12 + n > d.size()) throw std::runtime_error("png: truncated chunk");
        const uint8_t* body = &d[at + 8];
        if (!memcmp(type, "IHDR", 4)) { if (n < 13) throw std::runtime_error("png: short IHDR"); W = be32(body); H = be32(body + 4); depth = body[8];
            ctype = body[9]; interlace = body[12]; if (body[10] || body[11]) throw std::runtime_error("png: unknown compression / filter method"); }
        else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + n);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + n);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
        else if (!memcmp(type, "IEND", 4)) break;
        at += 12 + n;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!W || !H || !channels || !(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) || interlace > 1)
        throw std::runtime_error("png: unsupported header");
    if (ctype == 3 && palette.empty()) throw std::runtime_error("png: palette image without PLTE");
    if (!ImageDimensionsOk(W, H)) throw std::runtime_error("png: dimensions beyond the 16384 a 2-D texture can have");
    if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) throw std::runtime_error("png: bit depth not allowed for this colour type");
    if (ctype == 3 && depth == 16) throw std::runtime_error("png: 16-bit palette indices");
    std::vector<uint8_t> raw; zlibInflate(idat, raw);
    const size_t bitsPerPixel = (size_t)channels * depth, bpp = (bitsPerPixel + 7) / 8;
    {   /* the scanlines the header promises must all be there before anything of that size is allocated */
        static const uint32_t xs0[7] = {0, 4, 0, 2, 0, 1, 0}, ys0[7] = {0, 0, 4, 0, 2, 0, 1}, dxs0[7] = {8, 8, 4, 4, 2, 2, 1}, dys0[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t want = 0;
        if (!interlace) want = (size_t)H * (((size_t)W * bitsPerPixel + 7) / 8 + 1);
        else for (int p = 0; p < 7; p++) { const size_t pw = (W + dxs0[p] - 1 - xs0[p]) / dxs0[p], ph = (H + dys0[p] - 1 - ys0[p]) / dys0[p];
            if (pw && ph) want += ph * ((pw * bitsPerPixel + 7) / 8 + 1); }
        if (raw.size() < want) throw std::runtime_error("png: image data too short");
    }
    /* samples[y][x][c] as 16-bit values at the file's bit depth */
    std::vector<uint16_t> samples((size_t)W * H * channels, 0);
    auto unpackPass = [&](const uint8_t* src, uint32_t pw, uint32_t ph, uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) -> size_t {
        if (!pw || !ph) return 0;
        const size_t rowBytes = ((size_t)pw * bitsPerPixel + 7) / 8;
        std::vector<uint8_t> px; unfilter(src, ph, rowBytes, bpp, px);
        for (uint32_t y = 0; y < ph; y++) for (uint32_t x = 0; x < pw; x++) for (int c = 0; c < channels; c++) {
            const uint8_t* row = px.data() + (size_t)y * rowBytes; uint16_t v;
            if (depth == 16) { const uint8_t* p = row + ((size_t)x * channels + c) * 2; v = (uint16_t)((p[0] << 8) | p[1]); }
            else if (depth == 8) v = row[(size_t)x * channels + c];
            else { const size_t bit = ((size_t)x * channels + c) * depth; v = (uint16_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
            samples[(((size_t)(y0 + y * dy)) * W + (x0 + x * dx)) * channels + c] = v;
        }
        return (size_t)ph * (rowBytes + 1);
    };
    if (!interlace) {
        if (raw.size() < (size_t)H * ((W * bitsPerPixel + 7) / 8 + 1)) throw std::runtime_error("png: image data too short");
        unpackPass(raw.data(), W, H, 0, 0, 1, 1);
    } else {
        static const uint32_t xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dxs[7] = {8, 8, 4, 4, 2, 2, 1}, dys[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t at = 0;
        for (int p = 0; p < 7; p++) {
            const uint32_t pw = (W + dxs[p] - 1 - xs[p]) / dxs[p], ph = (H + dys[p] - 1 - ys[p]) / dys[p];
            if (at > raw.size()) throw std::runtime_error("png: interlaced data too short");
            at += unpackPass(raw.data() + at, pw, ph, xs[p], ys[p], dxs[p], dys[p]);
        }
    }
    img.width = W; img.height = H; img.normalized = true; img.hasAlpha = false;
    img.texels.resize((size_t)W * H);
    const float maxv = (float)((1u << depth) - 1);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const uint16_t* s = &samples[i * channels];
        TbFloat4 t;
        switch (ctype) {
        case 0: { /* WIC grey -> R8_UNORM / R16_UNORM; a tRNS colour key makes WIC expand to RGBA */
            float g = (float)s[0] / maxv;
            if (trns.size() >= 2) { bool key = s[0] == (uint16_t)((trns[0] << 8) | trns[1]); t = TbFloat4{g, g, g, key ? 0.0f : 1.0f}; }
            else t = TbFloat4{g, 0.0f, 0.0f, 1.0f};
            break; }
        case 2: {
            bool key = trns.size() >= 6 && s[0] == (uint16_t)((trns[0] << 8) | trns[1]) && s[1] == (uint16_t)((trns[2] << 8) | trns[3]) && s[2] ==
                (uint16_t)((trns[4] << 8) | trns[5]);
            t = TbFloat4{(float)s[0] / maxv, (float)s[1] / maxv, (float)s[2] / maxv, key ? 0.0f : 1.0f};
            break; }
        case 3: {
            const size_t idx = s[0];
            if (idx * 3 + 2 >= palette.size()) throw std::runtime_error("png: palette index out of range");
            const float a = idx < trns.size() ? (float)trns[idx] / 255.0f : 1.0f;
            t = TbFloat4{(float)palette[idx * 3] / 255.0f, (float)palette[idx * 3 + 1] / 255.0f, (float)palette[idx * 3 + 2] / 255.0f, a};
            break; }
        case 4: { float g = (float)s[0] / maxv; t = TbFloat4{g, g, g, (float)s[1] / maxv}; break; }
        default: t = TbFloat4{(float)s[0] / maxv, (float)s[1] / maxv, (float)s[2] / maxv, (float)s[3] / maxv}; break;
        }
        if (t.w != 1.0f) img.hasAlpha = true;
        img.texels[i] = t;
    }
    return true;
}

/* ---- TGA (DirectXTex LoadFromTGAFile: 8-bit grey -> R8_UNORM, 16-bit -> B5G5R5A1, 24/32-bit -> R8G8B8A8) --------- */
bool decodeTga(const std::vector<uint8_t>& d, DecodedImage& img, std::string& err)
{
    if (d.size() < 18) { err = "TGA: file too short"; return false; }
    const int idLen = d[0], cmapType = d[1], type = d[2], cmapFirst = d[3] | (d[4] << 8), cmapLen = d[5] | (d[6] << 8), cmapBits = d[7];
    const uint32_t W = d[12] | (d[13] << 8), H = d[14] | (d[15] << 8); const int bits = d[16], desc = d[17];
    const bool rle = type >= 9; const int base = rle ? type - 8 : type;
    if (!W || !H || !(base == 1 || base == 2 || base == 3)) { err = "TGA: unsupported image type"; return false; }
    if ((base == 2 && !(bits == 16 || bits == 24 || bits == 32)) || (base == 3 && bits != 8) || (base == 1 && (bits != 8 || cmapType != 1))) { err =
        "TGA: unsupported pixel depth"; return false; }
    size_t at = 18 + (size_t)idLen;
    const size_t cmapEntry = (size_t)(cmapBits + 7) / 8;
    if (base == 1 && !(cmapBits == 15 || cmapBits == 16 || cmapBits == 24 || cmapBits == 32)) { err = "TGA: unsupported colour-map entry size"; return false; }
    const uint8_t* cmap = cmapType ? d.data() + at : nullptr;
    at += cmapType ? (size_t)cmapLen * cmapEntry : 0;
    if (at > d.size()) throw std::runtime_error("TGA: truncated colour map");
    if (!ImageDimensionsOk(W, H)) throw std::runtime_error("TGA: dimensions beyond the 16384 a 2-D texture can have");
    const size_t px = (size_t)bits / 8;
    if ((size_t)W * H * px > (rle ? 128 : 1) * (d.size() - at)) throw std::runtime_error("TGA: truncated");   /* a run packet holds at most 128 pixels */
    std::vector<uint8_t> pix((size_t)W * H * px);
    if (!rle) { if (at + pix.size() > d.size()) throw std::runtime_error("TGA: truncated"); memcpy(pix.data(), d.data() + at, pix.size()); }
    else {
        size_t o = 0;
        while (o < pix.size()) {
            if (at >= d.size()) throw std::runtime_error("TGA: truncated RLE stream");
            const int hdr = d[at++], count = (hdr & 0x7f) + 1;
            if (hdr & 0x80) { if (at + px > d.size()) throw std::runtime_error("TGA: truncated");
                for (int k = 0; k < count && o < pix.size(); k++) { memcpy(&pix[o], &d[at], px); o += px; } at += px; }
            else { const size_t n = (size_t)count * px; if (at + n > d.size() || o + n > pix.size()) throw std::runtime_error("TGA: truncated");
                memcpy(&pix[o], &d[at], n); o += n; at += n; }
        }
    }
    auto colour = [&](const uint8_t* p, size_t nbytes) -> TbFloat4 { /* little-endian BGR(A) / A1R5G5B5 */
        if (nbytes == 2) { const uint16_t v = (uint16_t)(p[0] | (p[1] << 8));
            return TbFloat4{(float)((v >> 10) & 31) / 31.0f, (float)((v >> 5) & 31) / 31.0f, (float)(v & 31) / 31.0f, (desc & 0x0f) ? (float)(v >> 15) : 1.0f};
            }
        return TbFloat4{(float)p[2] / 255.0f, (float)p[1] / 255.0f, (float)p[0] / 255.0f, nbytes == 4 ? (float)p[3] / 255.0f : 1.0f};
    };
    img.width = W; img.height = H; img.normalized = true; img.hasAlpha = false;
    img.texels.resize((size_t)W * H);
    const bool topDown = (desc & 0x20) != 0, rightLeft = (desc & 0x10) != 0;
    bool anyAlpha = false, allZeroAlpha = true;
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
        const uint8_t* p = &pix[((size_t)y * W + x) * px];
        TbFloat4 t;
        if (base == 3) t = TbFloat4{(float)p[0] / 255.0f, 0.0f, 0.0f, 1.0f};
        else if (base == 1) { const int idx = p[0] - cmapFirst; if (idx < 0 || idx >= cmapLen) throw std::runtime_error("TGA: colour-map index out of range");
            t = colour(cmap + (size_t)idx * cmapEntry, cmapEntry); }
        else t = colour(p, px);
        if (t.w != 0.0f) allZeroAlpha = false;
        if (t.w != 1.0f) anyAlpha = true;
        img.texels[(size_t)(topDown ? y : H - 1 - y) * W + (rightLeft ? W - 1 - x : x)] = t;
    }
    if (allZeroAlpha) { for (TbFloat4& t : img.texels) t.w = 1.0f; anyAlpha = false; } /* DirectXTex: an all-zero alpha channel is treated as opaque */
    img.hasAlpha = anyAlpha;
    return true;
}

} // namespace

bool DecodeJpegBmpDds(const std::string& file, const std::vector<uint8_t>& d, int kind, DecodedImage& img, std::string& err); /* image_formats.cpp */

bool DecodeImageFile(const std::string& file, DecodedImage& img, std::string& err)
{
    std::vector<uint8_t> d;
    if (!readFile(file, d, err)) return false;
    auto ends = [&](const char* s) { size_t n = strlen(s); if (file.size() < n) return false;
        for (size_t i = 0; i < n; i++) if (tolower(file[file.size() - n + i]) != s[i]) return false; return true; };
    try {
        if (ends(".png")) return decodePng(d, img, err);
        if (ends(".tga")) return decodeTga(d, img, err);
        if (ends(".jpg") || ends(".jpeg")) return DecodeJpegBmpDds(file, d, 0, img, err);
        if (ends(".bmp")) return DecodeJpegBmpDds(file, d, 1, img, err);
        if (ends(".dds")) return DecodeJpegBmpDds(file, d, 2, img, err);
    } catch (const std::exception& e) { err = std::string(e.what()) + " ('" + file + "')"; return false; }
    err = "unsupported image format for '" + file + "' (this build decodes .hdr, .pfm, .png, .tga, .jpg, .bmp and .dds)";
    return false;
}

} // namespace tbhost
