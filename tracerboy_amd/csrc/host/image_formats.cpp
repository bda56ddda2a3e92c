/* image_formats.cpp -- JPEG, BMP and DDS decoders for image textures.
 *
 * The reference hands .dds files to DirectXTex::LoadFromDDSFile(DDS_FLAGS_NO_16BPP) and everything that is not .hdr / .tga / .dds to
 * Windows Imaging (LoadFromWICFile): /root/reference/TracerBoy/TracerBoy.cpp:2214-2226.  What reaches the shaders is the typed
 * load of the resulting DXGI format, reproduced here as RGBA32F texels like image_decode.cpp does for PNG / TGA:
 *   JPEG  sequential (baseline / extended) and progressive Huffman, 8-bit, 1 or 3 components (YCbCr or Adobe RGB), any sampling
 *         factors, restart intervals, any split into scans.  Arithmetic as in the IJG library every decoder is measured against: the "slow integer" inverse DCT
 *         (jidctint), triangle ("fancy") chroma upsampling for 2:1 horizontally and 2:1 x 2:1 (jdsample), 16-bit fixed-point
 *         YCbCr -> RGB (jdcolor).  WIC's own decoder is not bit-specified; tests pin this one against Pillow (libjpeg-turbo).
 *         Grey -> (g, 0, 0, 1) like 8bppGray -> R8_UNORM; colour -> R8G8B8A8_UNORM, alpha 1.
 *   BMP   BITMAPINFOHEADER / V4 / V5; 1, 4, 8 bit palettes, 16 (5-5-5 or bit fields), 24, 32 bit, BI_RGB / BI_BITFIELDS, both row orders
 *   DDS   top mip of the first surface: uncompressed 8 / 16 / 24 / 32-bit masks (16-bit formats expanded to 8888: NO_16BPP),
 *         L8 / A8L8 / A8, BC1-BC5 and BC7 (DXT1-5, ATI1/2; DX10 header incl. _SRGB, which only flags gamma), R16G16B16A16_FLOAT / UNORM,
 *         R32G32B32A32_FLOAT, R32_FLOAT.  Block formats are decoded the way D3D specifies the sampler's view of them (endpoints
 *         as UNORM, interpolated in floating point).
 * No third-party code. */
#include "host_scene.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace tbhost {
namespace {

TbFloat4 px(float r, float g, float b, float a) { TbFloat4 t; t.x = r; t.y = g; t.z = b; t.w = a; return t; }

/* ---- JPEG ------------------------------------------------------------------------------------------------------------------ */
struct JpegHuff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool present = false;
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) { valptr[l] = k; mincode[l] = code; code += bits[l]; k += bits[l]; maxcode[l] = bits[l] ? code - 1 : -1; code <<= 1; }
        maxcode[17] = 0x7fffffff;
    }
};

struct JpegBits {
    const uint8_t* p; size_t n, at; uint32_t acc = 0; int cnt = 0; bool hitMarker = false;
    JpegBits(const uint8_t* d, size_t len, size_t start) : p(d), n(len), at(start) {}
    void fill()
    {
        while (cnt <= 24) {
            uint32_t b = 0;
            if (!hitMarker && at < n) {
                b = p[at];
                if (b == 0xff) {
                    if (at + 1 < n && p[at + 1] == 0x00) at += 2;            /* stuffed zero */
                    else { hitMarker = true; b = 0; }                        /* a marker: feed zeros until the caller deals with it */
                } else at++;
            }
            acc |= b << (24 - cnt); cnt += 8;
        }
    }
    int get(int k) { if (!k) return 0; if (cnt < k) fill(); const int v = (int)(acc >> (32 - k)); acc <<= k; cnt -= k; return v; }
    int decode(const JpegHuff& h)
    {
        int code = 0;
        for (int l = 1; l <= 16; l++) { code = (code << 1) | get(1);
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]]; }
        throw std::runtime_error("jpeg: bad Huffman code");
    }
    static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
    void restart() /* at an RSTn: drop the bit buffer, step over the marker */
    {
        acc = 0; cnt = 0;
        if (hitMarker) { hitMarker = false; if (at + 1 < n && p[at] == 0xff && p[at + 1] >= 0xd0 && p[at + 1] <= 0xd7) at += 2;
            else throw std::runtime_error("jpeg: restart marker expected"); }
        else { /* the marker may not have been reached by the bit reader yet */
            while (at + 1 < n && !(p[at] == 0xff && p[at + 1] >= 0xd0 && p[at + 1] <= 0xd7)) at++;
            if (at + 1 >= n) throw std::runtime_error("jpeg: restart marker expected");
            at += 2;
        }
    }
};

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* jidctint.c (IJG "slow but accurate integer" inverse DCT, Loeffler-Ligtenberg-Moschytz): CONST_BITS 13, PASS1_BITS 2 */
void idctIslow(const int* coef /* dequantised, natural order */, uint8_t* out, size_t stride)
{
    const int CONST_BITS = 13, PASS1_BITS = 2;
    const int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137,
        F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
    int ws[64];
    for (int c = 0; c < 8; c++) {
        const int* in = coef + c; int* w = ws + c;
        if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) { const int dc = in[0] * (1 << PASS1_BITS);
            for (int r = 0; r < 8; r++) w[8 * r] = dc; continue; }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F_0_541, tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        z2 = in[0]; z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CONST_BITS), tmp1 = (z2 - z3) * (1L << CONST_BITS);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS); w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS); w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; r++) {
        const int* w = ws + 8 * r; uint8_t* o = out + stride * r;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541, tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        long tmp0 = ((long)w[0] + w[4]) * (1L << CONST_BITS), tmp1 = ((long)w[0] - w[4]) * (1L << CONST_BITS);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CONST_BITS + PASS1_BITS + 3;
        o[0] = clamp255(descale(tmp10 + tmp3, S) + 128); o[7] = clamp255(descale(tmp10 - tmp3, S) + 128);
        o[1] = clamp255(descale(tmp11 + tmp2, S) + 128); o[6] = clamp255(descale(tmp11 - tmp2, S) + 128);
        o[2] = clamp255(descale(tmp12 + tmp1, S) + 128); o[5] = clamp255(descale(tmp12 - tmp1, S) + 128);
        o[3] = clamp255(descale(tmp13 + tmp0, S) + 128); o[4] = clamp255(descale(tmp13 - tmp0, S) + 128);
    }
}

struct JpegComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    uint32_t bw = 0, bh = 0 /* blocks of the padded (MCU) grid */, cw = 0, ch = 0 /* blocks the component itself needs */, dw = 0,
    dh = 0 /* downsampled size */;
                  std::vector<int16_t> coef; /* bw * bh * 64, natural order, before dequantisation */ std::vector<uint8_t> plane; size_t stride = 0; };

/* One scan of a sequential or progressive frame into the components' coefficient arrays (ITU T.81 F.2 / G.1).  Sequential scans carry
 * whole blocks (Ss = 0, Se = 63, Ah = Al = 0); progressive ones a band of coefficients at a bit position, first pass or refinement. */
void decodeJpegScan(JpegBits& br, std::vector<JpegComp*>& sc, const JpegHuff* dc, const JpegHuff* ac, bool progressive, int Ss, int Se, int Ah, int Al,
                    uint32_t mcusX, uint32_t mcusY, int restartInterval)
{
    const bool interleaved = sc.size() > 1;
    uint32_t unitsX = mcusX, unitsY = mcusY;
    if (!interleaved) { unitsX = sc[0]->cw; unitsY = sc[0]->ch; } /* a one-component scan walks that component's own blocks */
    for (JpegComp* c : sc) c->pred = 0;
    uint32_t eobrun = 0; int toRestart = restartInterval;
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto block = [&](JpegComp& c, uint32_t bx, uint32_t by) {
        int16_t* co = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
        if (!progressive) {
            const int t = br.decode(dc[c.td]); if (t > 11) throw std::runtime_error("jpeg: bad DC size");
            c.pred += t ? JpegBits::extend(br.get(t), t) : 0; co[0] = (int16_t)c.pred;
            for (int k = 1; k < 64;) {
                const int rs = br.decode(ac[c.ta]), r = rs >> 4, sz = rs & 15;
                if (!sz) { if (r == 15) { k += 16; continue; } break; }
                k += r; if (k > 63) throw std::runtime_error("jpeg: AC run past the block");
                co[kZigzag[k]] = (int16_t)JpegBits::extend(br.get(sz), sz); k++;
            }
            return;
        }
        if (Ss == 0) { /* DC band */
            if (Ah == 0) { const int t = br.decode(dc[c.td]); if (t > 11) throw std::runtime_error("jpeg: bad DC size");
                c.pred += t ? JpegBits::extend(br.get(t), t) : 0; co[0] = (int16_t)(c.pred * (1 << Al)); }
            else if (br.get(1)) co[0] = (int16_t)(co[0] | p1);
            return;
        }
        if (Ah == 0) { /* AC band, first pass (G.1.2.2) */
            if (eobrun) { eobrun--; return; }
            for (int k = Ss; k <= Se;) {
                const int rs = br.decode(ac[c.ta]), r = rs >> 4, sz = rs & 15;
                if (!sz) { if (r == 15) { k += 16; continue; } eobrun = (1u << r) - 1u; if (r) eobrun += (uint32_t)br.get(r); break; }
                k += r; if (k > Se) throw std::runtime_error("jpeg: AC run past the band");
                co[kZigzag[k]] = (int16_t)(JpegBits::extend(br.get(sz), sz) * (1 << Al)); k++;
            }
            return;
        }
        /* AC band, refinement (G.1.2.3): new coefficients of magnitude 1 at this bit, correction bits for the ones already nonzero */
        int k = Ss;
        if (!eobrun) {
            for (; k <= Se; k++) {
                const int rs = br.decode(ac[c.ta]); int r = rs >> 4; const int sz = rs & 15; int val = 0;
                if (sz) { if (sz != 1) throw std::runtime_error("jpeg: bad refinement code"); val = br.get(1) ? p1 : m1; }
                else if (r != 15) { eobrun = 1u << r; if (r) eobrun += (uint32_t)br.get(r); break; }
                for (; k <= Se; k++) { /* skip r zero-history coefficients, refining the nonzero ones passed on the way */
                    int16_t& q = co[kZigzag[k]];
                    if (q) { if (br.get(1) && !(q & p1)) q = (int16_t)(q + (q >= 0 ? p1 : m1)); }
                    else { if (--r < 0) break; }
                }
                if (val && k <= Se) co[kZigzag[k]] = (int16_t)val;
            }
        }
        if (eobrun) { /* the rest of this block: correction bits only */
            for (; k <= Se; k++) { int16_t& q = co[kZigzag[k]]; if (q && br.get(1) && !(q & p1)) q = (int16_t)(q + (q >= 0 ? p1 : m1)); }
            eobrun--;
        }
    };
    for (uint32_t uy = 0; uy < unitsY; uy++) for (uint32_t ux = 0; ux < unitsX; ux++) {
        if (restartInterval && toRestart == 0) { br.restart(); for (JpegComp* c : sc) c->pred = 0; eobrun = 0; toRestart = restartInterval; }
        if (interleaved) { for (JpegComp* c : sc) for (int by = 0; by < c->v; by++) for (int bx = 0; bx < c->h; bx++) block(*c,
            ux * (uint32_t)c->h + (uint32_t)bx, uy * (uint32_t)c->v + (uint32_t)by); }
        else block(*sc[0], ux, uy);
        if (restartInterval) toRestart--;
    }
}

bool decodeJpeg(const std::vector<uint8_t>& d, DecodedImage& img, std::string& err)
{
    if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) { err = "not a JPEG file"; return false; }
    uint16_t qt[4][64]; bool qtSet[4] = {false, false, false, false};
    JpegHuff dc[4], ac[4];
    std::vector<JpegComp> comps; uint32_t W = 0, H = 0; int hmax = 1, vmax = 1, restartInterval = 0; int adobeTransform = -1;
        bool sawSof = false, progressive = false, sawScan = false;
    uint32_t mcusX = 0, mcusY = 0;
    size_t at = 2;
    auto u16 = [&](size_t o) -> uint32_t { if (o + 1 >= d.size()) throw std::runtime_error("jpeg: truncated"); return ((uint32_t)d[o] << 8) | d[o + 1]; };
    for (bool done = false; !done;) {
        while (at < d.size() && d[at] != 0xff) at++;
        while (at < d.size() && d[at] == 0xff) at++;
        /* a missing EOI after complete scans is tolerated, like libjpeg */
        if (at >= d.size()) { if (sawScan) break; throw std::runtime_error("jpeg: no scan found"); }
        const uint8_t m = d[at++];
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01 || m == 0x00) continue;
        if (m == 0xd9) { if (!sawScan) throw std::runtime_error("jpeg: end of image before any scan"); break; }
        const uint32_t len = u16(at); if (len < 2 || at + len > d.size()) throw std::runtime_error("jpeg: bad segment length");
        const size_t seg = at + 2, end = at + len;
        if (m == 0xdb) { /* DQT */
            size_t p = seg;
            while (p < end) { const int pq = d[p] >> 4, tq = d[p] & 15; p++; if (tq > 3) throw std::runtime_error("jpeg: bad quantisation table id");
                for (int i = 0; i < 64; i++) { if (p + (pq ? 2 : 1) > end) throw std::runtime_error("jpeg: truncated DQT");
                    qt[tq][kZigzag[i]] = pq ? (uint16_t)u16(p) : d[p]; p += pq ? 2 : 1; }
                qtSet[tq] = true; }
        } else if (m == 0xc4) { /* DHT */
            size_t p = seg;
            while (p < end) { const int tc = d[p] >> 4, th = d[p] & 15; p++; if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad Huffman table id");
                JpegHuff& h = tc ? ac[th] : dc[th]; int total = 0;
                if (p + 16 > end) throw std::runtime_error("jpeg: truncated DHT");
                for (int l = 1; l <= 16; l++) { h.bits[l] = d[p++]; total += h.bits[l]; }
                if (total > 256 || p + total > end) throw std::runtime_error("jpeg: bad Huffman table");
                memcpy(h.vals, &d[p], (size_t)total); p += total; h.build(); h.present = true; }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) { /* SOF0 / SOF1 sequential, SOF2 progressive; Huffman, 8 bits */
            if (sawSof) throw std::runtime_error("jpeg: second frame header");
            if (len < 8) throw std::runtime_error("jpeg: short frame header");
            if (d[seg] != 8) throw std::runtime_error("jpeg: only 8-bit precision is supported");
            progressive = m == 0xc2;
            H = u16(seg + 1); W = u16(seg + 3); const int nc = d[seg + 5];
            if (!W || !H || (nc != 1 && nc != 3)) throw std::runtime_error("jpeg: unsupported component count (grey and three-component files are decoded)");
            if (seg + 6 + 3 * (size_t)nc > end) throw std::runtime_error("jpeg: short frame header");
            if (!ImageDimensionsOk(W, H)) throw std::runtime_error("jpeg: dimensions beyond the 16384 a 2-D texture can have");
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) { JpegComp& c = comps[(size_t)i]; c.id = d[seg + 6 + 3 * i]; c.h = d[seg + 7 + 3 * i] >> 4;
                c.v = d[seg + 7 + 3 * i] & 15; c.tq = d[seg + 8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw std::runtime_error("jpeg: bad sampling factors"); hmax = std::max(hmax, c.h);
                    vmax = std::max(vmax, c.v); }
            /* a single component is never interleaved: its factors only scale the (absent) others */
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
            mcusX = (W + 8u * (uint32_t)hmax - 1) / (8u * (uint32_t)hmax); mcusY = (H + 8u * (uint32_t)vmax - 1) / (8u * (uint32_t)vmax);
            for (JpegComp& c : comps) {
                c.dw = (W * (uint32_t)c.h + (uint32_t)hmax - 1) / (uint32_t)hmax; c.dh = (H * (uint32_t)c.v + (uint32_t)vmax - 1) / (uint32_t)vmax;
                c.cw = (c.dw + 7) / 8; c.ch = (c.dh + 7) / 8; c.bw = mcusX * (uint32_t)c.h; c.bh = mcusY * (uint32_t)c.v;
                if ((uint64_t)c.bw * c.bh > (1u << 24)) throw std::runtime_error("jpeg: image too large");
                /* a block costs at least a bit of DC */
                if ((uint64_t)c.bw * c.bh > 64ull * d.size() + 4096) throw std::runtime_error("jpeg: the file is too short for the frame its header describes");
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
            sawSof = true;
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            err = "lossless / hierarchical / arithmetic-coded JPEG is not supported (sequential and progressive Huffman files are)"; return false;
        } else if (m == 0xdd) { if (len < 4) throw std::runtime_error("jpeg: short DRI"); restartInterval = (int)u16(seg); }
        else if (m == 0xee && len >= 14 && !memcmp(&d[seg], "Adobe", 5)) adobeTransform = d[seg + 11];
        else if (m == 0xda) { /* SOS */
            if (!sawSof) throw std::runtime_error("jpeg: scan before frame header");
            const int ns = d[seg];
                if (ns < 1 || ns > (int)comps.size() || seg + 1 + 2 * (size_t)ns + 3 > end) throw std::runtime_error("jpeg: bad scan header");
            std::vector<JpegComp*> sc;
            for (int i = 0; i < ns; i++) { const int cid = d[seg + 1 + 2 * i]; JpegComp* found = nullptr;
                for (JpegComp& c : comps) if (c.id == cid) found = &c;
                if (!found) throw std::runtime_error("jpeg: scan names an unknown component");
                found->td = d[seg + 2 + 2 * i] >> 4; found->ta = d[seg + 2 + 2 * i] & 15;
                    if (found->td > 3 || found->ta > 3) throw std::runtime_error("jpeg: bad table selector"); sc.push_back(found); }
            const int Ss = d[seg + 1 + 2 * ns], Se = d[seg + 2 + 2 * ns], Ah = d[seg + 3 + 2 * ns] >> 4, Al = d[seg + 3 + 2 * ns] & 15;
            if (progressive) { if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13)
                throw std::runtime_error("jpeg: bad progressive scan parameters"); }
            else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) throw std::runtime_error("jpeg: bad sequential scan parameters");
            for (JpegComp* c : sc) { if ((!progressive || (Ss == 0 && Ah == 0)) && !dc[c->td].present)
                throw std::runtime_error("jpeg: scan refers to a DC table that was not defined");
                                     if ((!progressive || Ss > 0) && !ac[c->ta].present)
                                         throw std::runtime_error("jpeg: scan refers to an AC table that was not defined"); }
            JpegBits br(d.data(), d.size(), end);
            decodeJpegScan(br, sc, dc, ac, progressive, Ss, Se, Ah, Al, mcusX, mcusY, restartInterval);
            sawScan = true;
            at = br.at; continue; /* the next marker is searched from where the entropy decoder stopped */
        }
        at = end;
    }
    /* dequantise + inverse DCT into the component planes */
    int coefBlock[64];
    for (JpegComp& c : comps) {
        if (!qtSet[c.tq]) throw std::runtime_error("jpeg: frame refers to a quantisation table that was not defined");
        c.stride = (size_t)c.bw * 8; c.plane.assign(c.stride * c.bh * 8, 0);
        for (uint32_t by = 0; by < c.bh; by++) for (uint32_t bx = 0; bx < c.bw; bx++) {
            const int16_t* co = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
            for (int i = 0; i < 64; i++) coefBlock[i] = (int)co[i] * (int)qt[c.tq][i];
            idctIslow(coefBlock, c.plane.data() + ((size_t)by * 8) * c.stride + (size_t)bx * 8, c.stride);
        }
        c.coef.clear(); c.coef.shrink_to_fit();
    }
    /* upsample every component to W x H (jdsample.c) */
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const JpegComp& c = comps[ci]; std::vector<uint8_t>& o = full[ci]; o.assign((size_t)W * H, 0);
        const int hx = hmax / c.h, vx = vmax / c.v; const bool exact = hmax % c.h == 0 && vmax % c.v == 0;
        auto in = [&](uint32_t x, uint32_t y) -> int { return c.plane[(size_t)y * c.stride + x]; };
        if (exact && hx == 1 && vx == 1) { for (uint32_t y = 0; y < H; y++) memcpy(&o[(size_t)y * W], &c.plane[(size_t)y * c.stride], W); }
        else if (exact && hx == 2 && vx == 1 && c.dw > 2) { /* h2v1_fancy_upsample: 3/4 nearer + 1/4 further, rounding 1 / 2 alternately */
            for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
                const uint32_t i = x >> 1; int v;
                if (x & 1) v = i + 1 < c.dw ? (3 * in(i, y) + in(i + 1, y) + 2) >> 2 : in(i, y);
                else v = i > 0 ? (3 * in(i, y) + in(i - 1, y) + 1) >> 2 : in(i, y);
                o[(size_t)y * W + x] = (uint8_t)v;
            }
        } else if (exact && hx == 2 && vx == 2 && c.dw > 2) { /* h2v2_fancy_upsample: 9/16, 3/16, 3/16, 1/16, rounding 8 / 7 alternately */
            for (uint32_t y = 0; y < H; y++) {
                /* the context row beyond an edge repeats the edge row */
                const uint32_t j = y >> 1; const uint32_t far = (y & 1) ? (j + 1 < c.dh ? j + 1 : j) : (j > 0 ? j - 1 : j);
                auto colsum = [&](uint32_t i) { return 3 * in(i, j) + in(i, far); };
                for (uint32_t x = 0; x < W; x++) {
                    const uint32_t i = x >> 1; const int cur = colsum(i); int v;
                    if (x & 1) v = i + 1 < c.dw ? (3 * cur + colsum(i + 1) + 7) >> 4 : (4 * cur + 7) >> 4;
                    else v = i > 0 ? (3 * cur + colsum(i - 1) + 8) >> 4 : (4 * cur + 8) >> 4;
                    o[(size_t)y * W + x] = (uint8_t)v;
                }
            }
        } else { /* replication (int_upsample) for every other ratio; ratios that are not whole numbers take the nearest sample below */
            for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
                const uint32_t sx = std::min<uint32_t>((uint32_t)((uint64_t)x * (uint32_t)c.h / (uint32_t)hmax), c.bw * 8 - 1),
                    sy = std::min<uint32_t>((uint32_t)((uint64_t)y * (uint32_t)c.v / (uint32_t)vmax), c.bh * 8 - 1);
                o[(size_t)y * W + x] = (uint8_t)in(sx, sy);
            }
        }
    }
    img.width = W; img.height = H; img.normalized = true; img.hasAlpha = false;
    img.texels.resize((size_t)W * H);
    if (comps.size() == 1) { for (size_t i = 0; i < img.texels.size(); i++) img.texels[i] = px((float)full[0][i] / 255.0f, 0.0f, 0.0f, 1.0f); return true; }
    const bool ycc = adobeTransform < 0 ? !(comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') : adobeTransform == 1;
    /* jdcolor.c build_ycc_rgb_table: SCALEBITS 16 */
    int crR[256], cbB[256]; long crG[256], cbG[256];
    for (int i = 0; i < 256; i++) { const long x = i - 128;
        crR[i] = (int)((91881L * x + 32768L) >> 16); cbB[i] = (int)((116130L * x + 32768L) >> 16); crG[i] = -46802L * x; cbG[i] = -22554L * x + 32768L; }
    for (size_t i = 0; i < img.texels.size(); i++) {
        int r = full[0][i], g = full[1][i], b = full[2][i];
        if (ycc) { const int y = r, cb = g, cr = b; r = clamp255(y + crR[cr]); g = clamp255(y + (int)((cbG[cb] + crG[cr]) >> 16)); b = clamp255(y + cbB[cb]); }
        img.texels[i] = px((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, 1.0f);
    }
    return true;
}

/* ---- BMP ------------------------------------------------------------------------------------------------------------------- */
uint32_t rd32(const std::vector<uint8_t>& d, size_t o) { if (o + 4 > d.size()) throw std::runtime_error("image: truncated");
    return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); }
uint32_t rd16(const std::vector<uint8_t>& d, size_t o) { if (o + 2 > d.size()) throw std::runtime_error("image: truncated");
    return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); }

/* value of a bit field scaled to [0, 1]: UNORM of the field's own width */
float maskUnorm(uint32_t v, uint32_t mask)
{
    if (!mask) return 0.0f;
    int shift = 0; while (!((mask >> shift) & 1u)) shift++;
    const uint32_t top = mask >> shift;
    return (float)((v & mask) >> shift) / (float)top;
}

bool decodeBmp(const std::vector<uint8_t>& d, DecodedImage& img, std::string& err)
{
    if (d.size() < 26 || d[0] != 'B' || d[1] != 'M') { err = "not a BMP file"; return false; }
    const uint32_t dataOff = rd32(d, 10), hdr = rd32(d, 14);
    if (hdr < 40) { err = "BMP: OS/2 core headers are not supported"; return false; }
    const int32_t w = (int32_t)rd32(d, 18), hs = (int32_t)rd32(d, 22);
    const uint32_t bpp = rd16(d, 28), comp = rd32(d, 30); uint32_t colours = rd32(d, 46);
    if (w <= 0 || hs == 0 || w > 16384 || hs > 16384 || hs < -16384) { err = "BMP: bad dimensions (a 2-D texture has at most 16384 texels a side)";
        return false; }
    const uint32_t W = (uint32_t)w, H = (uint32_t)(hs < 0 ? -hs : hs); const bool topDown = hs < 0;
    if (comp != 0 && comp != 3) { err = "BMP: RLE / embedded JPEG / PNG compression is not supported"; return false; }
    uint32_t rm = 0, gm = 0, bm = 0, am = 0;
    if (comp == 3) { const size_t mo = hdr >= 52 ? 54 : 14 + 40; rm = rd32(d, mo); gm = rd32(d, mo + 4); bm = rd32(d, mo + 8);
        if (hdr >= 56) am = rd32(d, mo + 12); }
    else if (bpp == 16) { rm = 0x7c00; gm = 0x03e0; bm = 0x001f; }
    else if (bpp == 32) { rm = 0x00ff0000; gm = 0x0000ff00; bm = 0x000000ff; if (hdr >= 56) am = rd32(d, 14 + 40 + 12); if (hdr < 108) am = 0; }
    std::vector<TbFloat4> pal;
    if (bpp <= 8) {
        if (!colours) colours = 1u << bpp;
        if (colours > 256) throw std::runtime_error("BMP: bad palette size");
        const size_t po = 14 + (size_t)hdr + (comp == 3 && hdr == 40 ? 12 : 0);
        for (uint32_t i = 0; i < colours; i++) { if (po + 4 * i + 3 >= d.size()) throw std::runtime_error("BMP: truncated palette");
            pal.push_back(px(d[po + 4 * i + 2] / 255.0f, d[po + 4 * i + 1] / 255.0f, d[po + 4 * i] / 255.0f, 1.0f)); }
    }
    if (bpp != 1 && bpp != 4 && bpp != 8 && bpp != 16 && bpp != 24 && bpp != 32) { err = "BMP: unsupported bit depth"; return false; }
    const size_t rowBytes = ((size_t)W * bpp + 31) / 32 * 4;
    if ((size_t)dataOff + rowBytes * H > d.size()) { err = "BMP: truncated pixel data"; return false; }
    img.width = W; img.height = H; img.normalized = true; img.texels.resize((size_t)W * H);
    bool anyAlpha = false;
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t* row = &d[dataOff + rowBytes * (topDown ? y : H - 1 - y)];
        for (uint32_t x = 0; x < W; x++) {
            TbFloat4 t;
            if (bpp <= 8) { const uint32_t per = 8 / bpp, idx = (row[x / per] >> ((per - 1 - x % per) * bpp)) & ((1u << bpp) - 1);
                t = idx < pal.size() ? pal[idx] : px(0, 0, 0, 1); }
            else if (bpp == 24) t = px(row[3 * x + 2] / 255.0f, row[3 * x + 1] / 255.0f, row[3 * x] / 255.0f, 1.0f);
            else { const uint32_t v = bpp == 16 ? ((uint32_t)row[2 * x] | ((uint32_t)row[2 * x + 1] << 8)) : ((uint32_t)row[4 * x] | ((uint32_t)row[4 * x +
                1] << 8) | ((uint32_t)row[4 * x + 2] << 16) | ((uint32_t)row[4 * x + 3] << 24));
                   t = px(maskUnorm(v, rm), maskUnorm(v, gm), maskUnorm(v, bm), am ? maskUnorm(v, am) : 1.0f); }
            if (t.w != 1.0f) anyAlpha = true;
            img.texels[(size_t)y * W + x] = t;
        }
    }
    img.hasAlpha = anyAlpha;
    return true;
}

/* ---- DDS ------------------------------------------------------------------------------------------------------------------- */
float halfToFloat(uint16_t h)
{
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u; uint32_t u;
    if (e == 0) { if (!m) u = s; else { int k = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; k++;
        } u = s | ((uint32_t)(113 - k) << 23) | ((mm & 1023u) << 13); } }
    else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

void bcColours(const uint8_t* b, TbFloat4 c[4], bool bc1) /* the colour half of a BC1 / BC2 / BC3 block */
{
    const uint32_t c0 = (uint32_t)b[0] | ((uint32_t)b[1] << 8), c1 = (uint32_t)b[2] | ((uint32_t)b[3] << 8);
    auto e = [](uint32_t v) { return px((float)(v >> 11) / 31.0f, (float)((v >> 5) & 63u) / 63.0f, (float)(v & 31u) / 31.0f, 1.0f); };
    c[0] = e(c0); c[1] = e(c1);
    auto mix = [](const TbFloat4& a, const TbFloat4& q, float wa, float wq) { return px(a.x * wa + q.x * wq, a.y * wa + q.y * wq, a.z * wa + q.z * wq, 1.0f); };
    if (!bc1 || c0 > c1) { c[2] = mix(c[0], c[1], 2.0f / 3.0f, 1.0f / 3.0f); c[3] = mix(c[0], c[1], 1.0f / 3.0f, 2.0f / 3.0f); }
    else { c[2] = mix(c[0], c[1], 0.5f, 0.5f); c[3] = px(0, 0, 0, 0); }
}
void bcAlpha8(const uint8_t* b, float a[8], bool snorm) /* the interpolated-alpha block of BC3 / BC4 / BC5 */
{
    float a0, a1;
    if (snorm) { auto s = [](uint8_t v) { const int i = (int8_t)v; return i <= -127 ? -1.0f : (float)i / 127.0f; }; a0 = s(b[0]); a1 = s(b[1]); }
    else { a0 = b[0] / 255.0f; a1 = b[1] / 255.0f; }
    a[0] = a0; a[1] = a1;
    const bool eight = snorm ? (int8_t)b[0] > (int8_t)b[1] : b[0] > b[1];
    if (eight) for (int i = 1; i < 7; i++) a[1 + i] = ((float)(7 - i) * a0 + (float)i * a1) / 7.0f;
    else { for (int i = 1; i < 5; i++) a[1 + i] = ((float)(5 - i) * a0 + (float)i * a1) / 5.0f; a[6] = snorm ? -1.0f : 0.0f; a[7] = 1.0f; }
}
uint32_t bcAlphaIndex(const uint8_t* b, int texel) { uint64_t bits = 0; for (int i = 0; i < 6; i++) bits |= (uint64_t)b[2 + i] << (8 * i);
    return (uint32_t)(bits >> (3 * texel)) & 7u; }

/* BC7 (DXGI 97-99): eight block modes, up to three endpoint subsets chosen by a fixed partition shape, 2-4 bit indices of which the
 * "anchor" pixel of every subset stores one bit less (D3D11 functional spec 19.5.9; the tables are the format's constants, regenerated by
 * tests/golden/make_bc7_tables.py from an independent decoder). */
const uint8_t bc7Part2[1024] = {
    0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,0,1,0,0,0,1,0,0,0,1,0,0,0,1,0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1,0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1,
    0,0,0,0,0,0,0,1,0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1,0,1,1,1,1,1,1,1,0,0,0,1,0,0,1,1,0,1,1,1,1,1,1,1,0,0,0,0,0,0,0,1,0,0,1,1,0,1,1,1,
    0,0,0,0,0,0,0,0,0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,1,0,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0,1,0,1,1,1,
    0,0,0,1,0,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,
    0,0,0,0,1,0,0,0,1,1,1,0,1,1,1,1,0,1,1,1,0,0,0,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,0,0,0,1,1,1,0,0,1,1,1,0,0,1,1,0,0,0,1,0,0,0,0,
    0,0,1,1,0,0,0,1,0,0,0,0,0,0,0,0,0,0,0,0,1,0,0,0,1,1,0,0,1,1,1,0,0,0,0,0,0,0,0,0,1,0,0,0,1,1,0,0,0,1,1,1,0,0,1,1,0,0,1,1,0,0,0,1,
    0,0,1,1,0,0,0,1,0,0,0,1,0,0,0,0,0,0,0,0,1,0,0,0,1,0,0,0,1,1,0,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,0,1,1,0,1,1,0,0,1,1,0,1,1,0,0,
    0,0,0,1,0,1,1,1,1,1,1,0,1,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,0,1,1,1,0,0,0,1,1,0,0,0,1,1,1,0,0,0,1,1,1,0,0,1,1,0,0,1,1,1,0,0,
    0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,1,0,1,1,0,1,0,0,1,0,1,1,0,1,0,0,0,1,1,0,0,1,1,1,1,0,0,1,1,0,0,
    0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,1,0,1,0,1,0,1,1,0,1,0,1,0,1,0,0,1,1,0,1,0,0,1,0,1,1,0,1,0,0,1,0,1,0,1,1,0,1,0,1,0,1,0,0,1,0,1,
    0,1,1,1,0,0,1,1,1,1,0,0,1,1,1,0,0,0,0,1,0,0,1,1,1,1,0,0,1,0,0,0,0,0,1,1,0,0,1,0,0,1,0,0,1,1,0,0,0,0,1,1,1,0,1,1,1,1,0,1,1,1,0,0,
    0,1,1,0,1,0,0,1,1,0,0,1,0,1,1,0,0,0,1,1,1,1,0,0,1,1,0,0,0,0,1,1,0,1,1,0,0,1,1,0,1,0,0,1,1,0,0,1,0,0,0,0,0,1,1,0,0,1,1,0,0,0,0,0,
    0,1,0,0,1,1,1,0,0,1,0,0,0,0,0,0,0,0,1,0,0,1,1,1,0,0,1,0,0,0,0,0,0,0,0,0,0,0,1,0,0,1,1,1,0,0,1,0,0,0,0,0,0,1,0,0,1,1,1,0,0,1,0,0,
    0,1,1,0,1,1,0,0,1,0,0,1,0,0,1,1,0,0,1,1,0,1,1,0,1,1,0,0,1,0,0,1,0,1,1,0,0,0,1,1,1,0,0,1,1,1,0,0,0,0,1,1,1,0,0,1,1,1,0,0,0,1,1,0,
    0,1,1,0,1,1,0,0,1,1,0,0,1,0,0,1,0,1,1,0,0,0,1,1,0,0,1,1,1,0,0,1,0,1,1,1,1,1,1,0,1,0,0,0,0,0,0,1,0,0,0,1,1,0,0,0,1,1,1,0,0,1,1,1,
    0,0,0,0,1,1,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,1,1,1,1,0,0,0,0,0,0,1,0,0,0,1,0,1,1,1,0,1,1,1,0,0,1,0,0,0,1,0,0,0,1,1,1,0,1,1,1,
};
const uint8_t bc7Part3[1024] = {
    0,0,1,1,0,0,1,1,0,2,2,1,2,2,2,2,0,0,0,1,0,0,1,1,2,2,1,1,2,2,2,1,0,0,0,0,2,0,0,1,2,2,1,1,2,2,1,1,0,2,2,2,0,0,2,2,0,0,1,1,0,1,1,1,
    0,0,0,0,0,0,0,0,1,1,2,2,1,1,2,2,0,0,1,1,0,0,1,1,0,0,2,2,0,0,2,2,0,0,2,2,0,0,2,2,1,1,1,1,1,1,1,1,0,0,1,1,0,0,1,1,2,2,1,1,2,2,1,1,
    0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,0,0,0,0,1,1,1,1,1,1,1,1,2,2,2,2,0,0,0,0,1,1,1,1,2,2,2,2,2,2,2,2,0,0,1,2,0,0,1,2,0,0,1,2,0,0,1,2,
    0,1,1,2,0,1,1,2,0,1,1,2,0,1,1,2,0,1,2,2,0,1,2,2,0,1,2,2,0,1,2,2,0,0,1,1,0,1,1,2,1,1,2,2,1,2,2,2,0,0,1,1,2,0,0,1,2,2,0,0,2,2,2,0,
    0,0,0,1,0,0,1,1,0,1,1,2,1,1,2,2,0,1,1,1,0,0,1,1,2,0,0,1,2,2,0,0,0,0,0,0,1,1,2,2,1,1,2,2,1,1,2,2,0,0,2,2,0,0,2,2,0,0,2,2,1,1,1,1,
    0,1,1,1,0,1,1,1,0,2,2,2,0,2,2,2,0,0,0,1,0,0,0,1,2,2,2,1,2,2,2,1,0,0,0,0,0,0,1,1,0,1,2,2,0,1,2,2,0,0,0,0,1,1,0,0,2,2,1,0,2,2,1,0,
    0,1,2,2,0,1,2,2,0,0,1,1,0,0,0,0,0,0,1,2,0,0,1,2,1,1,2,2,2,2,2,2,0,1,1,0,1,2,2,1,1,2,2,1,0,1,1,0,0,0,0,0,0,1,1,0,1,2,2,1,1,2,2,1,
    0,0,2,2,1,1,0,2,1,1,0,2,0,0,2,2,0,1,1,0,0,1,1,0,2,0,0,2,2,2,2,2,0,0,1,1,0,1,2,2,0,1,2,2,0,0,1,1,0,0,0,0,2,0,0,0,2,2,1,1,2,2,2,1,
    0,0,0,0,0,0,0,2,1,1,2,2,1,2,2,2,0,2,2,2,0,0,2,2,0,0,1,2,0,0,1,1,0,0,1,1,0,0,1,2,0,0,2,2,0,2,2,2,0,1,2,0,0,1,2,0,0,1,2,0,0,1,2,0,
    0,0,0,0,1,1,1,1,2,2,2,2,0,0,0,0,0,1,2,0,1,2,0,1,2,0,1,2,0,1,2,0,0,1,2,0,2,0,1,2,1,2,0,1,0,1,2,0,0,0,1,1,2,2,0,0,1,1,2,2,0,0,1,1,
    0,0,1,1,1,1,2,2,2,2,0,0,0,0,1,1,0,1,0,1,0,1,0,1,2,2,2,2,2,2,2,2,0,0,0,0,0,0,0,0,2,1,2,1,2,1,2,1,0,0,2,2,1,1,2,2,0,0,2,2,1,1,2,2,
    0,0,2,2,0,0,1,1,0,0,2,2,0,0,1,1,0,2,2,0,1,2,2,1,0,2,2,0,1,2,2,1,0,1,0,1,2,2,2,2,2,2,2,2,0,1,0,1,0,0,0,0,2,1,2,1,2,1,2,1,2,1,2,1,
    0,1,0,1,0,1,0,1,0,1,0,1,2,2,2,2,0,2,2,2,0,1,1,1,0,2,2,2,0,1,1,1,0,0,0,2,1,1,1,2,0,0,0,2,1,1,1,2,0,0,0,0,2,1,1,2,2,1,1,2,2,1,1,2,
    0,2,2,2,0,1,1,1,0,1,1,1,0,2,2,2,0,0,0,2,1,1,1,2,1,1,1,2,0,0,0,2,0,1,1,0,0,1,1,0,0,1,1,0,2,2,2,2,0,0,0,0,0,0,0,0,2,1,1,2,2,1,1,2,
    0,1,1,0,0,1,1,0,2,2,2,2,2,2,2,2,0,0,2,2,0,0,1,1,0,0,1,1,0,0,2,2,0,0,2,2,1,1,2,2,1,1,2,2,0,0,2,2,0,0,0,0,0,0,0,0,0,0,0,0,2,1,1,2,
    0,0,0,2,0,0,0,1,0,0,0,2,0,0,0,1,0,2,2,2,1,2,2,2,0,2,2,2,1,2,2,2,0,1,0,1,2,2,2,2,2,2,2,2,2,2,2,2,0,1,1,1,2,0,1,1,2,2,0,1,2,2,2,0,
};
const uint8_t bc7Anchor2[64] = {
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,2,8,2,2,8,8,15,2,8,2,2,8,8,2,2,
    15,15,6,8,2,8,15,15,2,8,2,2,2,15,15,6,6,2,6,8,15,15,2,2,15,15,15,15,15,2,2,15,
};
const uint8_t bc7Anchor3a[64] = {
    3,3,15,15,8,3,15,15,8,8,6,6,6,5,3,3,3,3,8,15,3,3,6,10,5,8,8,6,8,5,15,15,
    8,15,3,5,6,10,8,15,15,3,15,5,15,15,15,15,3,15,5,5,5,8,5,10,5,10,8,13,15,12,3,3,
};
const uint8_t bc7Anchor3b[64] = {
    15,8,8,3,15,15,3,8,15,15,15,15,15,15,15,8,15,8,15,3,15,8,15,8,3,15,6,10,15,15,10,8,
    15,3,15,10,10,8,9,10,6,15,8,15,3,6,6,8,15,3,15,15,15,15,15,15,15,15,15,15,3,15,15,8,
};

struct Bc7Mode { uint8_t subsets, partBits, rotBits, idxSelBit, colourBits, alphaBits, endpointP, sharedP, idxBits, idx2Bits; };
const Bc7Mode bc7Modes[8] = {{3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
                             {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0}};
const uint8_t bc7Weights2[4] = {0, 21, 43, 64}, bc7Weights3[8] = {0, 9, 18, 27, 37, 46, 55, 64}, bc7Weights4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43,
    47, 51, 55, 60, 64};

void bc7Block(const uint8_t* b, TbFloat4 texel[16])
{
    uint32_t pos = 0;
    auto get = [&](uint32_t n) { uint32_t v = 0; for (uint32_t i = 0; i < n; i++, pos++) v |= (uint32_t)((b[pos >> 3] >> (pos & 7)) & 1u) << i; return v; };
    uint32_t mode = 0; while (mode < 8 && !get(1)) mode++;
    if (mode == 8) { for (int t = 0; t < 16; t++) texel[t] = px(0, 0, 0, 0); return; }      /* reserved: the hardware returns zeros */
    const Bc7Mode& m = bc7Modes[mode];
    const uint32_t partition = get(m.partBits), rotation = get(m.rotBits), idxSel = get(m.idxSelBit), numEp = 2u * m.subsets;
    uint32_t ep[6][4];
    for (int c = 0; c < 3; c++) for (uint32_t e = 0; e < numEp; e++) ep[e][c] = get(m.colourBits);
    for (uint32_t e = 0; e < numEp; e++) ep[e][3] = m.alphaBits ? get(m.alphaBits) : 255u;
    uint32_t cb = m.colourBits, ab = m.alphaBits;
    if (m.endpointP || m.sharedP) {
        uint32_t pb[6]; if (m.endpointP) for (uint32_t e = 0; e < numEp; e++) pb[e] = get(1);
            else for (uint32_t k = 0; k < m.subsets; k++) pb[2 * k] = pb[2 * k + 1] = get(1);
        for (uint32_t e = 0; e < numEp; e++) { for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | pb[e]; if (ab) ep[e][3] = (ep[e][3] << 1) | pb[e]; }
        cb++; if (ab) ab++;
    }
    for (uint32_t e = 0; e < numEp; e++) {   /* to 8 bits: left-align, replicate the top bits below */
        for (int c = 0; c < 3; c++) { const uint32_t v = ep[e][c] << (8 - cb); ep[e][c] = v | (v >> cb); }
        if (ab) { const uint32_t v = ep[e][3] << (8 - ab); ep[e][3] = v | (v >> ab); }
    }
    uint8_t subset[16]; uint32_t anchor[3] = {0, 0, 0};
    for (int t = 0; t < 16; t++) subset[t] = m.subsets == 1 ? 0 : (m.subsets == 2 ? bc7Part2[partition * 16 + t] : bc7Part3[partition * 16 + t]);
    if (m.subsets == 2) anchor[1] = bc7Anchor2[partition]; else if (m.subsets == 3) { anchor[1] = bc7Anchor3a[partition]; anchor[2] = bc7Anchor3b[partition]; }
    uint32_t idx[16], idx2[16];
    for (uint32_t t = 0; t < 16; t++) idx[t] = get(t == anchor[subset[t]] ? m.idxBits - 1u : m.idxBits);
    for (uint32_t t = 0; t < 16; t++) idx2[t] = m.idx2Bits ? get(t == 0 ? m.idx2Bits - 1u : m.idx2Bits) : 0;
    auto weight = [](uint32_t bits, uint32_t i) { return (uint32_t)(bits == 2 ? bc7Weights2[i] : (bits == 3 ? bc7Weights3[i] : bc7Weights4[i])); };
    for (uint32_t t = 0; t < 16; t++) {
        const uint32_t* e0 = ep[2 * subset[t]]; const uint32_t* e1 = ep[2 * subset[t] + 1];
        uint32_t wc, wa;
        if (!m.idx2Bits) wc = wa = weight(m.idxBits, idx[t]);
        else if (idxSel) { wc = weight(m.idx2Bits, idx2[t]); wa = weight(m.idxBits, idx[t]); }
        else { wc = weight(m.idxBits, idx[t]); wa = weight(m.idx2Bits, idx2[t]); }
        uint32_t v[4];
        for (int c = 0; c < 3; c++) v[c] = ((64u - wc) * e0[c] + wc * e1[c] + 32u) >> 6;
        v[3] = m.alphaBits ? ((64u - wa) * e0[3] + wa * e1[3] + 32u) >> 6 : 255u;
        if (rotation) { const uint32_t k = rotation - 1u, tmp = v[k]; v[k] = v[3]; v[3] = tmp; }
        texel[t] = px((float)v[0] / 255.0f, (float)v[1] / 255.0f, (float)v[2] / 255.0f, (float)v[3] / 255.0f);
    }
}

bool decodeDds(const std::vector<uint8_t>& d, DecodedImage& img, std::string& err)
{
    if (d.size() < 128 || memcmp(d.data(), "DDS ", 4) || rd32(d, 4) != 124) { err = "not a DDS file"; return false; }
    const uint32_t H = rd32(d, 12), W = rd32(d, 16), pfFlags = rd32(d, 80), fourcc = rd32(d, 84), bits = rd32(d, 88);
    const uint32_t rm = rd32(d, 92), gm = rd32(d, 96), bm = rd32(d, 100), am = rd32(d, 104);
    if (!ImageDimensionsOk(W, H)) { err = "DDS: bad dimensions (a 2-D texture has at most 16384 texels a side)"; return false; }
    size_t off = 128; uint32_t dxgi = 0;
    auto cc = [](const char* s) { return (uint32_t)(uint8_t)s[0] | ((uint32_t)(uint8_t)s[1] << 8) | ((uint32_t)(uint8_t)s[2] << 16) |
        ((uint32_t)(uint8_t)s[3] << 24); };
    enum { RAW, BC1, BC2, BC3, BC4, BC5, BC4S, BC5S, BC7, F16, F32, F32R, U16 } kind = RAW;
    if (pfFlags & 4u) {
        if (fourcc == cc("DX10")) { if (d.size() < 148) { err = "DDS: truncated DX10 header"; return false; } dxgi = rd32(d, 128); off = 148; }
        else if (fourcc == cc("DXT1")) kind = BC1; else if (fourcc == cc("DXT2") || fourcc == cc("DXT3")) kind = BC2;
            else if (fourcc == cc("DXT4") || fourcc == cc("DXT5")) kind = BC3;
        else if (fourcc == cc("ATI1") || fourcc == cc("BC4U")) kind = BC4; else if (fourcc == cc("BC4S")) kind = BC4S;
            else if (fourcc == cc("ATI2") || fourcc == cc("BC5U")) kind = BC5; else if (fourcc == cc("BC5S")) kind = BC5S;
        else if (fourcc == 113) kind = F16; else if (fourcc == 116) kind = F32; else if (fourcc == 114) kind = F32R; else if (fourcc == 36) kind = U16;
        else { err = "DDS: unsupported FourCC"; return false; }
    }
    uint32_t m[4] = {rm, gm, bm, (pfFlags & 1u) ? am : 0u}; uint32_t rawBits = bits;
        bool lum = (pfFlags & 0x20000u) != 0, alphaOnly = (pfFlags & 2u) != 0 && !(pfFlags & 0x40u) && !lum;
    if (dxgi) {
        switch (dxgi) {
        case 71: case 72: kind = BC1; break; case 74: case 75: kind = BC2; break; case 77: case 78: kind = BC3; break; case 80: kind = BC4; break;
            case 81: kind = BC4S; break;
        case 83: kind = BC5; break; case 84: kind = BC5S; break; case 97: case 98: case 99: kind = BC7; break; case 10: kind = F16; break; case 2: kind = F32;
            break; case 41: kind = F32R; break; case 11: kind = U16; break;
        case 28: case 29: kind = RAW; rawBits = 32; m[0] = 0xff; m[1] = 0xff00; m[2] = 0xff0000; m[3] = 0xff000000u; break;          /* R8G8B8A8_UNORM(_SRGB) */
        case 87: case 91: kind = RAW; rawBits = 32; m[0] = 0xff0000; m[1] = 0xff00; m[2] = 0xff; m[3] = 0xff000000u; break;          /* B8G8R8A8 */
        case 88: case 93: kind = RAW; rawBits = 32; m[0] = 0xff0000; m[1] = 0xff00; m[2] = 0xff; m[3] = 0; break;                    /* B8G8R8X8 */
        case 61: kind = RAW; rawBits = 8; m[0] = 0xff; m[1] = m[2] = m[3] = 0; lum = false; break;                                  /* R8_UNORM */
        case 85: kind = RAW; rawBits = 16; m[0] = 0xf800; m[1] = 0x07e0; m[2] = 0x001f; m[3] = 0; break;                           /* B5G6R5 */
        case 86: kind = RAW; rawBits = 16; m[0] = 0x7c00; m[1] = 0x03e0; m[2] = 0x001f; m[3] = 0x8000; break;                      /* B5G5R5A1 */
        default: err = "DDS: unsupported DXGI format " + std::to_string(dxgi); return false;
        }
    }
    bool sized = false;
    auto need = [&](size_t bytes) { if (off + bytes > d.size()) throw std::runtime_error("DDS: truncated surface");
                                    /* nothing of the header's size is allocated before the surface is known to be there */
                                    if (!sized) { img.texels.assign((size_t)W * H, px(0, 0, 0, 1)); sized = true; } };
    img.width = W; img.height = H; img.normalized = !(kind == F16 || kind == F32 || kind == F32R);
    if (kind == RAW) {
        if (rawBits != 8 && rawBits != 16 && rawBits != 24 && rawBits != 32) { err = "DDS: unsupported bit count"; return false; }
        const size_t bpp = rawBits / 8; need((size_t)W * H * bpp);
        for (size_t i = 0; i < (size_t)W * H; i++) {
            uint32_t v = 0; for (size_t k = 0; k < bpp; k++) v |= (uint32_t)d[off + i * bpp + k] << (8 * k);
            TbFloat4 t;
            if (alphaOnly) t = px(0, 0, 0, maskUnorm(v, am ? am : 0xffu));
            /* L8 / A8L8: DirectXTex expands luminance to grey RGB */
            else if (lum) { const float l = maskUnorm(v, m[0]); t = px(l, l, l, m[3] ? maskUnorm(v, m[3]) : 1.0f); }
            else if (!m[1] && !m[2] && dxgi == 61) t = px(maskUnorm(v, m[0]), 0.0f, 0.0f, 1.0f);
            else t = px(maskUnorm(v, m[0]), maskUnorm(v, m[1]), maskUnorm(v, m[2]), m[3] ? maskUnorm(v, m[3]) : 1.0f);
            img.texels[i] = t;
        }
    } else if (kind == F16 || kind == F32 || kind == F32R || kind == U16) {
        const size_t bpp = kind == F32 ? 16 : (kind == F32R ? 4 : 8); need((size_t)W * H * bpp);
        for (size_t i = 0; i < (size_t)W * H; i++) {
            const size_t o = off + i * bpp; float f[4] = {0, 0, 0, 1};
            if (kind == F32) memcpy(f, &d[o], 16); else if (kind == F32R) memcpy(f, &d[o], 4);
            else for (int k = 0; k < 4; k++) { const uint16_t h = (uint16_t)rd16(d, o + 2 * (size_t)k);
                f[k] = kind == F16 ? halfToFloat(h) : (float)h / 65535.0f; }
            img.texels[i] = px(f[0], f[1], f[2], f[3]);
        }
    } else {
        const size_t blockBytes = (kind == BC1 || kind == BC4 || kind == BC4S) ? 8 : 16; const uint32_t bw = (W + 3) / 4, bh = (H + 3) / 4;
        need((size_t)bw * bh * blockBytes);
        for (uint32_t by = 0; by < bh; by++) for (uint32_t bx = 0; bx < bw; bx++) {
            const uint8_t* b = &d[off + ((size_t)by * bw + bx) * blockBytes];
            TbFloat4 texel[16];
            if (kind == BC1 || kind == BC2 || kind == BC3) {
                const uint8_t* cb = kind == BC1 ? b : b + 8; TbFloat4 c[4]; bcColours(cb, c, kind == BC1);
                const uint32_t idx = (uint32_t)cb[4] | ((uint32_t)cb[5] << 8) | ((uint32_t)cb[6] << 16) | ((uint32_t)cb[7] << 24);
                float a8[8]; if (kind == BC3) bcAlpha8(b, a8, false);
                for (int t = 0; t < 16; t++) { texel[t] = c[(idx >> (2 * t)) & 3u];
                    if (kind == BC2) texel[t].w = (float)((b[t / 2] >> (4 * (t & 1))) & 15) / 15.0f; else if (kind == BC3) texel[t].w = a8[bcAlphaIndex(b, t)];
                        }
            } else if (kind == BC7) bc7Block(b, texel);
            else {
                const bool sn = kind == BC4S || kind == BC5S; float r8[8], g8[8]; bcAlpha8(b, r8, sn); if (kind == BC5 || kind == BC5S) bcAlpha8(b + 8, g8, sn);
                for (int t = 0; t < 16; t++) texel[t] = px(r8[bcAlphaIndex(b, t)], (kind == BC5 || kind == BC5S) ? g8[bcAlphaIndex(b + 8, t)] : 0.0f, 0.0f,
                    1.0f);
            }
            for (int t = 0; t < 16; t++) { const uint32_t x = bx * 4 + (uint32_t)(t & 3), y = by * 4 + (uint32_t)(t >> 2);
                if (x < W && y < H) img.texels[(size_t)y * W + x] = texel[t]; }
        }
    }
    bool anyAlpha = false; for (const TbFloat4& t : img.texels) if (t.w != 1.0f) { anyAlpha = true; break; }
    img.hasAlpha = anyAlpha;
    return true;
}

} // namespace

bool DecodeJpegBmpDds(const std::string& file, const std::vector<uint8_t>& d, int kind, DecodedImage& img, std::string& err)
{
    try { return kind == 0 ? decodeJpeg(d, img, err) : (kind == 1 ? decodeBmp(d, img, err) : decodeDds(d, img, err)); }
    catch (const std::exception& e) { err = std::string(e.what()) + " ('" + file + "')"; return false; }
}

} // namespace tbhost
