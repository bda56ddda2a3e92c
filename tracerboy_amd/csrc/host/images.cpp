/* images.cpp -- image decode for environment maps and image textures: Radiance RGBE (.hdr, flat
 * and new-style RLE scanlines) and PFM.  Replaces DirectXTex's LoadFromHDRFile as used by
 * TracerBoy::InitializeTexture (/root/reference/TracerBoy/TracerBoy.cpp:2188-2255) for the formats
 * the reference scenes ship; output is RGBA32F, top row first, alpha 1.
 * RGBE decode: value = mantissa * 2^(e - 136)  (Ward's format; e == 0 -> 0).
 */
#include "host_scene.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace tbhost {

namespace {

bool endsWith(const std::string& s, const char* suf)
{
    size_t n = strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) if (tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

bool loadHdr(const std::string& file, std::vector<TbFloat4>& texels, uint32_t& W, uint32_t& H, std::string& err)
{
    std::ifstream in(file, std::ios::binary);
    if (!in) { err = "could not open image '" + file + "'"; return false; }
    std::string line;
    std::getline(in, line);
    if (line.substr(0, 2) != "#?") { err = file + ": not a Radiance HDR file"; return false; }
    bool rgbe = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break;
        if (line.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) rgbe = true;
    }
    (void)rgbe;
    std::getline(in, line);
    int h = 0, w = 0; char sy = 0, sx = 0, ay = 0, ax = 0;
    if (sscanf(line.c_str(), "%c%c %d %c%c %d", &sy, &ay, &h, &sx, &ax,
        &w) != 6 || ay != 'Y' || ax != 'X' || w <= 0 || h <= 0) { err = file + ": unsupported HDR resolution line '" + line + "'"; return false; }
    W = (uint32_t)w; H = (uint32_t)h;
    if (!ImageDimensionsOk(W, H)) { err = file + ": HDR dimensions beyond the 16384 a 2-D texture can have"; return false; }
    {   /* a run-length packet holds at most 127 values of one channel in 2 bytes: refuse a header the rest of the file cannot fill */
        const std::streampos here = in.tellg(); in.seekg(0, std::ios::end); const uint64_t left = (uint64_t)(in.tellg() - here); in.seekg(here);
        if ((uint64_t)W * H * 4 > 64 * left) { err = file + ": truncated HDR data"; return false; }
    }
    texels.assign((size_t)W * H, TbFloat4{0, 0, 0, 1});
    std::vector<unsigned char> scan((size_t)W * 4);
    for (uint32_t y = 0; y < H; y++) {
        unsigned char hd[4];
        in.read((char*)hd, 4);
        if (!in) { err = file + ": truncated HDR data"; return false; }
        if (hd[0] == 2 && hd[1] == 2 && !(hd[2] & 0x80) && W >= 8 && W < 32768 && (uint32_t)((hd[2] << 8) | hd[3]) == W) {
            for (int c = 0; c < 4; c++) {
                uint32_t x = 0;
                while (x < W) {
                    unsigned char cnt; in.read((char*)&cnt, 1);
                    if (!in) { err = file + ": truncated HDR data"; return false; }
                    if (cnt > 128) { unsigned char v; in.read((char*)&v, 1); uint32_t n = cnt - 128u; if (x + n > W) { err = file + ": bad HDR run";
                        return false; } for (uint32_t k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = v; }
                    else { uint32_t n = cnt; if (n == 0 || x + n > W) { err = file + ": bad HDR run"; return false;
                        } for (uint32_t k = 0; k < n; k++) { unsigned char v; in.read((char*)&v, 1); scan[(size_t)(x++) * 4 + c] = v; } }
                }
            }
        } else {
            memcpy(scan.data(), hd, 4);
            in.read((char*)scan.data() + 4, (std::streamsize)((size_t)W * 4 - 4));
            if (!in) { err = file + ": truncated HDR data"; return false; }
        }
        uint32_t row = (sy == '-') ? y : (H - 1 - y);
        for (uint32_t x = 0; x < W; x++) {
            const unsigned char* p = &scan[(size_t)x * 4];
            TbFloat4 t = {0, 0, 0, 1};
            if (p[3]) { float f = ldexpf(1.0f, (int)p[3] - 136); t.x = p[0] * f; t.y = p[1] * f; t.z = p[2] * f; }
            uint32_t col = (sx == '+') ? x : (W - 1 - x);
            texels[(size_t)row * W + col] = t;
        }
    }
    return true;
}

bool loadPfm(const std::string& file, std::vector<TbFloat4>& texels, uint32_t& W, uint32_t& H, std::string& err)
{
    std::ifstream in(file, std::ios::binary);
    if (!in) { err = "could not open image '" + file + "'"; return false; }
    std::string magic; int w = 0, h = 0; float scale = 0;
    in >> magic >> w >> h >> scale;
    in.get();
    int ch = magic == "PF" ? 3 : (magic == "Pf" ? 1 : 0);
    if (!ch || w <= 0 || h <= 0) { err = file + ": not a PFM file"; return false; }
    W = (uint32_t)w; H = (uint32_t)h;
    if (!ImageDimensionsOk(W, H)) { err = file + ": PFM dimensions beyond the 16384 a 2-D texture can have"; return false; }
    {
        const std::streampos here = in.tellg(); in.seekg(0, std::ios::end); const uint64_t left = here < 0 ? 0 : (uint64_t)(in.tellg() - here); in.seekg(here);
        if ((uint64_t)W * H * (uint64_t)ch * 4 > left) { err = file + ": truncated PFM data"; return false; }
    }
    std::vector<float> raw((size_t)W * H * ch);
    in.read((char*)raw.data(), (std::streamsize)(raw.size() * 4));
    if (!in) { err = file + ": truncated PFM data"; return false; }
    if (scale > 0) { err = file + ": big-endian PFM not supported"; return false; }
    texels.resize((size_t)W * H);
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
        const float* p = &raw[((size_t)(H - 1 - y) * W + x) * ch]; /* PFM stores bottom row first */
        TbFloat4 t = {p[0], ch == 3 ? p[1] : p[0], ch == 3 ? p[2] : p[0], 1.0f};
        texels[(size_t)y * W + x] = t;
    }
    return true;
}

} // namespace

bool LoadImageRGBA32F(const std::string& file, std::vector<TbFloat4>& texels, uint32_t& w, uint32_t& h, bool& normalizedFormat, std::string& err,
    bool* hasAlpha)
{
    normalizedFormat = false;
    if (hasAlpha) *hasAlpha = false;
    if (endsWith(file, ".hdr")) return loadHdr(file, texels, w, h, err);
    if (endsWith(file, ".pfm")) return loadPfm(file, texels, w, h, err);
    DecodedImage img; /* .png / .tga (TracerBoy.cpp:2216-2227) */
    if (!DecodeImageFile(file, img, err)) return false;
    texels.swap(img.texels); w = img.width; h = img.height; normalizedFormat = img.normalized;
    if (hasAlpha) *hasAlpha = img.hasAlpha;
    return true;
}

} // namespace tbhost

/* ---- blue-noise tiles ---------------------------------------------------------------------------
 * TracerBoy binds two 256x256 RGBA8 tiles at t14/t15 (TracerBoy.cpp:2126-2134, files
 * TracerBoy/Textures/LDR_RGBA_{0,1}.png).  The build ships them as raw bytes in
 * tracerboy_amd/data/ next to the library; TB_DATA_DIR overrides the location. */
#include <dlfcn.h>

namespace tbhost {

static std::string dataDir()
{
    if (const char* e = getenv("TB_DATA_DIR")) return std::string(e);
    Dl_info info;
    if (dladdr((const void*)&dataDir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.find_last_of('/');
        return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/data";
    }
    return "data";
}

bool LoadBlueNoiseTiles(HostScene& scene)
{
    std::vector<TbFloat4>* dst[2] = {&scene.blueNoise0, &scene.blueNoise1};
    for (int i = 0; i < 2; i++) {
        std::ifstream in(dataDir() + "/bluenoise" + std::to_string(i) + ".rgba8", std::ios::binary);
        std::vector<unsigned char> raw(256 * 256 * 4);
        if (!in || !in.read((char*)raw.data(), (std::streamsize)raw.size())) { scene.blueNoise0.clear(); scene.blueNoise1.clear(); return false; }
        dst[i]->resize(256 * 256);
        for (size_t p = 0; p < 256 * 256; p++) (*dst[i])[p] = TbFloat4{raw[4 * p] / 255.0f, raw[4 * p + 1] / 255.0f, raw[4 * p + 2] / 255.0f,
            raw[4 * p + 3] / 255.0f};
    }
    return true;
}


/* ---- image writers for the headless output stage (tb_write_image_*) ------------------------------------------------ */
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n)
{
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c;
        } init = true; }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put32be(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x); }
void pngChunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data)
{
    put32be(out, (uint32_t)data.size());
    size_t at = out.size();
    out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
    put32be(out, crc32_update(0xffffffffu, out.data() + at, out.size() - at) ^ 0xffffffffu);
}
} // namespace

/* 8-bit RGBA PNG; the zlib stream uses stored (uncompressed) deflate blocks -- a valid PNG every decoder reads */
bool WritePngRGBA8(const std::string& file, uint32_t W, uint32_t H, const uint8_t* rgba, std::string& err)
{
    std::vector<uint8_t> raw; raw.reserve((size_t)H * (W * 4 + 1));
    for (uint32_t y = 0; y < H; y++) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * W * 4, rgba + (size_t)(y + 1) * W * 4); } /* filter 0 */
    std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw.size(); i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    for (size_t at = 0; at < raw.size() || at == 0; at += 65535) {
        size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + (long)at, raw.begin() + (long)(at + n));
        if (raw.empty()) break;
    }
    put32be(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr; put32be(ihdr, W); put32be(ihdr, H); ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0);
        ihdr.push_back(0);
    pngChunk(out, "IHDR", ihdr); pngChunk(out, "IDAT", z); pngChunk(out, "IEND", {});
    FILE* f = fopen(file.c_str(), "wb");
    if (!f) { err = "cannot open " + file + " for writing"; return false; }
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short write to " + file;
    return ok;
}

/* PFM: "PF", little-endian RGB floats, rows bottom-up */
bool WritePfmRGB(const std::string& file, uint32_t W, uint32_t H, const float* rgba, std::string& err)
{
    FILE* f = fopen(file.c_str(), "wb");
    if (!f) { err = "cannot open " + file + " for writing"; return false; }
    fprintf(f, "PF\n%u %u\n-1.0\n", W, H);
    std::vector<float> row((size_t)W * 3);
    bool ok = true;
    for (uint32_t y = 0; y < H && ok; y++) {
        const float* src = rgba + (size_t)(H - 1 - y) * W * 4;
        for (uint32_t x = 0; x < W; x++) { row[x * 3] = src[x * 4]; row[x * 3 + 1] = src[x * 4 + 1]; row[x * 3 + 2] = src[x * 4 + 2]; }
        ok = fwrite(row.data(), 4, row.size(), f) == row.size();
    }
    fclose(f);
    if (!ok) err = "short write to " + file;
    return ok;
}

/* OpenEXR, the simplest conforming layout: single part, scan lines, no compression, four FLOAT channels (stored in the
 * alphabetical order the format requires: A, B, G, R), increasing y.  Header = attributes "name\0type\0size value",
 * then one 64-bit file offset per scan line, then per line {y, byte count, channel-planar pixels}. */
bool WriteExrRGBA(const std::string& file, uint32_t W, uint32_t H, const float* rgba, std::string& err)
{
    std::vector<uint8_t> hd;
    auto bytes = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; hd.insert(hd.end(), b, b + n); };
    auto str = [&](const char* t) { bytes(t, strlen(t) + 1); };
    auto i32 = [&](int32_t v) { bytes(&v, 4); };
    auto f32 = [&](float v) { bytes(&v, 4); };
    auto attr = [&](const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); };
    const uint32_t magic = 20000630u, version = 2u;
    bytes(&magic, 4); bytes(&version, 4);
    attr("channels", "chlist", 4 * 18 + 1);
    for (const char* ch : {"A", "B", "G", "R"}) { str(ch); i32(2 /* FLOAT */); const uint8_t lin[4] = {0, 0, 0, 0}; bytes(lin, 4); i32(1); i32(1); }
    hd.push_back(0);
    attr("compression", "compression", 1); hd.push_back(0);
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32((int32_t)W - 1); i32((int32_t)H - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32((int32_t)W - 1); i32((int32_t)H - 1);
    attr("lineOrder", "lineOrder", 1); hd.push_back(0);
    attr("pixelAspectRatio", "float", 4); f32(1.0f);
    attr("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
    attr("screenWindowWidth", "float", 4); f32(1.0f);
    hd.push_back(0);
    FILE* f = fopen(file.c_str(), "wb");
    if (!f) { err = "cannot open " + file + " for writing"; return false; }
    const uint64_t lineBytes = 8 + 16ull * W, first = hd.size() + 8ull * H;
    std::vector<uint64_t> table(H);
    for (uint32_t y = 0; y < H; y++) table[y] = first + lineBytes * y;
    bool ok = fwrite(hd.data(), 1, hd.size(), f) == hd.size() && fwrite(table.data(), 8, H, f) == H;
    std::vector<float> line((size_t)W * 4);
    for (uint32_t y = 0; y < H && ok; y++) {
        const float* src = rgba + (size_t)y * W * 4;
        for (uint32_t x = 0; x < W; x++) { line[x] = src[x * 4 + 3]; line[W + x] = src[x * 4 + 2]; line[2 * (size_t)W + x] = src[x * 4 + 1];
            line[3 * (size_t)W + x] = src[x * 4]; }
        const int32_t head[2] = {(int32_t)y, (int32_t)(16u * W)};
        ok = fwrite(head, 4, 2, f) == 2 && fwrite(line.data(), 4, line.size(), f) == line.size();
    }
    fclose(f);
    if (!ok) err = "short write to " + file;
    return ok;
}

} // namespace tbhost
