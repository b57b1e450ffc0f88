// Host-side I/O around the render path: error text, image decoding for textures,
// pass normalisation to 8 bit and image writers.
//
// Reference behaviour restated here:
//   normalize_* / gamma_correct / color_value   src/buffer.cuh:202-318, src/vec3.cuh:491-493
//   write_image                                 src/image.cuh:100-118
//   image texture load (8-bit, HDR tone-mapped) src/texture.cuh:143-158 (stb_image)
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "crt_internal.h"

namespace {

thread_local std::string g_last_error;
std::mutex g_loader_mutex;
crt_image_loader_fn g_loader = nullptr;
void* g_loader_user = nullptr;


bool read_file(const std::string& path, std::vector<uint8_t>& data) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    data.resize((size_t)n);
    if (n > 0) f.read(reinterpret_cast<char*>(data.data()), n);
    return (bool)f;
}

// ------------------------------------------------------------------ PNG decode ----------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}
// Largest image side the decoders accept (stb_image's STBI_MAX_DIMENSIONS): a header that claims
// more is rejected before anything is allocated.
constexpr int kMaxImageSide = 1 << 24;
bool plausible_size(long long w, long long h) {
    return w > 0 && h > 0 && w <= kMaxImageSide && h <= kMaxImageSide && w * h <= (1ll << 30);
}

bool decode_png(const std::vector<uint8_t>& file, int& w, int& h, int& ch, std::vector<uint8_t>& out, std::string& err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    size_t pos = 8;
    int bit_depth = 0, color_type = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool have_ihdr = false;
    while (pos + 8 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + len > file.size()) { err = "truncated PNG"; return false; }
        const uint8_t* body = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad PNG header"; return false; }
            if (!plausible_size(be32(body), be32(body + 4))) { err = "PNG dimensions out of range"; return false; }
            w = (int)be32(body); h = (int)be32(body + 4);
            bit_depth = body[8]; color_type = body[9]; interlace = body[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (!have_ihdr || w <= 0 || h <= 0) { err = "bad PNG header"; return false; }
    if (interlace > 1) { err = "bad PNG interlace method"; return false; }
    if (bit_depth != 1 && bit_depth != 2 && bit_depth != 4 && bit_depth != 8 && bit_depth != 16) { err = "bad PNG bit depth"; return false; }
    const int src_ch = color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 3 ? 1 : color_type == 4 ? 2 : color_type == 6 ? 4 : 0;
    if (!src_ch || (color_type == 3 && bit_depth > 8) || (bit_depth < 8 && color_type != 0 && color_type != 3)) { err = "PNG colour type not supported"; return false; }
    // inflate everything (the size is known: filter byte + packed row per scanline of each pass)
    static const int px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int passes = interlace ? 7 : 1;
    auto pass_dim = [&](int p, int& pw, int& ph) {
        if (!interlace) { pw = w; ph = h; return; }
        pw = (w - px0[p] + pdx[p] - 1) / pdx[p];
        ph = (h - py0[p] + pdy[p] - 1) / pdy[p];
    };
    const size_t bits_pp = (size_t)src_ch * bit_depth;
    size_t total = 0;
    for (int p = 0; p < passes; p++) {
        int pw, ph;
        pass_dim(p, pw, ph);
        if (pw > 0 && ph > 0) total += (((size_t)pw * bits_pp + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw(total);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { err = "PNG inflate failed"; return false; }
    // samples at full width: one uint16 per channel sample (palette indices and grey levels unscaled)
    std::vector<uint16_t> samples((size_t)w * h * src_ch);
    const size_t fbpp = bits_pp >= 8 ? bits_pp / 8 : 1;              // filter distance in bytes
    size_t rpos = 0;
    for (int p = 0; p < passes; p++) {
        int pw, ph;
        pass_dim(p, pw, ph);
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (int y = 0; y < ph; y++) {
            const uint8_t filter = raw[rpos++];
            const uint8_t* in = &raw[rpos];
            rpos += stride;
            for (size_t x = 0; x < stride; x++) {
                int a = x >= fbpp ? cur[x - fbpp] : 0, b = prev[x], c = x >= fbpp ? prev[x - fbpp] : 0;
                int v = in[x];
                switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "bad PNG filter"; return false;
                }
                cur[x] = (uint8_t)v;
            }
            const int oy = interlace ? py0[p] + y * pdy[p] : y;
            for (int x = 0; x < pw; x++) {
                const int ox = interlace ? px0[p] + x * pdx[p] : x;
                uint16_t* dst = &samples[((size_t)oy * w + ox) * src_ch];
                for (int k = 0; k < src_ch; k++) {
                    const size_t idx = (size_t)x * src_ch + k;
                    if (bit_depth == 16) dst[k] = (uint16_t)((cur[idx * 2] << 8) | cur[idx * 2 + 1]);
                    else if (bit_depth == 8) dst[k] = cur[idx];
                    else {
                        const size_t bit = idx * bit_depth;
                        dst[k] = (uint16_t)((cur[bit >> 3] >> (8 - bit_depth - (bit & 7))) & ((1 << bit_depth) - 1));
                    }
                }
            }
            prev.swap(cur);
        }
    }
    // to 8 bit: palette lookup, grey levels scaled to 0..255, high byte of 16-bit samples; a tRNS
    // colour key adds an alpha channel (0 where the pixel equals the key)
    const bool keyed = color_type != 3 && !trns.empty() && (color_type == 0 || color_type == 2) && trns.size() >= (size_t)src_ch * 2;
    const int out_ch = color_type == 3 ? (trns.empty() ? 3 : 4) : src_ch + (keyed ? 1 : 0);
    static const int grey_scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    out.resize((size_t)w * h * out_ch);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint16_t* sp = &samples[i * src_ch];
        uint8_t* o = &out[i * out_ch];
        if (color_type == 3) {
            const size_t idx = sp[0];
            for (int k = 0; k < 3; k++) o[k] = idx * 3 + k < palette.size() ? palette[idx * 3 + k] : 0;
            if (out_ch == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
            continue;
        }
        bool is_key = keyed;
        for (int k = 0; k < src_ch; k++) {
            if (keyed && sp[k] != (uint16_t)((trns[2 * k] << 8) | trns[2 * k + 1])) is_key = false;
            o[k] = bit_depth == 16 ? (uint8_t)(sp[k] >> 8) : bit_depth == 8 ? (uint8_t)sp[k] : (uint8_t)(sp[k] * grey_scale[bit_depth]);
        }
        if (keyed) o[src_ch] = is_key ? 0 : 255;
    }
    ch = out_ch;
    return true;
}

// ------------------------------------------------------------------ PNM decode ----------
bool decode_pnm(const std::vector<uint8_t>& file, int& w, int& h, int& ch, std::vector<uint8_t>& out, std::string& err) {
    if (file.size() < 2 || file[0] != 'P' || (file[1] != '5' && file[1] != '6')) { err = "not a binary PGM/PPM file"; return false; }
    ch = file[1] == '5' ? 1 : 3;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < file.size()) {
        while (pos < file.size() && std::isspace(file[pos])) pos++;
        if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; continue; }
        int v = 0; bool any = false;
        while (pos < file.size() && std::isdigit(file[pos])) {
            if (v > kMaxImageSide) { err = "PNM header value out of range"; return false; }
            v = v * 10 + (file[pos] - '0'); pos++; any = true;
        }
        if (!any) break;
        vals[got++] = v;
    }
    if (got < 3 || vals[2] != 255) { err = "unsupported PNM header"; return false; }
    pos++;   // single whitespace after maxval
    if (!plausible_size(vals[0], vals[1])) { err = "PNM dimensions out of range"; return false; }
    w = vals[0]; h = vals[1];
    size_t n = (size_t)w * h * ch;
    if (pos + n > file.size()) { err = "truncated PNM"; return false; }
    out.assign(file.begin() + pos, file.begin() + pos + n);
    return true;
}

// ------------------------------------------------------------------ Radiance HDR decode -
// Decodes RGBE to float, then converts to 8 bit the way stb_image does for an HDR file that
// is requested as LDR: pow(v * (1/scale), 1/gamma) * 255 + 0.5, clamped to [0,255].
bool decode_hdr(const std::vector<uint8_t>& file, float gamma, float scale, int& w, int& h, int& ch,
                std::vector<uint8_t>& out, std::string& err) {
    size_t pos = 0;
    auto getline = [&](std::string& line) {
        line.clear();
        while (pos < file.size() && file[pos] != '\n') line += (char)file[pos++];
        if (pos < file.size()) pos++;
        return true;
    };
    std::string line;
    getline(line);
    if (line != "#?RADIANCE" && line != "#?RGBE") { err = "not a Radiance HDR file"; return false; }
    bool format_ok = false;
    while (pos < file.size()) {
        getline(line);
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) { err = "unsupported HDR format"; return false; }
    getline(line);
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2) { err = "unsupported HDR orientation"; return false; }
    if (!plausible_size(w, h)) { err = "HDR dimensions out of range"; return false; }
    std::vector<float> img((size_t)w * h * 3);
    auto convert = [&](size_t pix, const uint8_t* rgbe) {
        if (rgbe[3] != 0) {
            float f1 = std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
            img[pix * 3 + 0] = rgbe[0] * f1; img[pix * 3 + 1] = rgbe[1] * f1; img[pix * 3 + 2] = rgbe[2] * f1;
        } else {
            img[pix * 3 + 0] = img[pix * 3 + 1] = img[pix * 3 + 2] = 0.0f;
        }
    };
    if (w < 8 || w >= 32768) {
        for (size_t i = 0; i < (size_t)w * h; i++) {
            if (pos + 4 > file.size()) { err = "truncated HDR"; return false; }
            convert(i, &file[pos]);
            pos += 4;
        }
    } else {
        std::vector<uint8_t> scan((size_t)w * 4);
        for (int y = 0; y < h; y++) {
            if (pos + 4 > file.size()) { err = "truncated HDR"; return false; }
            if (file[pos] != 2 || file[pos + 1] != 2 || (file[pos + 2] & 0x80)) {
                // flat (non-RLE) data for the rest of the image
                for (size_t i = (size_t)y * w; i < (size_t)w * h; i++) {
                    if (pos + 4 > file.size()) { err = "truncated HDR"; return false; }
                    convert(i, &file[pos]);
                    pos += 4;
                }
                break;
            }
            int len = (file[pos + 2] << 8) | file[pos + 3];
            pos += 4;
            if (len != w) { err = "corrupt HDR scanline"; return false; }
            for (int k = 0; k < 4; k++) {
                int i = 0;
                while (i < w) {
                    if (pos >= file.size()) { err = "truncated HDR"; return false; }
                    uint8_t count = file[pos++];
                    if (count > 128) {
                        count -= 128;
                        if (pos >= file.size() || i + count > w) { err = "corrupt HDR run"; return false; }
                        uint8_t value = file[pos++];
                        for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = value;
                    } else {
                        if (pos + count > file.size() || i + count > w) { err = "corrupt HDR run"; return false; }
                        for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = file[pos++];
                    }
                }
            }
            for (int x = 0; x < w; x++) convert((size_t)y * w + x, &scan[(size_t)x * 4]);
        }
    }
    float inv_gamma = 1.0f / gamma, inv_scale = 1.0f / scale;
    out.resize((size_t)w * h * 3);
    for (size_t i = 0; i < out.size(); i++) {
        float z = (float)std::pow(img[i] * inv_scale, inv_gamma) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        out[i] = (uint8_t)(int)z;
    }
    ch = 3;
    return true;
}

// Convert between channel counts like stb's stbi__convert_format (grey = (77r+150g+29b)>>8).
void convert_channels(std::vector<uint8_t>& px, int w, int h, int from, int to) {
    if (from == to) return;
    std::vector<uint8_t> out((size_t)w * h * to);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* s = &px[i * from];
        uint8_t r, g, b, a = 255;
        if (from <= 2) { r = g = b = s[0]; if (from == 2) a = s[1]; }
        else { r = s[0]; g = s[1]; b = s[2]; if (from == 4) a = s[3]; }
        uint8_t y = (uint8_t)(((int)r * 77 + (int)g * 150 + (int)b * 29) >> 8);
        uint8_t* d = &out[i * to];
        switch (to) {
        case 1: d[0] = from <= 2 ? r : y; break;
        case 2: d[0] = from <= 2 ? r : y; d[1] = a; break;
        case 3: d[0] = r; d[1] = g; d[2] = b; break;
        case 4: d[0] = r; d[1] = g; d[2] = b; d[3] = a; break;
        }
    }
    px.swap(out);
}

// ------------------------------------------------------------------ writers -------------
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void png_chunk(std::vector<uint8_t>& file, const char* type, const std::vector<uint8_t>& body) {
    put_be32(file, (uint32_t)body.size());
    size_t start = file.size();
    file.insert(file.end(), type, type + 4);
    file.insert(file.end(), body.begin(), body.end());
    uint32_t crc = crc32(0L, &file[start], (uInt)(file.size() - start));
    put_be32(file, crc);
}
bool write_file(const char* filename, const std::vector<uint8_t>& data) {
    FILE* f = std::fopen(filename, "wb");
    if (!f) return false;
    size_t n = std::fwrite(data.data(), 1, data.size(), f);
    return std::fclose(f) == 0 && n == data.size();
}
bool write_png(const char* filename, int w, int h, const uint8_t* rgb) {
    std::vector<uint8_t> raw((size_t)(w * 3 + 1) * h);
    for (int y = 0; y < h; y++) {
        raw[(size_t)(w * 3 + 1) * y] = 0;
        std::memcpy(&raw[(size_t)(w * 3 + 1) * y + 1], rgb + (size_t)w * 3 * y, (size_t)w * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    z.resize(zlen);
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(file, "IHDR", ihdr);
    png_chunk(file, "IDAT", z);
    png_chunk(file, "IEND", {});
    return write_file(filename, file);
}
bool write_bmp(const char* filename, int w, int h, const uint8_t* rgb) {
    int pad = (-w * 3) & 3;
    uint32_t data_size = (uint32_t)(w * 3 + pad) * h;
    std::vector<uint8_t> f;
    auto le16 = [&](uint16_t x) { f.push_back(x & 255); f.push_back(x >> 8); };
    auto le32 = [&](uint32_t x) { for (int i = 0; i < 4; i++) f.push_back((x >> (8 * i)) & 255); };
    f.push_back('B'); f.push_back('M');
    le32(14 + 40 + data_size); le16(0); le16(0); le32(14 + 40);
    le32(40); le32((uint32_t)w); le32((uint32_t)h); le16(1); le16(24); le32(0); le32(0); le32(0); le32(0); le32(0); le32(0);
    for (int y = h - 1; y >= 0; y--) {
        for (int x = 0; x < w; x++) {
            const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
            f.push_back(p[2]); f.push_back(p[1]); f.push_back(p[0]);
        }
        for (int k = 0; k < pad; k++) f.push_back(0);
    }
    return write_file(filename, f);
}
bool write_tga(const char* filename, int w, int h, const uint8_t* rgb) {
    std::vector<uint8_t> f = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (uint8_t)(w & 255), (uint8_t)(w >> 8), (uint8_t)(h & 255), (uint8_t)(h >> 8), 24, 0};
    for (int y = h - 1; y >= 0; y--)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
            f.push_back(p[2]); f.push_back(p[1]); f.push_back(p[0]);
        }
    return write_file(filename, f);
}
bool write_ppm(const char* filename, int w, int h, const uint8_t* rgb) {
    std::string head = "P6\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    std::vector<uint8_t> f(head.begin(), head.end());
    f.insert(f.end(), rgb, rgb + (size_t)w * h * 3);
    return write_file(filename, f);
}
// Radiance .hdr with flat RGBE pixels of data/255 (image.cuh:109-114)
bool write_hdr(const char* filename, int w, int h, const uint8_t* rgb) {
    std::string head = "#?RADIANCE\n# Written by crt\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(h) + " +X " + std::to_string(w) + "\n";
    std::vector<uint8_t> f(head.begin(), head.end());
    for (size_t i = 0; i < (size_t)w * h; i++) {
        float c[3] = {rgb[3 * i] / 255.0f, rgb[3 * i + 1] / 255.0f, rgb[3 * i + 2] / 255.0f};
        float m = std::fmax(c[0], std::fmax(c[1], c[2]));
        if (m < 1e-32f) { f.push_back(0); f.push_back(0); f.push_back(0); f.push_back(0); continue; }
        int e;
        float norm = (float)std::frexp(m, &e) * 256.0f / m;
        f.push_back((uint8_t)(c[0] * norm)); f.push_back((uint8_t)(c[1] * norm)); f.push_back((uint8_t)(c[2] * norm));
        f.push_back((uint8_t)(e + 128));
    }
    return write_file(filename, f);
}

// color_value (vec3.cuh:491-493); a NaN converts like the x86 cvttss2si the host build uses.
inline uint8_t color_value(float value) {
    if (std::isnan(value)) return 0;
    float c = value < 0.0f ? 0.0f : (value > 0.9999999f ? 0.9999999f : value);
    return (uint8_t)static_cast<int>(256 * c);
}

}  // namespace

namespace crt {

bool decode_jpeg(const std::vector<uint8_t>& file, int req_comp, int& w, int& h, int& native, int& out_comp,
                 std::vector<uint8_t>& out, std::string& err);   // jpeg_decode.cpp
bool write_jpeg(const char* filename, int w, int h, const uint8_t* rgb, int quality);   // jpeg_encode.cpp

int set_error(int code, const std::string& message) {
    g_last_error = message;
    return code;
}
void clear_error() { g_last_error.clear(); }

bool load_image_file(const std::string& path, float hdr_gamma, float hdr_scale, int desired_channels, bool flip_y,
                     int& width, int& height, int& channels, std::vector<uint8_t>& pixels, std::string& error) {
    crt_image_loader_fn fn;
    void* user;
    {
        std::lock_guard<std::mutex> lock(g_loader_mutex);
        fn = g_loader;
        user = g_loader_user;
    }
    if (fn) {
        uint8_t* px = nullptr;
        int rc = fn(path.c_str(), hdr_gamma, hdr_scale, desired_channels, flip_y ? 1 : 0, &width, &height, &channels, &px, user);
        if (rc != 0 || !px || width <= 0 || height <= 0 || channels < 1 || channels > 4) {
            error = "image loader callback failed";
            if (px) std::free(px);
            return false;
        }
        pixels.assign(px, px + (size_t)width * height * channels);
        std::free(px);
        return true;
    }
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { error = "cannot read file"; return false; }
    bool ok;
    const bool is_jpeg = file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8;
    if (is_jpeg) {
        // like stb_image, a JPEG is converted to the requested channel count by its own decoder
        // (a grey request on a YCbCr file returns the Y plane, not a luma of the RGB values)
        int native = 0, got = 0;
        ok = decode_jpeg(file, (desired_channels >= 1 && desired_channels <= 4) ? desired_channels : 0, width, height, native, got, pixels, error);
        channels = got;
    }
    // the format is told from the file's first bytes, like stb_image does, not from its name
    else if (file.size() >= 8 && !std::memcmp(file.data(), "\x89PNG\r\n\x1a\n", 8)) ok = decode_png(file, width, height, channels, pixels, error);
    else if (file.size() >= 6 && (!std::memcmp(file.data(), "#?RADIANCE", std::min<size_t>(10, file.size())) || !std::memcmp(file.data(), "#?RGBE", 6)))
        ok = decode_hdr(file, hdr_gamma, hdr_scale, width, height, channels, pixels, error);
    else if (file.size() >= 2 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) ok = decode_pnm(file, width, height, channels, pixels, error);
    else if (looks_like_bmp(file)) ok = decode_bmp(file, width, height, channels, pixels, error);
    // stb_image's GIF, PSD and PIC readers have signatures too: named here so that such a file is not taken for a TGA
    else if (file.size() >= 4 && (!std::memcmp(file.data(), "GIF8", 4) || !std::memcmp(file.data(), "8BPS", 4) || !std::memcmp(file.data(), "\x53\x80\xF6\x34", 4))) {
        error = "no built-in decoder for GIF, PSD or PIC files (PNG, JPEG, BMP, TGA, Radiance HDR and binary PGM/PPM are built in): install a decoder with crt_set_image_loader";
        return false;
    }
    else if (looks_like_tga(file)) ok = decode_tga(file, width, height, channels, pixels, error);       // no signature: tested last, like stb_image does
    else { error = "no built-in decoder for this file type (PNG, JPEG, BMP, TGA, Radiance HDR and binary PGM/PPM are built in; stb_image's GIF, PSD and PIC are not: install a decoder with crt_set_image_loader)"; return false; }
    if (!ok) return false;
    if (desired_channels >= 1 && desired_channels <= 4 && desired_channels != channels) {
        convert_channels(pixels, width, height, channels, desired_channels);
        channels = desired_channels;
    }
    if (flip_y) {
        size_t row = (size_t)width * channels;
        std::vector<uint8_t> tmp(row);
        for (int y = 0; y < height / 2; y++) {
            uint8_t* a = &pixels[row * y];
            uint8_t* b = &pixels[row * (height - 1 - y)];
            std::memcpy(tmp.data(), a, row); std::memcpy(a, b, row); std::memcpy(b, tmp.data(), row);
        }
    }
    return true;
}

}  // namespace crt

extern "C" {

const char* crt_last_error(void) { return g_last_error.c_str(); }
const char* crt_version(void) { return "cudaraytracer_amd 0.1 (gfx950)"; }

int crt_set_image_loader(crt_image_loader_fn fn, void* user) {
    std::lock_guard<std::mutex> lock(g_loader_mutex);
    g_loader = fn;
    g_loader_user = user;
    return CRT_OK;
}

int crt_load_image(const char* filename, float hdr_gamma, float hdr_scale, int desired_channels, int flip_y,
                   int* width, int* height, int* channels, uint8_t** pixels) {
    crt::clear_error();
    if (!filename || !width || !height || !channels || !pixels) return crt::set_error(CRT_ERR_INVALID, "crt_load_image: null argument");
    std::vector<uint8_t> px;
    std::string err;
    try {                                                    // no exception may cross the C boundary
        if (!crt::load_image_file(filename, hdr_gamma, hdr_scale, desired_channels, flip_y != 0, *width, *height, *channels, px, err))
            return crt::set_error(CRT_ERR_IO, std::string("crt_load_image: ") + filename + ": " + err);
    } catch (const std::exception& e) {
        return crt::set_error(CRT_ERR_IO, std::string("crt_load_image: ") + filename + ": " + e.what());
    }
    *pixels = static_cast<uint8_t*>(std::malloc(px.size() ? px.size() : 1));
    if (!*pixels) return crt::set_error(CRT_ERR_IO, "crt_load_image: out of memory");
    std::memcpy(*pixels, px.data(), px.size());
    return CRT_OK;
}

void crt_free(void* p) { std::free(p); }

int crt_normalize_pass(int pass, int width, int height, float gamma, const float* data, uint8_t* out) {
    crt::clear_error();
    if (!data || !out || width <= 0 || height <= 0 || pass < 0 || pass > 5) return crt::set_error(CRT_ERR_INVALID, "crt_normalize_pass: bad arguments");
    const size_t n = (size_t)width * height;
    if (pass <= 2) {            // normalize_color / albedo / emission (buffer.cuh:202-240)
        const float inv_gamma = 1.0f / gamma;
        for (size_t i = 0; i < n * 3; i++) out[i] = color_value(std::pow(data[i], inv_gamma));
    } else if (pass == 3) {     // normalize_normal (buffer.cuh:247-255)
        for (size_t i = 0; i < n * 3; i++) out[i] = color_value((data[i] + 1.0f) / 2.0f);
    } else if (pass == 4) {     // normalize_depth (buffer.cuh:262-278)
        float max_depth = 0.0f, min_depth = std::numeric_limits<float>::max();
        for (size_t i = 0; i < n; i++) {
            if (!std::isnan(data[i])) {
                max_depth = max_depth > data[i] ? max_depth : data[i];
                min_depth = min_depth < data[i] ? min_depth : data[i];
            }
        }
        for (size_t i = 0; i < n; i++) {
            float d = data[i] < max_depth ? data[i] : max_depth;     // NaN -> max_depth
            float m = 0.0f + (d - min_depth) * (1.0f - 0.0f) / (max_depth - min_depth);
            out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = color_value(m);
        }
    } else {                    // normalize_opacity (buffer.cuh:285-293)
        for (size_t i = 0; i < n; i++) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = color_value(data[i]);
    }
    return CRT_OK;
}

int crt_write_image(const char* filename, const char* fmt, int width, int height, const uint8_t* rgb) {
    crt::clear_error();
    if (!filename || !fmt || !rgb || width <= 0 || height <= 0) return crt::set_error(CRT_ERR_INVALID, "crt_write_image: bad arguments");
    std::string f = fmt;
    bool ok;
    if (f == "png") ok = write_png(filename, width, height, rgb);
    else if (f == "jpg") ok = crt::write_jpeg(filename, width, height, rgb, 100);    // write_image's default quality (image.cuh:100)
    else if (f == "bmp") ok = write_bmp(filename, width, height, rgb);
    else if (f == "tga") ok = write_tga(filename, width, height, rgb);
    else if (f == "hdr") ok = write_hdr(filename, width, height, rgb);
    else if (f == "ppm") ok = write_ppm(filename, width, height, rgb);
    else return crt::set_error(CRT_ERR_UNSUPPORTED, "Unsupported image format: " + f);
    if (!ok) return crt::set_error(CRT_ERR_IO, std::string("could not write '") + filename + "'");
    return CRT_OK;
}

}  // extern "C"
