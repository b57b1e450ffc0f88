// BMP and TGA decoders for image textures.
//
// The reference loads every image texture through stbi_load (src/texture.cuh:143-147), which tells
// the format from the file's bytes.  These two decoders give what stb_image v2.30 gives for the
// same file (external/stb/stb_image.h:5346-5732 BMP, :5739-6074 TGA): channel count, pixel values
// and the quirks a scene's textures can depend on:
//   BMP  headers of 12 / 40 / 56 / 108 / 124 bytes; 1, 4, 8 bits with a palette; 16 and 32 bits
//        through channel masks (5-5-5 and 8-8-8-8 by default), 24 bits; no RLE; rows bottom-up
//        unless the height is negative; a 32-bit file whose alpha bytes are all zero is opaque;
//        a channel of n < 8 bits is widened by repeating its bits.
//   TGA  types 1, 2, 3 and their run-length forms 9, 10, 11; 8-bit grey -> 1 channel, 16-bit grey
//        -> 2 channels (grey, alpha), 15/16-bit colour -> RGB with c * 255 / 31, 24 / 32 bits ->
//        RGB / RGBA; colour-mapped images with 8- or 16-bit indices (an index beyond the map reads
//        entry 0); rows bottom-up unless bit 5 of the descriptor is set; bytes missing at the end
//        of the file read as zero.
// Different by design: every read is bounds-checked and sizes are validated before anything is
// allocated; where stb_image would read palette entries it never initialised (an index beyond a
// BMP's palette) the colour is black.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "crt_internal.h"

namespace {

// Little-endian reader over the file; beyond the end every byte reads as zero (stb_image's get8).
class Bytes {
public:
    explicit Bytes(const std::vector<uint8_t>& file) : data_(file.data()), size_(file.size()) {}
    uint8_t u8() { return pos_ < size_ ? data_[pos_++] : (pos_++, uint8_t(0)); }
    uint32_t u16() { const uint32_t lo = u8(); return lo | (uint32_t(u8()) << 8); }
    uint32_t u32() { const uint32_t lo = u16(); return lo | (u16() << 16); }
    int32_t i32() { return (int32_t)u32(); }
    void skip(long long n) { if (n > 0) pos_ += (size_t)n; }           // (a negative skip does nothing, like stbi__skip)
    size_t position() const { return pos_; }
    bool exhausted() const { return pos_ >= size_; }
private:
    const uint8_t* data_;
    size_t size_, pos_ = 0;
};

bool sane_size(long long w, long long h) { return w > 0 && h > 0 && w <= (1 << 24) && h <= (1 << 24) && w * h <= (1ll << 30); }

int top_bit(uint32_t v) {
    int n = -1;
    while (v) { n++; v >>= 1; }
    return n;
}
int bit_count(uint32_t v) {
    int n = 0;
    for (; v; v &= v - 1) n++;
    return n;
}
// The `bits` highest of the 8 bits that end at bit `shift` + 7 of v, widened to 8 bits by repetition.
int widen(uint32_t v, int shift, int bits) {
    static const uint32_t repeat[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
    static const int down[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
    v = shift < 0 ? v << -shift : v >> shift;
    v = (v & 0xff) >> (8 - bits);
    return (int)((v * repeat[bits]) >> down[bits]);
}

void flip_rows(std::vector<uint8_t>& px, int w, int h, int ch) {
    const size_t row = (size_t)w * ch;
    std::vector<uint8_t> tmp(row);
    for (int y = 0; y < h / 2; y++) {
        uint8_t* a = &px[row * y];
        uint8_t* b = &px[row * (h - 1 - y)];
        std::memcpy(tmp.data(), a, row); std::memcpy(a, b, row); std::memcpy(b, tmp.data(), row);
    }
}

}  // namespace

namespace crt {

bool looks_like_bmp(const std::vector<uint8_t>& file) {
    if (file.size() < 18 || file[0] != 'B' || file[1] != 'M') return false;
    Bytes in(file);
    in.skip(14);
    const uint32_t header = in.u32();
    return header == 12 || header == 40 || header == 56 || header == 108 || header == 124;
}

bool decode_bmp(const std::vector<uint8_t>& file, int& w, int& h, int& ch, std::vector<uint8_t>& out, std::string& err) {
    Bytes in(file);
    if (in.u8() != 'B' || in.u8() != 'M') { err = "not a BMP file"; return false; }
    in.skip(8);                                                // file size, reserved
    const long long offset = in.i32();
    const int header = in.i32();
    if (offset < 0) { err = "bad BMP: negative data offset"; return false; }
    if (header != 12 && header != 40 && header != 56 && header != 108 && header != 124) { err = "BMP header type not supported"; return false; }
    long long width, height;
    if (header == 12) { width = in.u16(); height = in.u16(); }
    else { width = in.i32(); height = in.i32(); }
    if (in.u16() != 1) { err = "bad BMP: colour planes"; return false; }
    const int bpp = (int)in.u16();
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0, any_alpha = 255;
    long long header_bytes = 14;                               // bytes in front of the info header, plus masks read after it
    auto default_masks = [&](int compress) {
        if (compress != 0) return;
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; any_alpha = 0; }   // alpha counts only if some pixel has it
        else mr = mg = mb = ma = 0;
    };
    if (header != 12) {
        const int compress = in.i32();
        if (compress == 1 || compress == 2) { err = "BMP type not supported: RLE"; return false; }
        if (compress >= 4 || compress < 0) { err = "BMP type not supported: compression"; return false; }
        if (compress == 3 && bpp != 16 && bpp != 32) { err = "bad BMP: channel masks need 16 or 32 bits per pixel"; return false; }
        in.skip(20);                                           // image size, resolution, colours used / important
        if (header == 40 || header == 56) {
            if (header == 56) in.skip(16);
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) default_masks(0);
                else {                                         // (compress == 3)
                    mr = in.u32(); mg = in.u32(); mb = in.u32();
                    header_bytes += 12;
                    if (mr == mg && mg == mb) { err = "bad BMP: channel masks"; return false; }
                }
            }
        } else {
            mr = in.u32(); mg = in.u32(); mb = in.u32(); ma = in.u32();
            if (compress != 3) default_masks(compress);
            in.skip(header == 124 ? 68 : 52);                  // colour space (and rendering intent, profile)
        }
    }
    const bool bottom_up = height > 0;
    if (height < 0) height = -height;
    if (!sane_size(width, height)) { err = "BMP dimensions out of range"; return false; }
    long long palette = 0;
    if (header == 12) { if (bpp < 24) palette = (offset - header_bytes - 24) / 3; }
    else if (bpp < 16) palette = (offset - header_bytes - header) >> 2;
    if (palette == 0) {
        // some bytes may follow the header, but an offset before its end or far beyond it is a broken file
        const long long read = (long long)in.position();
        if (read > 1024 || offset < read || offset - read > 1024) { err = "corrupt BMP: data offset"; return false; }
        in.skip(offset - read);
    }
    ch = (bpp == 24 && ma == 0xff000000u) ? 3 : (ma ? 4 : 3);
    w = (int)width; h = (int)height;
    out.assign((size_t)w * h * ch, 0);
    size_t z = 0;
    if (bpp < 16) {
        if (palette <= 0 || palette > 256) { err = "corrupt BMP: palette"; return false; }
        uint8_t pal[256][3] = {};
        for (long long i = 0; i < palette; i++) {
            pal[i][2] = in.u8(); pal[i][1] = in.u8(); pal[i][0] = in.u8();
            if (header != 12) in.u8();
        }
        in.skip(offset - header_bytes - header - palette * (header == 12 ? 3 : 4));
        if (bpp != 1 && bpp != 4 && bpp != 8) { err = "corrupt BMP: bits per pixel"; return false; }
        const int row_bytes = bpp == 1 ? (w + 7) >> 3 : (bpp == 4 ? (w + 1) >> 1 : w);
        const int pad = (-row_bytes) & 3;
        const int per_byte = 8 / bpp, mask = (1 << bpp) - 1;
        for (int y = 0; y < h; y++) {
            int byte = 0;
            for (int x = 0; x < w; x++) {
                if (x % per_byte == 0) byte = in.u8();
                const int index = (byte >> (bpp * (per_byte - 1 - x % per_byte))) & mask;
                out[z++] = pal[index][0]; out[z++] = pal[index][1]; out[z++] = pal[index][2];
                if (ch == 4) out[z++] = 255;
            }
            in.skip(pad);
        }
    } else {
        in.skip(offset - header_bytes - header);
        if (bpp != 16 && bpp != 24 && bpp != 32) { err = "corrupt BMP: bits per pixel"; return false; }
        const int pad = bpp == 24 ? (-3 * w) & 3 : (bpp == 16 ? (-2 * w) & 3 : 0);
        const int plain = bpp == 24 ? 1 : ((bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0xff0000u && ma == 0xff000000u) ? 2 : 0);
        int rs = 0, gs = 0, bs = 0, as = 0, rc = 0, gc = 0, bc = 0, ac = 0;
        if (!plain) {
            if (!mr || !mg || !mb) { err = "corrupt BMP: channel masks"; return false; }
            rs = top_bit(mr) - 7; rc = bit_count(mr);
            gs = top_bit(mg) - 7; gc = bit_count(mg);
            bs = top_bit(mb) - 7; bc = bit_count(mb);
            as = top_bit(ma) - 7; ac = bit_count(ma);
            if (rc > 8 || gc > 8 || bc > 8 || ac > 8) { err = "corrupt BMP: channel masks"; return false; }
        }
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                uint32_t r, g, b, a;
                if (plain) {
                    b = in.u8(); g = in.u8(); r = in.u8();
                    a = plain == 2 ? in.u8() : 255;
                } else {
                    const uint32_t v = bpp == 16 ? in.u16() : in.u32();
                    r = (uint32_t)widen(v & mr, rs, rc); g = (uint32_t)widen(v & mg, gs, gc); b = (uint32_t)widen(v & mb, bs, bc);
                    a = ma ? (uint32_t)widen(v & ma, as, ac) : 255;
                }
                any_alpha |= a;
                out[z++] = (uint8_t)r; out[z++] = (uint8_t)g; out[z++] = (uint8_t)b;
                if (ch == 4) out[z++] = (uint8_t)a;
            }
            in.skip(pad);
        }
    }
    if (ch == 4 && any_alpha == 0)
        for (size_t i = 3; i < out.size(); i += 4) out[i] = 255;
    if (bottom_up) flip_rows(out, w, h, ch);
    return true;
}

namespace {
// Channels of a TGA pixel (or colour-map entry) of `bits` bits; 0 = not a format this decoder reads.
int tga_channels(int bits, bool grey, bool* rgb16) {
    if (rgb16) *rgb16 = false;
    switch (bits) {
    case 8: return 1;
    case 16: if (grey) return 2;
             /* fallthrough */
    case 15: if (rgb16) *rgb16 = true;
             return 3;
    case 24: return 3;
    case 32: return 4;
    default: return 0;
    }
}
}  // namespace

// TGA files carry no signature: the header has to be consistent (stb_image tests for TGA last).
bool looks_like_tga(const std::vector<uint8_t>& file) {
    if (file.size() < 18) return false;
    Bytes in(file);
    in.u8();
    const int mapped = in.u8();
    if (mapped > 1) return false;
    const int type = in.u8();
    if (mapped == 1) {
        if (type != 1 && type != 9) return false;
        in.skip(4);
        const int entry_bits = in.u8();
        if (entry_bits != 8 && entry_bits != 15 && entry_bits != 16 && entry_bits != 24 && entry_bits != 32) return false;
        in.skip(4);
    } else {
        if (type != 2 && type != 3 && type != 10 && type != 11) return false;
        in.skip(9);
    }
    if (in.u16() < 1 || in.u16() < 1) return false;
    const int bits = in.u8();
    if (mapped == 1 && bits != 8 && bits != 16) return false;
    return bits == 8 || bits == 15 || bits == 16 || bits == 24 || bits == 32;
}

bool decode_tga(const std::vector<uint8_t>& file, int& w, int& h, int& ch, std::vector<uint8_t>& out, std::string& err) {
    Bytes in(file);
    const int id_length = in.u8();
    const bool mapped = in.u8() != 0;
    int type = in.u8();
    const int map_first = (int)in.u16(), map_length = (int)in.u16(), map_bits = in.u8();
    in.skip(4);                                                // x and y origin
    w = (int)in.u16(); h = (int)in.u16();
    const int bits = in.u8();
    const bool top_down = (in.u8() >> 5) & 1;
    const bool rle = type >= 8;
    if (rle) type -= 8;
    bool rgb16 = false;
    ch = mapped ? tga_channels(map_bits, false, &rgb16) : tga_channels(bits, type == 3, &rgb16);
    if (!ch) { err = "TGA pixel format not supported"; return false; }
    if (!sane_size(w, h)) { err = "TGA dimensions out of range"; return false; }
    out.assign((size_t)w * h * ch, 0);
    in.skip(id_length);
    auto colour16 = [&](uint8_t* px) {                        // 5-5-5, stored red first
        const uint32_t v = in.u16();
        px[0] = (uint8_t)((((v >> 10) & 31u) * 255u) / 31u);
        px[1] = (uint8_t)((((v >> 5) & 31u) * 255u) / 31u);
        px[2] = (uint8_t)(((v & 31u) * 255u) / 31u);
    };
    std::vector<uint8_t> map;
    if (mapped) {
        if (map_length == 0) { err = "corrupt TGA: empty colour map"; return false; }
        in.skip(map_first);
        map.assign((size_t)map_length * ch, 0);
        for (int i = 0; i < map_length; i++) {
            if (rgb16) colour16(&map[(size_t)i * ch]);
            else for (int k = 0; k < ch; k++) map[(size_t)i * ch + k] = in.u8();
        }
    }
    uint8_t px[4] = {0, 0, 0, 0};
    int run = 0;
    bool repeat = false;
    for (size_t i = 0, n = (size_t)w * h; i < n; i++) {
        bool read = true;
        if (rle) {
            if (run == 0) {
                const int packet = in.u8();
                run = 1 + (packet & 127);
                repeat = (packet >> 7) != 0;
            } else if (repeat) {
                read = false;
            }
            run--;
        }
        if (read) {
            if (mapped) {
                int index = bits == 8 ? in.u8() : (int)in.u16();
                if (index >= map_length) index = 0;
                std::memcpy(px, &map[(size_t)index * ch], (size_t)ch);
            } else if (rgb16) {
                colour16(px);
            } else {
                for (int k = 0; k < ch; k++) px[k] = in.u8();
            }
        }
        std::memcpy(&out[i * ch], px, (size_t)ch);
    }
    if (!top_down) flip_rows(out, w, h, ch);
    if (ch >= 3 && !rgb16)                                     // stored blue first
        for (size_t i = 0; i < out.size(); i += (size_t)ch) std::swap(out[i], out[i + 2]);
    return true;
}

}  // namespace crt
