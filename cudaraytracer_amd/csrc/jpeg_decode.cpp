// JPEG decoder for image textures (baseline, extended-sequential and progressive Huffman,
// 8-bit samples, 1/3/4 components, any 1..4 sampling factors, restart intervals).
//
// The reference loads textures with stb_image (src/texture.cuh:143-158 -> stbi_load,
// external/stb/stb_image.h), so texel values have to come out byte-identical to that decoder.
// Entropy decoding is fixed by the JPEG standard (ITU-T T.81 Annex F/G); the three places where
// decoders legitimately differ are restated with stb_image's arithmetic:
//   * inverse DCT: the 12-bit fixed-point "islow" butterfly, column pass kept at 2 extra bits
//     (+512 >> 10), row pass rounded and level-shifted in one step (+65536 + (128 << 17) >> 17)
//     (stb_image.h:2426-2524),
//   * chroma upsampling: centred triangle filter, (3a + b + 2) >> 2 along one axis and
//     (3(3a + b) + (3a' + b') + 8) >> 4 for 2x2, nearest-neighbour for other ratios
//     (stb_image.h:3465-3527, 3646-3655), rows scheduled as in stb_image.h:3907-3940,
//   * YCbCr -> RGB in 20-bit fixed point with the chroma products truncated to 16 fractional
//     bits the way stb_image.h:3659-3683 does.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace crt {

namespace {

const uint8_t kZigzag[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
    13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45,
    38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // a corrupt run may step past 63; park those writes on the last coefficient
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffmanTable {
    bool present = false;
    uint8_t values[256];
    int min_code[17], max_code[18], first_value[17];     // per code length 1..16 (T.81 F.2.2.3)
    void build(const uint8_t counts[16], const uint8_t* symbols, int total) {
        std::memcpy(values, symbols, (size_t)total);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            first_value[len] = k;
            min_code[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            max_code[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        max_code[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, dc_table = 0, ac_table = 0;
    int width = 0, height = 0;          // samples that carry image content
    int blocks_w = 0, blocks_h = 0;     // allocated blocks (whole MCUs)
    int stride = 0;                     // blocks_w * 8
    int dc_pred = 0;
    std::vector<uint8_t> plane;         // decoded samples
    std::vector<int16_t> coef;          // progressive: 64 per block, natural order
};

class Decoder {
public:
    Decoder(const uint8_t* data, size_t size) : p_(data), n_(size) {}
    bool run(int req_comp, int& w, int& h, int& native, int& out_comp, std::vector<uint8_t>& out, std::string& err);

private:
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    std::string err_;
    uint16_t quant_[4][64] = {};        // natural order
    HuffmanTable dc_[4], ac_[4];
    Component comp_[4];
    int ncomp_ = 0, width_ = 0, height_ = 0, hmax_ = 1, vmax_ = 1, mcus_x_ = 0, mcus_y_ = 0;
    bool progressive_ = false, jfif_ = false;
    int adobe_transform_ = -1, rgb_ids_ = 0, restart_interval_ = 0;
    // scan state
    int scan_n_ = 0, order_[4] = {0, 0, 0, 0}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0, eob_run_ = 0, todo_ = 0;
    // entropy-coded segment reader
    uint32_t bit_buf_ = 0;
    int bit_count_ = 0;
    int pending_marker_ = -1;           // marker met inside the entropy-coded data (0xFF stripped)
    bool exhausted_ = false;

    bool fail(const char* msg) { err_ = msg; return false; }
    int get8() { return pos_ < n_ ? p_[pos_++] : 0; }
    int get16() { int a = get8(); return (a << 8) | get8(); }
    bool eof() const { return pos_ >= n_; }

    void reset_bits() { bit_buf_ = 0; bit_count_ = 0; exhausted_ = false; pending_marker_ = -1; }
    void refill() {
        while (bit_count_ <= 24) {
            int b = 0;
            if (!exhausted_) {
                b = get8();
                if (b == 0xFF) {
                    int c = get8();
                    while (c == 0xFF) c = get8();          // fill bytes
                    if (c != 0) { pending_marker_ = c; exhausted_ = true; b = 0; }
                }
            }
            bit_buf_ |= (uint32_t)b << (24 - bit_count_);
            bit_count_ += 8;
        }
    }
    int get_bits(int count) {
        if (count == 0) return 0;
        if (bit_count_ < count) refill();
        int v = (int)(bit_buf_ >> (32 - count));
        bit_buf_ <<= count;
        bit_count_ -= count;
        return v;
    }
    int get_bit() { return get_bits(1); }
    int decode_symbol(const HuffmanTable& t) {
        if (bit_count_ < 16) refill();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | (int)(bit_buf_ >> 31);
            bit_buf_ <<= 1;
            bit_count_--;
            if (code <= t.max_code[len]) return t.values[t.first_value[len] + code - t.min_code[len]];
        }
        return -1;
    }
    // T.81 F.2.2.1 EXTEND
    int receive_extend(int bits) {
        if (bits == 0) return 0;
        int v = get_bits(bits);
        return v < (1 << (bits - 1)) ? v - (1 << bits) + 1 : v;
    }

    bool read_segment(int marker);
    bool read_frame(int marker);
    bool read_scan_header();
    bool decode_scan();
    bool block_baseline(Component& c, int16_t block[64]);
    bool block_dc_progressive(Component& c, int16_t* block);
    bool block_ac_progressive(Component& c, int16_t* block);
    bool restart_due();
    void finish_progressive();
    int next_marker();
};

// ---------------------------------------------------------------- inverse DCT ----------
inline uint8_t clamp255(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }

inline int fixed12(double x) { return (int)(x * 4096 + 0.5); }

// One 8-point pass: even part into e0..e3, odd part into o0..o3; the caller combines
// e_k +/- o_(3-k) with its own rounding.
struct Pass { int e0, e1, e2, e3, o0, o1, o2, o3; };
inline Pass idct_pass(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
    Pass r;
    int p2 = s2, p3 = s6;
    int p1 = (p2 + p3) * fixed12(0.5411961f);
    int t2 = p1 + p3 * fixed12(-1.847759065f);
    int t3 = p1 + p2 * fixed12(0.765366865f);
    int t0 = (s0 + s4) * 4096, t1 = (s0 - s4) * 4096;
    r.e0 = t0 + t3; r.e3 = t0 - t3; r.e1 = t1 + t2; r.e2 = t1 - t2;
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;
    p3 = t0 + t2;
    int p4 = t1 + t3;
    p1 = t0 + t3;
    p2 = t1 + t2;
    int p5 = (p3 + p4) * fixed12(1.175875602f);
    t0 = t0 * fixed12(0.298631336f);
    t1 = t1 * fixed12(2.053119869f);
    t2 = t2 * fixed12(3.072711026f);
    t3 = t3 * fixed12(1.501321110f);
    p1 = p5 + p1 * fixed12(-0.899976223f);
    p2 = p5 + p2 * fixed12(-2.562915447f);
    p3 = p3 * fixed12(-1.961570560f);
    p4 = p4 * fixed12(-0.390180644f);
    r.o3 = t3 + p1 + p4; r.o2 = t2 + p2 + p3; r.o1 = t1 + p2 + p4; r.o0 = t0 + p1 + p3;
    return r;
}

void idct_block(uint8_t* out, int stride, const int16_t d[64]) {
    int v[64];
    for (int i = 0; i < 8; i++) {
        const int16_t* c = d + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            int dc = c[0] * 4;
            for (int k = 0; k < 8; k++) v[i + 8 * k] = dc;
        } else {
            Pass q = idct_pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
            q.e0 += 512; q.e1 += 512; q.e2 += 512; q.e3 += 512;
            v[i + 0] = (q.e0 + q.o3) >> 10;  v[i + 56] = (q.e0 - q.o3) >> 10;
            v[i + 8] = (q.e1 + q.o2) >> 10;  v[i + 48] = (q.e1 - q.o2) >> 10;
            v[i + 16] = (q.e2 + q.o1) >> 10; v[i + 40] = (q.e2 - q.o1) >> 10;
            v[i + 24] = (q.e3 + q.o0) >> 10; v[i + 32] = (q.e3 - q.o0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {
        const int* r = v + 8 * i;
        uint8_t* o = out + (size_t)stride * i;
        Pass q = idct_pass(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        const int bias = 65536 + (128 << 17);
        q.e0 += bias; q.e1 += bias; q.e2 += bias; q.e3 += bias;
        o[0] = clamp255((q.e0 + q.o3) >> 17); o[7] = clamp255((q.e0 - q.o3) >> 17);
        o[1] = clamp255((q.e1 + q.o2) >> 17); o[6] = clamp255((q.e1 - q.o2) >> 17);
        o[2] = clamp255((q.e2 + q.o1) >> 17); o[5] = clamp255((q.e2 - q.o1) >> 17);
        o[3] = clamp255((q.e3 + q.o0) >> 17); o[4] = clamp255((q.e3 - q.o0) >> 17);
    }
}

// ---------------------------------------------------------------- markers --------------
int Decoder::next_marker() {
    if (pending_marker_ >= 0) { int m = pending_marker_; pending_marker_ = -1; return m; }
    int x = get8();
    if (x != 0xFF) return -1;
    while (x == 0xFF) x = get8();
    return x;
}

bool Decoder::read_segment(int m) {
    switch (m) {
    case -1: return fail("expected a JPEG marker");
    case 0xDD: {                                                   // DRI
        if (get16() != 4) return fail("bad DRI length");
        restart_interval_ = get16();
        return true;
    }
    case 0xDB: {                                                   // DQT
        int len = get16() - 2;
        while (len > 0) {
            int q = get8(), precision16 = q >> 4, t = q & 15;
            if (precision16 > 1) return fail("bad DQT precision");
            if (t > 3) return fail("bad DQT table");
            for (int i = 0; i < 64; i++) quant_[t][kZigzag[i]] = (uint16_t)(precision16 ? get16() : get8());
            len -= precision16 ? 129 : 65;
        }
        return len == 0 ? true : fail("bad DQT length");
    }
    case 0xC4: {                                                   // DHT
        int len = get16() - 2;
        while (len > 0) {
            int q = get8(), cls = q >> 4, t = q & 15;
            if (cls > 1 || t > 3) return fail("bad DHT header");
            uint8_t counts[16], symbols[256];
            int total = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); total += counts[i]; }
            if (total > 256) return fail("bad DHT header");
            len -= 17;
            for (int i = 0; i < total; i++) symbols[i] = (uint8_t)get8();
            (cls == 0 ? dc_[t] : ac_[t]).build(counts, symbols, total);
            len -= total;
        }
        return len == 0 ? true : fail("bad DHT length");
    }
    default:
        break;
    }
    if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {                   // APPn, COM
        int len = get16();
        if (len < 2) return fail(m == 0xFE ? "bad COM length" : "bad APP length");
        len -= 2;
        if (m == 0xE0 && len >= 5) {
            static const char tag[5] = {'J', 'F', 'I', 'F', 0};
            bool ok = true;
            for (int i = 0; i < 5; i++) ok = (get8() == (uint8_t)tag[i]) && ok;
            len -= 5;
            if (ok) jfif_ = true;
        } else if (m == 0xEE && len >= 12) {
            static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
            bool ok = true;
            for (int i = 0; i < 6; i++) ok = (get8() == (uint8_t)tag[i]) && ok;
            len -= 6;
            if (ok) {
                get8(); get16(); get16();                          // version, flags0, flags1
                adobe_transform_ = get8();
                len -= 6;
            }
        }
        pos_ += (size_t)len;
        return true;
    }
    return fail("unknown JPEG marker");
}

bool Decoder::read_frame(int marker) {
    progressive_ = marker == 0xC2;
    int len = get16();
    if (len < 11) return fail("bad SOF length");
    if (get8() != 8) return fail("only 8-bit JPEG is supported");
    height_ = get16();
    width_ = get16();
    if (height_ == 0) return fail("JPEG without a height is not supported");
    if (width_ == 0) return fail("JPEG with zero width");
    ncomp_ = get8();
    if (ncomp_ != 1 && ncomp_ != 3 && ncomp_ != 4) return fail("bad JPEG component count");
    if (len != 8 + 3 * ncomp_) return fail("bad SOF length");
    static const char rgb[3] = {'R', 'G', 'B'};
    rgb_ids_ = 0;
    for (int i = 0; i < ncomp_; i++) {
        Component& c = comp_[i];
        c.id = get8();
        if (ncomp_ == 3 && c.id == rgb[i]) rgb_ids_++;
        int q = get8();
        c.h = q >> 4; c.v = q & 15;
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("bad JPEG sampling factor");
        c.tq = get8();
        if (c.tq > 3) return fail("bad JPEG quantisation table index");
    }
    hmax_ = vmax_ = 1;
    for (int i = 0; i < ncomp_; i++) { if (comp_[i].h > hmax_) hmax_ = comp_[i].h; if (comp_[i].v > vmax_) vmax_ = comp_[i].v; }
    for (int i = 0; i < ncomp_; i++)
        if (hmax_ % comp_[i].h != 0 || vmax_ % comp_[i].v != 0) return fail("bad JPEG sampling factor");
    const int mcu_w = hmax_ * 8, mcu_h = vmax_ * 8;
    mcus_x_ = (width_ + mcu_w - 1) / mcu_w;
    mcus_y_ = (height_ + mcu_h - 1) / mcu_h;
    for (int i = 0; i < ncomp_; i++) {
        Component& c = comp_[i];
        c.width = (width_ * c.h + hmax_ - 1) / hmax_;
        c.height = (height_ * c.v + vmax_ - 1) / vmax_;
        c.blocks_w = mcus_x_ * c.h;
        c.blocks_h = mcus_y_ * c.v;
        c.stride = c.blocks_w * 8;
        c.plane.assign((size_t)c.stride * c.blocks_h * 8, 0);
        if (progressive_) c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
    }
    return true;
}

bool Decoder::read_scan_header() {
    int len = get16();
    scan_n_ = get8();
    if (scan_n_ < 1 || scan_n_ > 4 || scan_n_ > ncomp_) return fail("bad SOS component count");
    if (len != 6 + 2 * scan_n_) return fail("bad SOS length");
    for (int i = 0; i < scan_n_; i++) {
        int id = get8(), q = get8(), which = 0;
        while (which < ncomp_ && comp_[which].id != id) which++;
        if (which == ncomp_) return fail("SOS names an unknown component");
        comp_[which].dc_table = q >> 4;
        comp_[which].ac_table = q & 15;
        if (comp_[which].dc_table > 3 || comp_[which].ac_table > 3) return fail("bad Huffman table index");
        order_[i] = which;
    }
    ss_ = get8();
    se_ = get8();
    int a = get8();
    ah_ = a >> 4; al_ = a & 15;
    if (progressive_) {
        if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) return fail("bad SOS spectral selection");
    } else {
        if (ss_ != 0 || ah_ != 0 || al_ != 0) return fail("bad SOS spectral selection");
        se_ = 63;
    }
    return true;
}

// ---------------------------------------------------------------- entropy decoding ------
bool Decoder::block_baseline(Component& c, int16_t block[64]) {
    const HuffmanTable& hd = dc_[c.dc_table];
    const HuffmanTable& ha = ac_[c.ac_table];
    const uint16_t* q = quant_[c.tq];
    int t = decode_symbol(hd);
    if (t < 0 || t > 15) return fail("bad Huffman code");
    std::memset(block, 0, 64 * sizeof(int16_t));
    c.dc_pred += receive_extend(t);
    block[0] = (int16_t)(c.dc_pred * q[0]);
    int k = 1;
    do {
        int rs = decode_symbol(ha);
        if (rs < 0) return fail("bad Huffman code");
        int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (rs != 0xF0) break;                                 // end of block
            k += 16;
        } else {
            k += r;
            int zig = kZigzag[k++];
            block[zig] = (int16_t)(receive_extend(s) * q[zig]);
        }
    } while (k < 64);
    return true;
}

bool Decoder::block_dc_progressive(Component& c, int16_t* block) {
    if (se_ != 0) return fail("DC and AC coefficients cannot share a progressive scan");
    if (ah_ == 0) {
        std::memset(block, 0, 64 * sizeof(int16_t));
        int t = decode_symbol(dc_[c.dc_table]);
        if (t < 0 || t > 15) return fail("bad Huffman code");
        c.dc_pred += receive_extend(t);
        block[0] = (int16_t)(c.dc_pred * (1 << al_));
    } else if (get_bit()) {
        block[0] = (int16_t)(block[0] + (1 << al_));
    }
    return true;
}

bool Decoder::block_ac_progressive(Component& c, int16_t* block) {
    if (ss_ == 0) return fail("DC and AC coefficients cannot share a progressive scan");
    const HuffmanTable& ha = ac_[c.ac_table];
    if (ah_ == 0) {                                                // first pass over this band (G.1.2.2)
        if (eob_run_) { eob_run_--; return true; }
        int k = ss_;
        do {
            int rs = decode_symbol(ha);
            if (rs < 0) return fail("bad Huffman code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run_ = 1 << r;
                    if (r) eob_run_ += get_bits(r);
                    eob_run_--;
                    break;
                }
                k += 16;
            } else {
                k += r;
                int zig = kZigzag[k++];
                block[zig] = (int16_t)(receive_extend(s) * (1 << al_));
            }
        } while (k <= se_);
        return true;
    }
    // correction pass (G.1.2.3)
    const int16_t bit = (int16_t)(1 << al_);
    auto correct = [&](int16_t& v) {
        if (get_bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    };
    if (eob_run_) {
        eob_run_--;
        for (int k = ss_; k <= se_; k++) {
            int16_t& v = block[kZigzag[k]];
            if (v != 0) correct(v);
        }
        return true;
    }
    int k = ss_;
    do {
        int rs = decode_symbol(ha);
        if (rs < 0) return fail("bad Huffman code");
        int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (r < 15) {
                eob_run_ = (1 << r) - 1;
                if (r) eob_run_ += get_bits(r);
                r = 64;                                            // run to the end of the band
            }
        } else {
            if (s != 1) return fail("bad Huffman code");
            s = get_bit() ? bit : -bit;
        }
        while (k <= se_) {
            int16_t& v = block[kZigzag[k++]];
            if (v != 0) {
                correct(v);
            } else {
                if (r == 0) { v = (int16_t)s; break; }
                r--;
            }
        }
    } while (k <= se_);
    return true;
}

// Called after every MCU; true when decoding of the scan goes on.
bool Decoder::restart_due() {
    if (--todo_ > 0) return true;
    if (bit_count_ < 24) refill();
    if (!(pending_marker_ >= 0xD0 && pending_marker_ <= 0xD7)) return false;   // no restart marker: the scan ends here
    reset_bits();
    for (int i = 0; i < 4; i++) comp_[i].dc_pred = 0;
    eob_run_ = 0;
    todo_ = restart_interval_ ? restart_interval_ : 0x7fffffff;
    return true;
}

bool Decoder::decode_scan() {
    reset_bits();
    for (int i = 0; i < 4; i++) comp_[i].dc_pred = 0;
    eob_run_ = 0;
    todo_ = restart_interval_ ? restart_interval_ : 0x7fffffff;
    for (int i = 0; i < scan_n_; i++) {
        const Component& c = comp_[order_[i]];
        const bool need_dc = !progressive_ || ss_ == 0, need_ac = !progressive_ || ss_ != 0;
        if (need_dc && !(progressive_ && ah_ != 0) && !dc_[c.dc_table].present) return fail("scan uses an undefined Huffman table");
        if (need_ac && !ac_[c.ac_table].present) return fail("scan uses an undefined Huffman table");
    }
    int16_t block[64];
    if (scan_n_ == 1) {
        // one component: its blocks in raster order, only those that carry image content
        Component& c = comp_[order_[0]];
        const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
        for (int j = 0; j < bh; j++) {
            for (int i = 0; i < bw; i++) {
                if (!progressive_) {
                    if (!block_baseline(c, block)) return false;
                    idct_block(&c.plane[(size_t)c.stride * j * 8 + i * 8], c.stride, block);
                } else {
                    int16_t* b = &c.coef[64 * ((size_t)i + (size_t)j * c.blocks_w)];
                    if (!(ss_ == 0 ? block_dc_progressive(c, b) : block_ac_progressive(c, b))) return false;
                }
                if (!restart_due()) return true;
            }
        }
        return true;
    }
    for (int j = 0; j < mcus_y_; j++) {
        for (int i = 0; i < mcus_x_; i++) {
            for (int k = 0; k < scan_n_; k++) {
                Component& c = comp_[order_[k]];
                for (int y = 0; y < c.v; y++) {
                    for (int x = 0; x < c.h; x++) {
                        const int bx = i * c.h + x, by = j * c.v + y;
                        if (!progressive_) {
                            if (!block_baseline(c, block)) return false;
                            idct_block(&c.plane[(size_t)c.stride * by * 8 + bx * 8], c.stride, block);
                        } else {
                            if (!block_dc_progressive(c, &c.coef[64 * ((size_t)bx + (size_t)by * c.blocks_w)])) return false;
                        }
                    }
                }
            }
            if (!restart_due()) return true;
        }
    }
    return true;
}

void Decoder::finish_progressive() {
    for (int n = 0; n < ncomp_; n++) {
        Component& c = comp_[n];
        const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
        for (int j = 0; j < bh; j++) {
            for (int i = 0; i < bw; i++) {
                int16_t* b = &c.coef[64 * ((size_t)i + (size_t)j * c.blocks_w)];
                for (int k = 0; k < 64; k++) b[k] = (int16_t)(b[k] * quant_[c.tq][k]);
                idct_block(&c.plane[(size_t)c.stride * j * 8 + i * 8], c.stride, b);
            }
        }
    }
}

// ---------------------------------------------------------------- upsampling, colour ----
const uint8_t* upsample_row(uint8_t* out, const uint8_t* near_row, const uint8_t* far_row, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near_row;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = near_row;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            int n = 3 * in[i] + 2;
            out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near_row[0] + far_row[0] + 2) >> 2); return out; }
        int t1 = 3 * near_row[0] + far_row[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            int t0 = t1;
            t1 = 3 * near_row[i] + far_row[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = near_row[i];
    return out;
}

inline int chroma_fixed(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

void ycbcr_row(uint8_t* out, const uint8_t* y, const uint8_t* cb_row, const uint8_t* cr_row, int count, int step) {
    for (int i = 0; i < count; i++) {
        int y_fixed = (y[i] << 20) + (1 << 19);
        int cr = cr_row[i] - 128, cb = cb_row[i] - 128;
        int r = y_fixed + cr * chroma_fixed(1.40200f);
        int g = y_fixed + (cr * -chroma_fixed(0.71414f)) + ((cb * -chroma_fixed(0.34414f)) & 0xffff0000);
        int b = y_fixed + cb * chroma_fixed(1.77200f);
        out[0] = clamp255(r >> 20);
        out[1] = clamp255(g >> 20);
        out[2] = clamp255(b >> 20);
        out[3] = 255;
        out += step;
    }
}

inline uint8_t mul255(uint8_t x, uint8_t y) {                      // rounded x*y/255
    unsigned t = x * y + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}
inline uint8_t luma(int r, int g, int b) { return (uint8_t)(((r * 77) + (g * 150) + (29 * b)) >> 8); }

bool Decoder::run(int req_comp, int& w, int& h, int& native, int& out_comp, std::vector<uint8_t>& out, std::string& err) {
    auto bail = [&]() { err = err_.empty() ? "corrupt JPEG" : err_; return false; };
    if (get8() != 0xFF || get8() != 0xD8) { err_ = "not a JPEG file"; return bail(); }
    int m = next_marker();
    while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {               // tables before the frame header
        if (!read_segment(m)) return bail();
        m = next_marker();
        while (m == -1) {
            if (eof()) { err_ = "no SOF marker"; return bail(); }
            m = next_marker();
        }
    }
    if (!read_frame(m)) return bail();
    m = next_marker();
    bool complete = true;
    while (m != 0xD9) {
        if (m == 0xDA) {
            if (!read_scan_header()) return bail();
            if (!decode_scan()) return bail();
            if (pending_marker_ < 0) {
                // junk after the entropy-coded data: look for the next thing that reads like a marker
                while (!eof()) {
                    int x = get8();
                    while (x == 0xFF) {
                        if (eof()) break;
                        x = get8();
                        if (x != 0x00 && x != 0xFF) { pending_marker_ = x; break; }
                    }
                    if (pending_marker_ >= 0) break;
                }
            }
            m = next_marker();
            if (m >= 0xD0 && m <= 0xD7) m = next_marker();
        } else if (m == 0xDC) {                                    // DNL
            int ld = get16(), nl = get16();
            if (ld != 4) { err_ = "bad DNL length"; return bail(); }
            if (nl != height_) { err_ = "bad DNL height"; return bail(); }
            m = next_marker();
        } else {
            // an unreadable segment ends the file: what was decoded so far is the image, and a
            // progressive file's coefficients are never turned into samples (stb_image.h:3440)
            if (!read_segment(m)) { complete = false; break; }
            m = next_marker();
        }
    }
    if (progressive_ && complete) finish_progressive();

    // ---- resample and convert
    const int n = req_comp ? req_comp : (ncomp_ >= 3 ? 3 : 1);
    const bool is_rgb = ncomp_ == 3 && (rgb_ids_ == 3 || (adobe_transform_ == 0 && !jfif_));
    const int decode_n = (ncomp_ == 3 && n < 3 && !is_rgb) ? 1 : ncomp_;
    struct Resample { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } rs[4];
    for (int k = 0; k < decode_n; k++) {
        Resample& r = rs[k];
        r.hs = hmax_ / comp_[k].h;
        r.vs = vmax_ / comp_[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (width_ + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = comp_[k].plane.data();
        r.buf.assign((size_t)width_ + 3 + 8, 0);
    }
    out.assign((size_t)n * width_ * height_ + 4, 0);               // rows are written 4 bytes at a time
    const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < height_; j++) {
        uint8_t* o = &out[(size_t)n * width_ * j];
        for (int k = 0; k < decode_n; k++) {
            Resample& r = rs[k];
            const bool bottom = r.ystep >= (r.vs >> 1);
            row[k] = upsample_row(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < comp_[k].height) r.line1 += comp_[k].stride;
            }
        }
        if (n >= 3) {
            if (ncomp_ == 3) {
                if (is_rgb) {
                    for (int i = 0; i < width_; i++, o += n) { o[0] = row[0][i]; o[1] = row[1][i]; o[2] = row[2][i]; o[3] = 255; }
                } else {
                    ycbcr_row(o, row[0], row[1], row[2], width_, n);
                }
            } else if (ncomp_ == 4) {
                if (adobe_transform_ == 0) {                       // CMYK
                    for (int i = 0; i < width_; i++, o += n) {
                        uint8_t k4 = row[3][i];
                        o[0] = mul255(row[0][i], k4); o[1] = mul255(row[1][i], k4); o[2] = mul255(row[2][i], k4); o[3] = 255;
                    }
                } else if (adobe_transform_ == 2) {                // YCCK
                    ycbcr_row(o, row[0], row[1], row[2], width_, n);
                    for (int i = 0; i < width_; i++, o += n) {
                        uint8_t k4 = row[3][i];
                        o[0] = mul255((uint8_t)(255 - o[0]), k4); o[1] = mul255((uint8_t)(255 - o[1]), k4); o[2] = mul255((uint8_t)(255 - o[2]), k4);
                    }
                } else {
                    ycbcr_row(o, row[0], row[1], row[2], width_, n);
                }
            } else {
                for (int i = 0; i < width_; i++, o += n) { o[0] = o[1] = o[2] = row[0][i]; o[3] = 255; }
            }
        } else {
            if (is_rgb) {
                for (int i = 0; i < width_; i++, o += n) { o[0] = luma(row[0][i], row[1][i], row[2][i]); if (n == 2) o[1] = 255; }
            } else if (ncomp_ == 4 && adobe_transform_ == 0) {
                for (int i = 0; i < width_; i++, o += n) {
                    uint8_t k4 = row[3][i];
                    o[0] = luma(mul255(row[0][i], k4), mul255(row[1][i], k4), mul255(row[2][i], k4));
                    o[1] = 255;
                }
            } else if (ncomp_ == 4 && adobe_transform_ == 2) {
                for (int i = 0; i < width_; i++, o += n) { o[0] = mul255((uint8_t)(255 - row[0][i]), row[3][i]); o[1] = 255; }
            } else {
                for (int i = 0; i < width_; i++, o += n) { o[0] = row[0][i]; if (n == 2) o[1] = 255; }
            }
        }
    }
    out.resize((size_t)n * width_ * height_);
    w = width_; h = height_;
    native = ncomp_ >= 3 ? 3 : 1;
    out_comp = n;
    return true;
}

}  // namespace

// Decodes a JPEG file image to `req_comp` interleaved 8-bit channels (0: the file's own 1 or 3).
// `native` receives the file's channel count (1 or 3), `out_comp` the count in `out`.
bool decode_jpeg(const std::vector<uint8_t>& file, int req_comp, int& w, int& h, int& native, int& out_comp,
                 std::vector<uint8_t>& out, std::string& err) {
    if (req_comp < 0 || req_comp > 4) { err = "bad channel count"; return false; }
    Decoder d(file.data(), file.size());
    return d.run(req_comp, w, h, native, out_comp, out, err);
}

}  // namespace crt
