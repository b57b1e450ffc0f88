// Baseline JPEG writer for the `-f jpg` output format (write_jpg, src/image.cuh:42-44, which
// calls stbi_write_jpg with quality 100).
//
// JFIF, 8-bit, three components, 4:4:4, Huffman-coded with the typical tables of ITU-T T.81
// Annex K; quantisation tables are the Annex K tables scaled by the usual quality rule
// (quality 100 = all ones, so the only loss left is the colour transform and DCT rounding).
// A lossy file cannot be compared byte for byte with another encoder's; the contract is that
// decoding the file returns the 8-bit pass data to within the format's rounding.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace crt {

namespace {

const uint8_t kZigzagOrder[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// T.81 Table K.1 / K.2 (natural order)
const int kLumaQuant[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                            69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                            81, 104, 113, 92, 49, 64, 78,  87,  103, 121, 120, 101, 72, 92, 95,  98,  112, 100, 103, 99};
const int kChromaQuant[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

// T.81 Tables K.3 - K.6: number of codes per length 1..16, then the symbols in code order
const uint8_t kDcLumaCounts[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaCounts[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcSymbols[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaCounts[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaSymbols[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaCounts[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaSymbols[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct CodeTable {
    uint16_t code[256];
    uint8_t length[256];
    CodeTable(const uint8_t counts[16], const uint8_t* symbols) {
        for (int i = 0; i < 256; i++) { code[i] = 0; length[i] = 0; }
        int c = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            for (int i = 0; i < counts[len - 1]; i++, k++) { code[symbols[k]] = (uint16_t)c++; length[symbols[k]] = (uint8_t)len; }
            c <<= 1;
        }
    }
};

struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int bits = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(int value, int count) {
        acc = (acc << count) | ((uint32_t)value & ((1u << count) - 1u));
        bits += count;
        while (bits >= 8) {
            uint8_t b = (uint8_t)(acc >> (bits - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);                       // byte stuffing
            bits -= 8;
        }
    }
    void flush() { if (bits) put(0x7F, 8 - bits); }                // pad with ones
};

void put16(std::vector<uint8_t>& v, int x) { v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }

void put_table(std::vector<uint8_t>& v, int cls_id, const uint8_t counts[16], const uint8_t* symbols, int n) {
    v.push_back((uint8_t)cls_id);
    v.insert(v.end(), counts, counts + 16);
    v.insert(v.end(), symbols, symbols + n);
}

// magnitude category and the bits that follow the Huffman code (T.81 F.1.2.1)
inline void magnitude(int v, int& size, int& bits) {
    int a = v < 0 ? -v : v;
    size = 0;
    while (a) { size++; a >>= 1; }
    bits = v < 0 ? v - 1 : v;
}

}  // namespace

bool write_jpeg(const char* filename, int w, int h, const uint8_t* rgb, int quality) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    int qt[2][64];
    for (int i = 0; i < 64; i++) {
        int l = (kLumaQuant[i] * scale + 50) / 100, c = (kChromaQuant[i] * scale + 50) / 100;
        qt[0][i] = l < 1 ? 1 : l > 255 ? 255 : l;
        qt[1][i] = c < 1 ? 1 : c > 255 ? 255 : c;
    }
    std::vector<uint8_t> f;
    f.reserve((size_t)w * h);
    static const uint8_t head[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    f.insert(f.end(), head, head + sizeof(head));
    f.push_back(0xFF); f.push_back(0xDB); put16(f, 2 + 2 * 65);
    for (int t = 0; t < 2; t++) {
        f.push_back((uint8_t)t);
        for (int i = 0; i < 64; i++) f.push_back((uint8_t)qt[t][kZigzagOrder[i]]);
    }
    f.push_back(0xFF); f.push_back(0xC0); put16(f, 17); f.push_back(8); put16(f, h); put16(f, w); f.push_back(3);
    static const uint8_t comps[9] = {1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1};
    f.insert(f.end(), comps, comps + 9);
    f.push_back(0xFF); f.push_back(0xC4); put16(f, 2 + 2 * (17 + 12) + 2 * (17 + 162));
    put_table(f, 0x00, kDcLumaCounts, kDcSymbols, 12);
    put_table(f, 0x10, kAcLumaCounts, kAcLumaSymbols, 162);
    put_table(f, 0x01, kDcChromaCounts, kDcSymbols, 12);
    put_table(f, 0x11, kAcChromaCounts, kAcChromaSymbols, 162);
    static const uint8_t sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
    f.insert(f.end(), sos, sos + sizeof(sos));

    const CodeTable dc_l(kDcLumaCounts, kDcSymbols), dc_c(kDcChromaCounts, kDcSymbols);
    const CodeTable ac_l(kAcLumaCounts, kAcLumaSymbols), ac_c(kAcChromaCounts, kAcChromaSymbols);
    double basis[8][8];                                            // basis[u][x] = c(u)/2 * cos((2x+1) u pi / 16)
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) basis[u][x] = (u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);

    BitWriter bw(f);
    int pred[3] = {0, 0, 0};
    for (int by = 0; by < h; by += 8) {
        for (int bx = 0; bx < w; bx += 8) {
            double plane[3][64];
            for (int y = 0; y < 8; y++) {
                for (int x = 0; x < 8; x++) {
                    const int sy = by + y < h ? by + y : h - 1, sx = bx + x < w ? bx + x : w - 1;   // edge blocks repeat the border
                    const uint8_t* p = rgb + ((size_t)sy * w + sx) * 3;
                    const double r = p[0], g = p[1], b = p[2];
                    plane[0][y * 8 + x] = 0.299 * r + 0.587 * g + 0.114 * b - 128.0;
                    plane[1][y * 8 + x] = -0.168736 * r - 0.331264 * g + 0.5 * b;
                    plane[2][y * 8 + x] = 0.5 * r - 0.418688 * g - 0.081312 * b;
                }
            }
            for (int c = 0; c < 3; c++) {
                double tmp[64], coef[64];
                for (int y = 0; y < 8; y++)
                    for (int u = 0; u < 8; u++) {
                        double s = 0;
                        for (int x = 0; x < 8; x++) s += plane[c][y * 8 + x] * basis[u][x];
                        tmp[y * 8 + u] = s;
                    }
                for (int v = 0; v < 8; v++)
                    for (int u = 0; u < 8; u++) {
                        double s = 0;
                        for (int y = 0; y < 8; y++) s += tmp[y * 8 + u] * basis[v][y];
                        coef[v * 8 + u] = s;
                    }
                const int* q = qt[c ? 1 : 0];
                int zz[64];
                for (int i = 0; i < 64; i++) {
                    const double v = coef[kZigzagOrder[i]] / q[kZigzagOrder[i]];
                    zz[i] = (int)(v < 0 ? v - 0.5 : v + 0.5);
                }
                const CodeTable& dct = c ? dc_c : dc_l;
                const CodeTable& act = c ? ac_c : ac_l;
                int size, bits;
                magnitude(zz[0] - pred[c], size, bits);
                pred[c] = zz[0];
                bw.put(dct.code[size], dct.length[size]);
                if (size) bw.put(bits, size);
                int last = 63;
                while (last > 0 && zz[last] == 0) last--;
                int run = 0;
                for (int i = 1; i <= last; i++) {
                    if (zz[i] == 0) { run++; continue; }
                    while (run >= 16) { bw.put(act.code[0xF0], act.length[0xF0]); run -= 16; }
                    magnitude(zz[i], size, bits);
                    const int sym = (run << 4) | size;
                    bw.put(act.code[sym], act.length[sym]);
                    bw.put(bits, size);
                    run = 0;
                }
                if (last != 63) bw.put(act.code[0], act.length[0]);  // end of block
            }
        }
    }
    bw.flush();
    f.push_back(0xFF); f.push_back(0xD9);
    FILE* fp = std::fopen(filename, "wb");
    if (!fp) return false;
    const bool ok = std::fwrite(f.data(), 1, f.size(), fp) == f.size();
    return std::fclose(fp) == 0 && ok;
}

}  // namespace crt
