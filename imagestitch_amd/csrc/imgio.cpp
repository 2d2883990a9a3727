// SURVEY §8(f) N4 — the on-disk format either side of the hot path: imread / imwrite of .bmp files
// (W:166 `imread(img_names[i])`, W:155-156,315 `imwrite("....bmp", ...)`; the reference's committed artefacts are BMPs).
// Host code only: uncompressed Windows bitmaps, 24-bit BGR and 8-bit paletted, bottom-up or top-down.
//   read : what cv::imread(path) (IMREAD_COLOR) returns — always 3 channels BGR, palettes expanded
//   write: what cv::imwrite does for CV_8UC3 (24-bit) and CV_8UC1 (8-bit, grey palette), rows padded to 4 bytes
// and imwrite of .jpg (B:1132 / S:1282 / W:315 `imwrite("pano.jpg", result)`): a baseline sequential JFIF encoder - 8-bit,
// Huffman, the Annex K tables, OpenCV's defaults (quality 95, 4:2:0 chroma for colour images).  The file is what any JPEG
// decoder reads; it is not libjpeg's byte stream (that would take its integer DCT and rounding).  Reading .jpg is not implemented.
#include <hip/hip_runtime.h>

#include <cmath>
#include <climits>
#include <cstdio>
#include <cstring>
#include <vector>

#include "isx_internal.hpp"

using namespace isx;

namespace {

struct BmpInfo {
    int width = 0, height = 0, bpp = 0;
    bool top_down = false;
    unsigned data_off = 0, ncolors = 0, palette_off = 0;
};

unsigned rd16(const unsigned char* p) { return p[0] | (p[1] << 8); }
unsigned rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
void wr16(unsigned char* p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }
void wr32(unsigned char* p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }

int parse_header(FILE* f, const char* path, BmpInfo& bi) {
    unsigned char h[54];
    ISX_CHECK_ARG(fread(h, 1, 54, f) == 54, ISX_ERR_INVALID, "imread: %s is shorter than a BMP header", path);
    ISX_CHECK_ARG(h[0] == 'B' && h[1] == 'M', ISX_ERR_UNSUPPORTED, "imread: %s is not a BMP file (only .bmp is implemented)", path);
    bi.data_off = rd32(h + 10);
    const unsigned hdr = rd32(h + 14);
    ISX_CHECK_ARG(hdr >= 40, ISX_ERR_UNSUPPORTED, "imread: %s: BITMAPCOREHEADER files are not supported", path);
    bi.width = (int)rd32(h + 18);
    const int hh = (int)rd32(h + 22);
    bi.top_down = hh < 0;
    bi.height = hh == INT_MIN ? 0 : (hh < 0 ? -hh : hh);      // INT_MIN has no negation: refused below as a bad size
    bi.bpp = (int)rd16(h + 28);
    const unsigned comp = rd32(h + 30);
    bi.ncolors = rd32(h + 46);
    bi.palette_off = 14 + hdr;
    ISX_CHECK_ARG(rd16(h + 26) == 1 && comp == 0 && (bi.bpp == 24 || bi.bpp == 8 || bi.bpp == 32), ISX_ERR_UNSUPPORTED,
                  "imread: %s: only uncompressed 8 / 24 / 32-bit bitmaps are supported (bpp %d, compression %u)", path, bi.bpp, comp);
    ISX_CHECK_ARG(bi.width > 0 && bi.height > 0 && bi.width < (1 << 24) && bi.height < (1 << 24), ISX_ERR_INVALID, "imread: %s: bad size %d x %d", path,
                  bi.width, bi.height);
    if (bi.bpp == 8 && bi.ncolors == 0) bi.ncolors = 256;
    return ISX_OK;
}

// ---- baseline JPEG writer -----------------------------------------------------------------------
// ITU-T T.81 Annex K tables
const unsigned char kQLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                  18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const unsigned char kQChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const unsigned char kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const unsigned char kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const unsigned char kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const unsigned char kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const unsigned char kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const unsigned char kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const unsigned char kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Huff { unsigned short code[256]; unsigned char len[256]; };
void build_huff(const unsigned char bits[16], const unsigned char* vals, Huff& h) {
    memset(&h, 0, sizeof(h));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) { h.code[vals[k]] = (unsigned short)code++; h.len[vals[k]] = (unsigned char)l; }
        code <<= 1;
    }
}

struct BitWriter {
    std::vector<unsigned char>& out;
    unsigned long long acc = 0;
    int nbits = 0;
    explicit BitWriter(std::vector<unsigned char>& o) : out(o) {}
    void put(unsigned code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        nbits += len;
        while (nbits >= 8) {
            const unsigned char b = (unsigned char)(acc >> (nbits - 8));
            out.push_back(b);
            if (b == 0xff) out.push_back(0);      // byte stuffing
            nbits -= 8;
        }
    }
    void flush() { if (nbits > 0) put(0x7f, 8 - nbits); }   // pad with ones
};

// 8x8 forward DCT (separable, double accumulation), input already level-shifted
void fdct8x8(const float in[64], float out[64]) {
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 8; ++u)
            for (int x = 0; x < 8; ++x) c[u][x] = (float)((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int u = 0; u < 8; ++u) {
            double a = 0;
            for (int x = 0; x < 8; ++x) a += (double)c[u][x] * in[8 * y + x];
            tmp[8 * y + u] = (float)a;
        }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) {
            double a = 0;
            for (int y = 0; y < 8; ++y) a += (double)c[v][y] * tmp[8 * y + u];
            out[8 * v + u] = (float)a;
        }
}

void encode_block(BitWriter& bw, const float px[64], const unsigned char q[64], int& dc_pred, const Huff& hdc, const Huff& hac) {
    float f[64];
    fdct8x8(px, f);
    int z[64];
    for (int i = 0; i < 64; ++i) z[i] = (int)std::lrint(f[kZigzag[i]] / (float)q[kZigzag[i]]);
    auto magnitude = [](int v, int& nb, unsigned& bits) {
        int a = v < 0 ? -v : v;
        nb = 0;
        while (a) { ++nb; a >>= 1; }
        bits = (unsigned)(v < 0 ? v - 1 : v) & ((1u << nb) - 1u);
    };
    int nb; unsigned bits;
    const int diff = z[0] - dc_pred;
    dc_pred = z[0];
    magnitude(diff, nb, bits);
    bw.put(hdc.code[nb], hdc.len[nb]);
    if (nb) bw.put(bits, nb);
    int run = 0;
    for (int i = 1; i < 64; ++i) {
        if (z[i] == 0) { ++run; continue; }
        while (run > 15) { bw.put(hac.code[0xf0], hac.len[0xf0]); run -= 16; }
        magnitude(z[i], nb, bits);
        const int sym = (run << 4) | nb;
        bw.put(hac.code[sym], hac.len[sym]);
        bw.put(bits, nb);
        run = 0;
    }
    if (run) bw.put(hac.code[0], hac.len[0]);   // EOB
}

void put_marker(std::vector<unsigned char>& o, unsigned m, size_t payload_len) {
    o.push_back(0xff); o.push_back((unsigned char)m);
    o.push_back((unsigned char)((payload_len + 2) >> 8)); o.push_back((unsigned char)((payload_len + 2) & 255));
}

}  // namespace

extern "C" {

int isx_jpeg_write(const char* path, const isx_mat* img, int quality) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(path != nullptr, ISX_ERR_INVALID, "imwrite: null path");
    ISX_TRY(check_mat(img, "imwrite: img"));
    ISX_CHECK_ARG(img->type == ISX_8UC3 || img->type == ISX_8UC1, ISX_ERR_TYPE, "imwrite: img must be CV_8UC3 or CV_8UC1, got %s", type_name(img->type));
    ISX_CHECK_ARG(quality >= 1 && quality <= 100, ISX_ERR_INVALID, "imwrite: JPEG quality %d not in 1..100", quality);
    ISX_CHECK_ARG(img->cols < 65536 && img->rows < 65536, ISX_ERR_UNSUPPORTED, "imwrite: %d x %d does not fit a JPEG frame", img->cols, img->rows);
    const int cn = img->type == ISX_8UC3 ? 3 : 1, W = img->cols, H = img->rows;
    const size_t dense = (size_t)W * cn;
    std::vector<unsigned char> host;
    const unsigned char* base = (const unsigned char*)img->data;
    size_t step = img->step;
    if (img->device >= 0) {
        host.resize(dense * H);
        ISX_HIP(hipSetDevice(img->device));
        ISX_HIP(hipMemcpy2D(host.data(), dense, img->data, img->step, dense, H, hipMemcpyDeviceToHost));
        base = host.data(); step = dense;
    }
    // quantisation tables scaled as libjpeg's jpeg_set_quality does
    unsigned char ql[64], qc[64];
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        int a = (kQLuma[i] * scale + 50) / 100, b = (kQChroma[i] * scale + 50) / 100;
        ql[i] = (unsigned char)(a < 1 ? 1 : (a > 255 ? 255 : a));
        qc[i] = (unsigned char)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    Huff hdl, hal, hdc, hac;
    build_huff(kDcLumaBits, kDcVals, hdl); build_huff(kAcLumaBits, kAcLumaVals, hal);
    build_huff(kDcChromaBits, kDcVals, hdc); build_huff(kAcChromaBits, kAcChromaVals, hac);

    std::vector<unsigned char> o;
    o.reserve((size_t)W * H / 2 + 1024);
    o.push_back(0xff); o.push_back(0xd8);                                                      // SOI
    put_marker(o, 0xe0, 14);                                                                   // APP0 JFIF 1.01, no density
    const unsigned char jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    o.insert(o.end(), jfif, jfif + 14);
    for (int t = 0; t < (cn == 3 ? 2 : 1); ++t) {                                              // DQT (zigzag order)
        put_marker(o, 0xdb, 65);
        o.push_back((unsigned char)t);
        for (int i = 0; i < 64; ++i) o.push_back((t ? qc : ql)[kZigzag[i]]);
    }
    put_marker(o, 0xc0, 6 + 3 * cn);                                                           // SOF0
    o.push_back(8); o.push_back((unsigned char)(H >> 8)); o.push_back((unsigned char)(H & 255)); o.push_back((unsigned char)(W >> 8)); o.push_back((unsigned char)(W & 255));
    o.push_back((unsigned char)cn);
    if (cn == 3) { const unsigned char c[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1}; o.insert(o.end(), c, c + 9); }   // Y 2x2, Cb, Cr 1x1
    else { const unsigned char c[3] = {1, 0x11, 0}; o.insert(o.end(), c, c + 3); }
    struct { int cls_id; const unsigned char* bits; const unsigned char* vals; int n; } dht[4] = {
        {0x00, kDcLumaBits, kDcVals, 12}, {0x10, kAcLumaBits, kAcLumaVals, 162}, {0x01, kDcChromaBits, kDcVals, 12}, {0x11, kAcChromaBits, kAcChromaVals, 162}};
    for (int t = 0; t < (cn == 3 ? 4 : 2); ++t) {                                              // DHT
        put_marker(o, 0xc4, 1 + 16 + (size_t)dht[t].n);
        o.push_back((unsigned char)dht[t].cls_id);
        o.insert(o.end(), dht[t].bits, dht[t].bits + 16);
        o.insert(o.end(), dht[t].vals, dht[t].vals + dht[t].n);
    }
    put_marker(o, 0xda, 4 + 2 * cn);                                                           // SOS
    o.push_back((unsigned char)cn);
    if (cn == 3) { const unsigned char c[6] = {1, 0x00, 2, 0x11, 3, 0x11}; o.insert(o.end(), c, c + 6); }
    else { o.push_back(1); o.push_back(0x00); }
    o.push_back(0); o.push_back(63); o.push_back(0);

    BitWriter bw(o);
    auto px = [&](int x, int y, int c) -> float {   // edge replication for partial blocks
        x = x < W ? x : W - 1; y = y < H ? y : H - 1;
        return (float)base[(size_t)y * step + (size_t)x * cn + c];
    };
    if (cn == 1) {
        int pred = 0;
        float blk[64];
        for (int by = 0; by < H; by += 8)
            for (int bx = 0; bx < W; bx += 8) {
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x) blk[8 * y + x] = px(bx + x, by + y, 0) - 128.f;
                encode_block(bw, blk, ql, pred, hdl, hal);
            }
    } else {
        int pred[3] = {0, 0, 0};
        float Y[4][64], Cb[64], Cr[64], yy[16][16], cb[16][16], cr[16][16];
        for (int my = 0; my < H; my += 16)
            for (int mx = 0; mx < W; mx += 16) {
                for (int y = 0; y < 16; ++y)
                    for (int x = 0; x < 16; ++x) {   // cv::Mat is BGR; JFIF YCbCr
                        const float b = px(mx + x, my + y, 0), g = px(mx + x, my + y, 1), r = px(mx + x, my + y, 2);
                        yy[y][x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
                        cb[y][x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                        cr[y][x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                    }
                for (int k = 0; k < 4; ++k)
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) Y[k][8 * y + x] = yy[8 * (k >> 1) + y][8 * (k & 1) + x];
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x) {
                        Cb[8 * y + x] = 0.25f * (cb[2 * y][2 * x] + cb[2 * y][2 * x + 1] + cb[2 * y + 1][2 * x] + cb[2 * y + 1][2 * x + 1]);
                        Cr[8 * y + x] = 0.25f * (cr[2 * y][2 * x] + cr[2 * y][2 * x + 1] + cr[2 * y + 1][2 * x] + cr[2 * y + 1][2 * x + 1]);
                    }
                for (int k = 0; k < 4; ++k) encode_block(bw, Y[k], ql, pred[0], hdl, hal);
                encode_block(bw, Cb, qc, pred[1], hdc, hac);
                encode_block(bw, Cr, qc, pred[2], hdc, hac);
            }
    }
    bw.flush();
    o.push_back(0xff); o.push_back(0xd9);                                                      // EOI
    FILE* f = fopen(path, "wb");
    ISX_CHECK_ARG(f != nullptr, ISX_ERR_INVALID, "imwrite: cannot create %s", path);
    bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    ok = (fclose(f) == 0) && ok;
    ISX_CHECK_ARG(ok, ISX_ERR_INVALID, "imwrite: short write to %s", path);
    return ISX_OK;
} ISX_EXIT("isx_jpeg_write")

int isx_bmp_size(const char* path, int* rows, int* cols) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(path && rows && cols, ISX_ERR_INVALID, "isx_bmp_size: null argument");
    FILE* f = fopen(path, "rb");
    ISX_CHECK_ARG(f != nullptr, ISX_ERR_INVALID, "imread: cannot open %s", path);
    BmpInfo bi;
    int rc = parse_header(f, path, bi);
    fclose(f);
    if (rc != ISX_OK) return rc;
    *rows = bi.height; *cols = bi.width;
    return ISX_OK;
} ISX_EXIT("isx_bmp_size")

int isx_bmp_read(const char* path, isx_mat* out) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(path != nullptr, ISX_ERR_INVALID, "imread: null path");
    ISX_TRY(check_mat(out, "imread: out"));
    ISX_CHECK_ARG(out->type == ISX_8UC3, ISX_ERR_TYPE, "imread: out must be CV_8UC3 (IMREAD_COLOR), got %s", type_name(out->type));
    FILE* f = fopen(path, "rb");
    ISX_CHECK_ARG(f != nullptr, ISX_ERR_INVALID, "imread: cannot open %s", path);
    BmpInfo bi;
    int rc = parse_header(f, path, bi);
    if (rc != ISX_OK) { fclose(f); return rc; }
    if (out->rows != bi.height || out->cols != bi.width) {
        fclose(f);
        return fail(ISX_ERR_SIZE, "imread: out is %dx%d, %s is %dx%d", out->cols, out->rows, path, bi.width, bi.height);
    }
    unsigned char pal[256 * 4];
    memset(pal, 0, sizeof(pal));
    if (bi.bpp == 8) {
        const unsigned n = bi.ncolors > 256 ? 256 : bi.ncolors;
        if (fseek(f, (long)bi.palette_off, SEEK_SET) != 0 || fread(pal, 4, n, f) != n) { fclose(f); return fail(ISX_ERR_INVALID, "imread: %s: truncated palette", path); }
    }
    const size_t src_row = ((size_t)bi.width * bi.bpp / 8 + 3) & ~(size_t)3;
    std::vector<unsigned char> row(src_row), host;
    const size_t dense = (size_t)bi.width * 3;
    unsigned char* base = (unsigned char*)out->data;
    size_t step = out->step;
    if (out->device >= 0) { host.resize(dense * bi.height); base = host.data(); step = dense; }
    if (fseek(f, (long)bi.data_off, SEEK_SET) != 0) { fclose(f); return fail(ISX_ERR_INVALID, "imread: %s: bad pixel data offset", path); }
    for (int i = 0; i < bi.height; ++i) {
        if (fread(row.data(), 1, src_row, f) != src_row) { fclose(f); return fail(ISX_ERR_INVALID, "imread: %s: truncated pixel data", path); }
        unsigned char* d = base + (size_t)(bi.top_down ? i : bi.height - 1 - i) * step;
        if (bi.bpp == 24) memcpy(d, row.data(), dense);
        else if (bi.bpp == 32) for (int x = 0; x < bi.width; ++x) { d[3 * x] = row[4 * x]; d[3 * x + 1] = row[4 * x + 1]; d[3 * x + 2] = row[4 * x + 2]; }
        else for (int x = 0; x < bi.width; ++x) { const unsigned char* p = pal + 4 * row[x]; d[3 * x] = p[0]; d[3 * x + 1] = p[1]; d[3 * x + 2] = p[2]; }
    }
    fclose(f);
    if (out->device >= 0) {
        ISX_HIP(hipSetDevice(out->device));
        ISX_HIP(hipMemcpy2D(out->data, out->step, host.data(), dense, dense, bi.height, hipMemcpyHostToDevice));
    }
    return ISX_OK;
} ISX_EXIT("isx_bmp_read")

int isx_bmp_write(const char* path, const isx_mat* img) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(path != nullptr, ISX_ERR_INVALID, "imwrite: null path");
    ISX_TRY(check_mat(img, "imwrite: img"));
    ISX_CHECK_ARG(img->type == ISX_8UC3 || img->type == ISX_8UC1, ISX_ERR_TYPE, "imwrite: img must be CV_8UC3 or CV_8UC1, got %s", type_name(img->type));
    const int cn = img->type == ISX_8UC3 ? 3 : 1;
    const size_t dense = (size_t)img->cols * cn, file_row = (dense + 3) & ~(size_t)3;
    const unsigned pal_bytes = cn == 1 ? 1024u : 0u, off = 54u + pal_bytes;
    const unsigned long long total = (unsigned long long)off + (unsigned long long)file_row * img->rows;
    ISX_CHECK_ARG(total < (1ull << 32), ISX_ERR_UNSUPPORTED, "imwrite: %d x %d does not fit a BMP file", img->cols, img->rows);
    std::vector<unsigned char> host;
    const unsigned char* base = (const unsigned char*)img->data;
    size_t step = img->step;
    if (img->device >= 0) {
        host.resize(dense * img->rows);
        ISX_HIP(hipSetDevice(img->device));
        ISX_HIP(hipMemcpy2D(host.data(), dense, img->data, img->step, dense, img->rows, hipMemcpyDeviceToHost));
        base = host.data(); step = dense;
    }
    FILE* f = fopen(path, "wb");
    ISX_CHECK_ARG(f != nullptr, ISX_ERR_INVALID, "imwrite: cannot create %s", path);
    unsigned char h[54];
    memset(h, 0, sizeof(h));
    h[0] = 'B'; h[1] = 'M';
    wr32(h + 2, (unsigned)total); wr32(h + 10, off); wr32(h + 14, 40);
    wr32(h + 18, (unsigned)img->cols); wr32(h + 22, (unsigned)img->rows);       // positive height: bottom-up rows
    wr16(h + 26, 1); wr16(h + 28, (unsigned)(8 * cn));
    bool ok = fwrite(h, 1, 54, f) == 54;
    if (cn == 1) {
        unsigned char pal[1024];
        for (int i = 0; i < 256; ++i) { pal[4 * i] = pal[4 * i + 1] = pal[4 * i + 2] = (unsigned char)i; pal[4 * i + 3] = 0; }
        ok = ok && fwrite(pal, 1, 1024, f) == 1024;
    }
    std::vector<unsigned char> row(file_row, 0);
    for (int i = img->rows - 1; i >= 0 && ok; --i) {
        memcpy(row.data(), base + (size_t)i * step, dense);
        ok = fwrite(row.data(), 1, file_row, f) == file_row;
    }
    ok = (fclose(f) == 0) && ok;
    ISX_CHECK_ARG(ok, ISX_ERR_INVALID, "imwrite: short write to %s", path);
    return ISX_OK;
} ISX_EXIT("isx_bmp_write")

}  // extern "C"
