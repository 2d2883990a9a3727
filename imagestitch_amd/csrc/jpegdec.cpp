// jpegdec.cpp — cv::imread for .jpg ("BMP/JPEG read/write"; the reference's own main()s read .bmp, W:166-169, and write
// pano.jpg, S:1282): a baseline / extended-sequential Huffman JPEG decoder on the host.
//
// cv::imread hands a JPEG to libjpeg with its defaults, so "what imread returns" is libjpeg's arithmetic, which is restated here from the
// published algorithm (IJG libjpeg 6b, unchanged in libjpeg-turbo): dequantisation + the accurate integer inverse DCT (jidctint.c:
// Loeffler-Ligtenberg-Moschytz, CONST_BITS 13, PASS1_BITS 2), "fancy" triangle-filter upsampling of h2v1 / h2v2 chroma (jdsample.c: 3/4 + 1/4
// with the alternating rounding bias, edge samples replicated), pixel replication for other ratios, and the fixed-point YCbCr -> RGB of
// jdcolor.c (16 fractional bits).  tests/test_imgio.py pins it bit for bit to Pillow's libjpeg-turbo on 4:4:4 / 4:2:2 / 4:2:0 / grey files
// of odd sizes, with and without restart intervals.  Progressive files (SOF2: spectral selection and successive approximation, jdphuff.c)
// are decoded into the same coefficient arrays scan by scan and then take the same inverse DCT / upsampling / colour conversion; a
// progressive file whose scans do not bring every coefficient to full precision (libjpeg would smooth its blocks) is refused.
// Not supported (ISX_ERR_UNSUPPORTED): arithmetic-coded and lossless files, 12-bit samples, CMYK.  Output: CV_8UC3 BGR (IMREAD_COLOR),
// host or device mat.
#include "isx_internal.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

using namespace isx;

namespace {

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int wblk = 0, hblk = 0; int dc_pred = 0; std::vector<short> coef; std::vector<unsigned char> plane; int pw = 0, ph = 0;
    signed char coef_bits[64];      // progressive: the point transform each coefficient (zigzag index) was last sent with, -1 = never (jdphuff.c)
};

struct HuffDec {
    bool present = false;
    int maxcode[18]; int valptr[17]; int mincode[17];
    unsigned char vals[256];
    unsigned char look_nbits[256], look_sym[256];     // 8-bit lookahead
};

struct Jpeg {
    std::vector<unsigned char> data;
    size_t pos = 0;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    int mcux = 0, mcuy = 0;               // MCUs per row / column of the frame (set with the frame header)
    Comp comp[4];
    unsigned short qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    HuffDec dc[4], ac[4];
    int restart = 0;
    bool adobe = false; int adobe_transform = 0;
    bool progressive = false, got_sof = false;
    int scans = 0;
    int ss = 0, se = 63, ah = 0, al = 0, eobrun = 0;      // the current scan's spectral band and bit position (progressive), its end-of-band run
    // bit reader
    unsigned long long bits = 0; int nbits = 0; bool hit_marker = false;
};

const unsigned char ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

int build_huff(HuffDec& h, const unsigned char counts[16], const unsigned char* vals, int nvals) {
    h.present = true;
    memcpy(h.vals, vals, (size_t)nvals);
    int code = 0, k = 0;
    int huffsize[257], huffcode[257];
    for (int l = 1; l <= 16; ++l)
        for (int i = 0; i < counts[l - 1]; ++i) huffsize[k++] = l;
    huffsize[k] = 0;
    const int n = k;
    k = 0;
    int si = huffsize[0];
    while (k < n) {
        while (k < n && huffsize[k] == si) huffcode[k++] = code++;
        if (code > (1 << si)) return -1;
        code <<= 1; ++si;
    }
    int p = 0;
    for (int l = 1; l <= 16; ++l) {
        if (counts[l - 1]) {
            h.valptr[l] = p; h.mincode[l] = huffcode[p];
            p += counts[l - 1];
            h.maxcode[l] = huffcode[p - 1];
        } else { h.maxcode[l] = -1; h.valptr[l] = 0; h.mincode[l] = 0; }
    }
    h.maxcode[17] = 0xFFFFF;
    memset(h.look_nbits, 0, sizeof(h.look_nbits));
    p = 0;
    for (int l = 1; l <= 8; ++l)
        for (int i = 0; i < counts[l - 1]; ++i, ++p) {
            const int look = huffcode[p] << (8 - l);
            for (int c = 0; c < (1 << (8 - l)); ++c) { h.look_nbits[look + c] = (unsigned char)l; h.look_sym[look + c] = h.vals[p]; }
        }
    return 0;
}

void fill_bits(Jpeg& j) {
    while (j.nbits <= 48) {
        unsigned c = 0;
        if (!j.hit_marker && j.pos < j.data.size()) {
            c = j.data[j.pos];
            if (c == 0xFF) {
                const unsigned n = j.pos + 1 < j.data.size() ? j.data[j.pos + 1] : 0xD9u;
                if (n == 0) j.pos += 2;                 // stuffed zero
                else { j.hit_marker = true; c = 0; }    // a marker: feed zeros (libjpeg does the same), leave pos on it
            } else ++j.pos;
        }
        j.bits = (j.bits << 8) | c; j.nbits += 8;
    }
}
inline int peek(Jpeg& j, int n) { return (int)((j.bits >> (j.nbits - n)) & ((1u << n) - 1)); }
inline int get_bits(Jpeg& j, int n) {
    if (n == 0) return 0;
    if (j.nbits < n) fill_bits(j);
    const int v = peek(j, n); j.nbits -= n; return v;
}
int decode_sym(Jpeg& j, const HuffDec& h) {
    if (j.nbits < 17) fill_bits(j);       // (the search below looks at up to 17 bits before it gives up)
    const int look = peek(j, 8);
    if (h.look_nbits[look]) { j.nbits -= h.look_nbits[look]; return h.look_sym[look]; }
    int l = 9, code = peek(j, 9);
    while (l <= 16 && code > h.maxcode[l]) { ++l; code = peek(j, l); }
    if (l > 16) return -1;
    j.nbits -= l;
    return h.vals[h.valptr[l] + code - h.mincode[l]];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }     // HUFF_EXTEND

int decode_block(Jpeg& j, Comp& c, short* blk) {
    memset(blk, 0, 64 * sizeof(short));
    int s = decode_sym(j, j.dc[c.td]);
    if (s < 0 || s > 15) return -1;
    int diff = s ? extend(get_bits(j, s), s) : 0;
    c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)diff);
    blk[0] = (short)c.dc_pred;
    for (int k = 1; k < 64;) {
        const int rs = decode_sym(j, j.ac[c.ta]);
        if (rs < 0) return -1;
        const int r = rs >> 4, ss = rs & 15;
        if (ss == 0) { if (r != 15) break; k += 16; continue; }
        k += r;
        if (k > 63) return -1;
        blk[ZIGZAG[k]] = (short)extend(get_bits(j, ss), ss);
        ++k;
    }
    return 0;
}

// ---- progressive scans (jdphuff.c): the block's coefficients persist between scans ----------------------------------------------
int prog_dc_first(Jpeg& j, Comp& c, short* blk) {
    const int s = decode_sym(j, j.dc[c.td]);
    if (s < 0 || s > 15) return -1;
    c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)(s ? extend(get_bits(j, s), s) : 0));      // (a hostile file can run the predictor past INT_MAX: wrap, as the stored short does)
    blk[0] = (short)((unsigned)c.dc_pred << j.al);
    return 0;
}
int prog_dc_refine(Jpeg& j, Comp&, short* blk) {
    if (get_bits(j, 1)) blk[0] = (short)(blk[0] | (1 << j.al));
    return 0;
}
int prog_ac_first(Jpeg& j, Comp& c, short* blk) {
    if (j.eobrun > 0) { --j.eobrun; return 0; }
    for (int k = j.ss; k <= j.se; ++k) {
        const int rs = decode_sym(j, j.ac[c.ta]);
        if (rs < 0) return -1;
        const int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return -1;
            blk[ZIGZAG[k]] = (short)(extend(get_bits(j, s), s) * (1 << j.al));
        } else if (r == 15) k += 15;
        else {
            j.eobrun = 1 << r;
            if (r) j.eobrun += get_bits(j, r);
            --j.eobrun;
            break;
        }
    }
    return 0;
}
inline void prog_correct(Jpeg& j, short& coef, int p1, int m1) {      // one correction bit for an already-nonzero coefficient
    if (get_bits(j, 1) && (coef & p1) == 0) coef = (short)(coef >= 0 ? coef + p1 : coef + m1);
}
int prog_ac_refine(Jpeg& j, Comp& c, short* blk) {
    const int p1 = 1 << j.al, m1 = -(1 << j.al);
    int k = j.ss;
    if (j.eobrun == 0) {
        for (; k <= j.se; ++k) {
            const int rs = decode_sym(j, j.ac[c.ta]);
            if (rs < 0) return -1;
            int r = rs >> 4, s = rs & 15;
            if (s) s = get_bits(j, 1) ? p1 : m1;      // (a size other than 1 is a bad code; libjpeg warns and goes on the same way)
            else if (r != 15) {
                j.eobrun = 1 << r;
                if (r) j.eobrun += get_bits(j, r);
                break;                                  // end of band
            }
            // over the already-nonzero coefficients (a correction bit each) and r still-zero ones
            do {
                short& t = blk[ZIGZAG[k]];
                if (t != 0) prog_correct(j, t, p1, m1);
                else if (--r < 0) break;
                ++k;
            } while (k <= j.se);
            if (s) {
                if (k > 63) return -1;
                blk[ZIGZAG[k]] = (short)s;
            }
        }
    }
    if (j.eobrun > 0) {
        for (; k <= j.se; ++k) {
            short& t = blk[ZIGZAG[k]];
            if (t != 0) prog_correct(j, t, p1, m1);
        }
        --j.eobrun;
    }
    return 0;
}

// jidctint.c (ISLOW): 8 x 8 inverse DCT of dequantised coefficients, output samples range-limited to 0..255
inline int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
inline unsigned char range_limit(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idct_islow(const short* coef, const unsigned short* q, unsigned char* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373, F_1_175875602 = 9633,
                   F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
    int ws[64];
    for (int x = 0; x < 8; ++x) {         // pass 1: columns
        const long d0 = (long)coef[x] * q[x], d1 = (long)coef[8 + x] * q[8 + x], d2 = (long)coef[16 + x] * q[16 + x], d3 = (long)coef[24 + x] * q[24 + x],
                   d4 = (long)coef[32 + x] * q[32 + x], d5 = (long)coef[40 + x] * q[40 + x], d6 = (long)coef[48 + x] * q[48 + x], d7 = (long)coef[56 + x] * q[56 + x];
        long z1 = (d2 + d6) * F_0_541196100;
        long tmp2 = z1 + d6 * (-F_1_847759065), tmp3 = z1 + d2 * F_0_765366865;
        long tmp0 = (d0 + d4) * (1L << CB), tmp1 = (d0 - d4) * (1L << CB);      // libjpeg's LEFT_SHIFT of a possibly negative value, as a multiplication
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = d7; tmp1 = d5; tmp2 = d3; tmp3 = d1;
        z1 = tmp0 + tmp3; long z2 = tmp1 + tmp2, z3 = tmp0 + tmp2, z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[x] = descale(tmp10 + tmp3, CB - P1); ws[56 + x] = descale(tmp10 - tmp3, CB - P1);
        ws[8 + x] = descale(tmp11 + tmp2, CB - P1); ws[48 + x] = descale(tmp11 - tmp2, CB - P1);
        ws[16 + x] = descale(tmp12 + tmp1, CB - P1); ws[40 + x] = descale(tmp12 - tmp1, CB - P1);
        ws[24 + x] = descale(tmp13 + tmp0, CB - P1); ws[32 + x] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int y = 0; y < 8; ++y) {         // pass 2: rows
        const int* w = ws + 8 * y;
        long z1 = ((long)w[2] + w[6]) * F_0_541196100;
        long tmp2 = z1 + (long)w[6] * (-F_1_847759065), tmp3 = z1 + (long)w[2] * F_0_765366865;
        long tmp0 = ((long)w[0] + w[4]) * (1L << CB), tmp1 = ((long)w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; long z2 = tmp1 + tmp2, z3 = tmp0 + tmp2, z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        unsigned char* o = out + (size_t)y * stride;
        constexpr int S = CB + P1 + 3;
        o[0] = range_limit(descale(tmp10 + tmp3, S) + 128); o[7] = range_limit(descale(tmp10 - tmp3, S) + 128);
        o[1] = range_limit(descale(tmp11 + tmp2, S) + 128); o[6] = range_limit(descale(tmp11 - tmp2, S) + 128);
        o[2] = range_limit(descale(tmp12 + tmp1, S) + 128); o[5] = range_limit(descale(tmp12 - tmp1, S) + 128);
        o[3] = range_limit(descale(tmp13 + tmp0, S) + 128); o[4] = range_limit(descale(tmp13 - tmp0, S) + 128);
    }
}

inline unsigned rd16(const Jpeg& j, size_t p) { return ((unsigned)j.data[p] << 8) | j.data[p + 1]; }

int parse(Jpeg& j, const char* path, bool header_only) {
    const size_t n = j.data.size();
    ISX_CHECK_ARG(n >= 4 && j.data[0] == 0xFF && j.data[1] == 0xD8, ISX_ERR_INVALID, "imread: %s is not a JPEG file", path);
    size_t p = 2;
    while (p + 4 <= n) {
        if (j.data[p] != 0xFF) { ++p; continue; }
        const unsigned m = j.data[p + 1];
        if (m == 0xFF) { ++p; continue; }
        if (m == 0x00) { p += 2; continue; }      // a stuffed byte of entropy-coded data
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
        const size_t len = rd16(j, p + 2);
        ISX_CHECK_ARG(len >= 2 && p + 2 + len <= n, ISX_ERR_INVALID, "imread: %s: truncated marker segment", path);
        const size_t s = p + 4, e = p + 2 + len;
        if (m == 0xDB) {                                  // DQT
            for (size_t q = s; q < e;) {
                const int pq = j.data[q] >> 4, tq = j.data[q] & 15; ++q;
                ISX_CHECK_ARG(tq < 4 && q + (pq ? 128 : 64) <= e, ISX_ERR_INVALID, "imread: %s: bad quantisation table", path);
                for (int i = 0; i < 64; ++i) { j.qt[tq][ZIGZAG[i]] = (unsigned short)(pq ? rd16(j, q) : j.data[q]); q += pq ? 2 : 1; }
                j.have_qt[tq] = true;
            }
        } else if (m == 0xC4) {                           // DHT
            for (size_t q = s; q < e;) {
                const int tc = j.data[q] >> 4, th = j.data[q] & 15; ++q;
                ISX_CHECK_ARG(tc < 2 && th < 4 && q + 16 <= e, ISX_ERR_INVALID, "imread: %s: bad Huffman table", path);
                int cnt = 0;
                for (int i = 0; i < 16; ++i) cnt += j.data[q + i];
                ISX_CHECK_ARG(cnt <= 256 && q + 16 + cnt <= e, ISX_ERR_INVALID, "imread: %s: bad Huffman table", path);
                ISX_CHECK_ARG(build_huff(tc ? j.ac[th] : j.dc[th], &j.data[q], &j.data[q + 16], cnt) == 0, ISX_ERR_INVALID, "imread: %s: bad Huffman code lengths", path);
                q += 16 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {     // SOFn
            ISX_CHECK_ARG(m == 0xC0 || m == 0xC1 || m == 0xC2, ISX_ERR_UNSUPPORTED, "imread: %s: only baseline / extended sequential / progressive Huffman JPEG is decoded (SOF%u)", path, m - 0xC0);
            j.progressive = m == 0xC2;
            ISX_CHECK_ARG(!j.got_sof, ISX_ERR_INVALID, "imread: %s: a second frame header (libjpeg: JERR_SOF_DUPLICATE)", path);
            ISX_CHECK_ARG(len >= 8 && j.data[s] == 8, ISX_ERR_UNSUPPORTED, "imread: %s: %u-bit samples", path, (unsigned)j.data[s]);
            j.height = (int)rd16(j, s + 1); j.width = (int)rd16(j, s + 3); j.ncomp = j.data[s + 5];
            ISX_CHECK_ARG(j.width > 0 && j.height > 0 && (j.ncomp == 1 || j.ncomp == 3) && s + 6 + 3 * (size_t)j.ncomp <= e, ISX_ERR_UNSUPPORTED,
                          "imread: %s: %d x %d with %d components", path, j.width, j.height, j.ncomp);
            for (int c = 0; c < j.ncomp; ++c) {
                Comp& k = j.comp[c];
                k.id = j.data[s + 6 + 3 * c]; k.h = j.data[s + 7 + 3 * c] >> 4; k.v = j.data[s + 7 + 3 * c] & 15; k.tq = j.data[s + 8 + 3 * c] & 3;
                ISX_CHECK_ARG(k.h >= 1 && k.h <= 4 && k.v >= 1 && k.v <= 4, ISX_ERR_INVALID, "imread: %s: bad sampling factors", path);
                j.hmax = std::max(j.hmax, k.h); j.vmax = std::max(j.vmax, k.v);
            }
            j.got_sof = true;
            // geometry, once and for all scans: blocks per component, padded to whole MCUs.  A frame whose MCUs could not possibly fit the
            // file (every coded block of a sequential frame takes at least two bits: an end-of-block after a zero DC difference; one bit in a progressive one) is refused before anything is
            // allocated, and so is one beyond the size the BMP reader accepts: the header of a few-hundred-byte file must not drive
            // gigabytes of allocations.
            j.mcux = (j.width + 8 * j.hmax - 1) / (8 * j.hmax); j.mcuy = (j.height + 8 * j.vmax - 1) / (8 * j.vmax);
            {
                unsigned long long blocks = 0;
                for (int c = 0; c < j.ncomp; ++c) blocks += (unsigned long long)j.mcux * j.comp[c].h * (unsigned long long)j.mcuy * j.comp[c].v;
                // (progressive frames: the DC-first scan can spend ONE bit per block - a zero difference under a 1-bit code, mozjpeg-style scan
                // scripts without DC successive approximation - and the AC scans next to nothing through EOBRUN: blocks / 8 there)
                ISX_CHECK_ARG((unsigned long long)j.width * j.height <= (1ull << 30) && blocks / (j.progressive ? 8 : 4) <= (unsigned long long)n, ISX_ERR_INVALID,
                              "imread: %s: a %d x %d frame cannot be held by a %zu-byte file", path, j.width, j.height, n);
            }
            if (header_only) return ISX_OK;      // (isx_jpeg_size: the caller allocates rows x cols from what this returns - after the check above)
            for (int c = 0; c < j.ncomp; ++c) {
                Comp& k = j.comp[c];
                k.wblk = j.mcux * k.h; k.hblk = j.mcuy * k.v;
                k.coef.assign((size_t)k.wblk * k.hblk * 64, 0);
                memset(k.coef_bits, -1, sizeof(k.coef_bits));
            }
        } else if (m == 0xDD) {
            ISX_CHECK_ARG(len >= 4, ISX_ERR_INVALID, "imread: %s: truncated DRI segment", path);
            j.restart = (int)rd16(j, s);
        } else if (m == 0xEE && len >= 14 && memcmp(&j.data[s], "Adobe", 5) == 0) { j.adobe = true; j.adobe_transform = j.data[s + 11];
        } else if (m == 0xDA) {                           // SOS: decode this scan
            ISX_CHECK_ARG(j.got_sof, ISX_ERR_INVALID, "imread: %s: scan before frame header", path);
            ISX_CHECK_ARG(len >= 3, ISX_ERR_INVALID, "imread: %s: truncated scan header", path);
            const int ns = j.data[s];
            ISX_CHECK_ARG(ns >= 1 && ns <= j.ncomp && len >= 6 + 2 * (size_t)ns, ISX_ERR_INVALID, "imread: %s: bad scan header", path);
            Comp* sc[4];
            for (int i = 0; i < ns; ++i) {
                const int id = j.data[s + 1 + 2 * i];
                Comp* k = nullptr;
                for (int c = 0; c < j.ncomp; ++c) if (j.comp[c].id == id) k = &j.comp[c];
                ISX_CHECK_ARG(k != nullptr, ISX_ERR_INVALID, "imread: %s: scan names an unknown component", path);
                k->td = j.data[s + 2 + 2 * i] >> 4; k->ta = j.data[s + 2 + 2 * i] & 15;
                ISX_CHECK_ARG(k->td < 4 && k->ta < 4, ISX_ERR_INVALID, "imread: %s: scan uses an undefined Huffman table", path);
                for (int q = 0; q < i; ++q) ISX_CHECK_ARG(sc[q] != k, ISX_ERR_INVALID, "imread: %s: scan names a component twice", path);
                sc[i] = k;
            }
            j.ss = j.data[s + 1 + 2 * ns]; j.se = j.data[s + 2 + 2 * ns]; j.ah = j.data[s + 3 + 2 * ns] >> 4; j.al = j.data[s + 3 + 2 * ns] & 15;
            int (*block_fn)(Jpeg&, Comp&, short*) = decode_block;
            if (j.progressive) {        // jdphuff.c start_pass_phuff_decoder: the scan's band and bit position must make sense
                const bool dc_scan = j.ss == 0;
                ISX_CHECK_ARG((dc_scan ? j.se == 0 : (j.se >= j.ss && j.se <= 63 && ns == 1)) && j.al <= 13 && (j.ah == 0 || j.ah == j.al + 1), ISX_ERR_INVALID,
                              "imread: %s: bad progressive scan parameters (Ss %d Se %d Ah %d Al %d, %d components)", path, j.ss, j.se, j.ah, j.al, ns);
                block_fn = dc_scan ? (j.ah ? prog_dc_refine : prog_dc_first) : (j.ah ? prog_ac_refine : prog_ac_first);
                for (int i = 0; i < ns; ++i) {
                    if (dc_scan && j.ah == 0) ISX_CHECK_ARG(j.dc[sc[i]->td].present, ISX_ERR_INVALID, "imread: %s: scan uses an undefined Huffman table", path);
                    if (!dc_scan) ISX_CHECK_ARG(j.ac[sc[i]->ta].present, ISX_ERR_INVALID, "imread: %s: scan uses an undefined Huffman table", path);
                    for (int c = j.ss; c <= j.se; ++c) sc[i]->coef_bits[c] = (signed char)j.al;
                }
            } else {
                for (int i = 0; i < ns; ++i)
                    ISX_CHECK_ARG(j.dc[sc[i]->td].present && j.ac[sc[i]->ta].present, ISX_ERR_INVALID, "imread: %s: scan uses an undefined Huffman table", path);
            }
            j.eobrun = 0;
            // a few hundred bytes of scan headers over a frame that passed the size check must not buy minutes of decoding: real encoders
            // use about ten scans (libjpeg's default script), the format's own bit positions and bands allow some hundreds
            ISX_CHECK_ARG(++j.scans <= 256, ISX_ERR_INVALID, "imread: %s: more than 256 scans", path);
            const int mcux = j.mcux, mcuy = j.mcuy;      // (blocks per component and the coefficient buffers were sized with the frame header)
            j.pos = e; j.bits = 0; j.nbits = 0; j.hit_marker = false;
            for (int c = 0; c < j.ncomp; ++c) j.comp[c].dc_pred = 0;
            int todo = j.restart;
            if (ns == 1) {                                // non-interleaved: the component's own blocks, ceil(size / 8) of them per row
                Comp& k = *sc[0];
                const int cw = (j.width * k.h + j.hmax - 1) / j.hmax, ch = (j.height * k.v + j.vmax - 1) / j.vmax;
                const int bw = (cw + 7) / 8, bh = (ch + 7) / 8;
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        if (j.restart && todo == 0) {
                            j.nbits = 0; j.bits = 0;
                            while (j.pos + 1 < n && !(j.data[j.pos] == 0xFF && j.data[j.pos + 1] >= 0xD0 && j.data[j.pos + 1] <= 0xD7)) ++j.pos;
                            j.pos += 2; j.hit_marker = false; k.dc_pred = 0; j.eobrun = 0; todo = j.restart;
                        }
                        ISX_CHECK_ARG(bx < k.wblk && by < k.hblk, ISX_ERR_INVALID, "imread: %s: scan passes the frame", path);
                        ISX_CHECK_ARG(block_fn(j, k, &k.coef[((size_t)by * k.wblk + bx) * 64]) == 0, ISX_ERR_INVALID, "imread: %s: corrupt entropy-coded data", path);
                        --todo;
                    }
            } else {
                for (int my = 0; my < mcuy; ++my)
                    for (int mx = 0; mx < mcux; ++mx) {
                        if (j.restart && todo == 0) {
                            j.nbits = 0; j.bits = 0;
                            while (j.pos + 1 < n && !(j.data[j.pos] == 0xFF && j.data[j.pos + 1] >= 0xD0 && j.data[j.pos + 1] <= 0xD7)) ++j.pos;
                            j.pos += 2; j.hit_marker = false;
                            for (int c = 0; c < j.ncomp; ++c) j.comp[c].dc_pred = 0;
                            j.eobrun = 0;
                            todo = j.restart;
                        }
                        for (int i = 0; i < ns; ++i) {
                            Comp& k = *sc[i];
                            for (int v = 0; v < k.v; ++v)
                                for (int h = 0; h < k.h; ++h)
                                    ISX_CHECK_ARG(my * k.v + v < k.hblk && mx * k.h + h < k.wblk && block_fn(j, k, &k.coef[((size_t)(my * k.v + v) * k.wblk + mx * k.h + h) * 64]) == 0, ISX_ERR_INVALID,
                                                  "imread: %s: corrupt entropy-coded data", path);
                        }
                        --todo;
                    }
            }
            p = j.pos;        // on the next marker (or in the padding before it)
            continue;
        }
        p = e;
    }
    ISX_CHECK_ARG(j.got_sof, ISX_ERR_INVALID, "imread: %s: no frame header", path);
    return ISX_OK;
}

// jdsample.c: bring a component plane (cw x ch valid samples in a pw-wide buffer) to the full image size
void upsample(const Comp& k, int cw, int ch, int hexp, int vexp, int W, int H, std::vector<unsigned char>& full) {
    full.assign((size_t)W * H, 0);
    const unsigned char* src = k.plane.data();
    const int pw = k.pw;
    auto row = [&](int y) { return src + (size_t)std::min(std::max(y, 0), ch - 1) * pw; };
    if (hexp == 1 && vexp == 1) {
        for (int y = 0; y < H; ++y) memcpy(&full[(size_t)y * W], row(y), (size_t)W);
    } else if (hexp == 2 && vexp == 1 && cw > 2) {  // h2v1_fancy_upsample (libjpeg: only when the component is more than 2 samples wide)
        for (int y = 0; y < H; ++y) {
            const unsigned char* in = row(y);
            unsigned char* o = &full[(size_t)y * W];
            std::vector<unsigned char> tmp((size_t)2 * cw);
            if (cw == 1) { tmp[0] = tmp[1] = in[0]; }
            else {
                tmp[0] = in[0]; tmp[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
                for (int x = 1; x < cw - 1; ++x) {
                    const int v = in[x] * 3;
                    tmp[2 * x] = (unsigned char)((v + in[x - 1] + 1) >> 2); tmp[2 * x + 1] = (unsigned char)((v + in[x + 1] + 2) >> 2);
                }
                tmp[2 * cw - 2] = (unsigned char)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2); tmp[2 * cw - 1] = in[cw - 1];
            }
            memcpy(o, tmp.data(), (size_t)std::min(W, 2 * cw));
        }
    } else if (hexp == 2 && vexp == 2 && cw > 2) {  // h2v2_fancy_upsample: 3/4 nearer row + 1/4 further row, then the same across columns
        std::vector<int> s0((size_t)cw);
        std::vector<unsigned char> tmp((size_t)2 * cw);
        for (int y = 0; y < H; ++y) {
            const int cy = y >> 1;
            const unsigned char* in0 = row(cy);
            const unsigned char* in1 = row((y & 1) ? cy + 1 : cy - 1);
            for (int x = 0; x < cw; ++x) s0[x] = in0[x] * 3 + in1[x];
            if (cw == 1) { tmp[0] = (unsigned char)((s0[0] * 4 + 8) >> 4); tmp[1] = (unsigned char)((s0[0] * 4 + 7) >> 4); }
            else {
                tmp[0] = (unsigned char)((s0[0] * 4 + 8) >> 4); tmp[1] = (unsigned char)((s0[0] * 3 + s0[1] + 7) >> 4);
                for (int x = 1; x < cw - 1; ++x) {
                    tmp[2 * x] = (unsigned char)((s0[x] * 3 + s0[x - 1] + 8) >> 4); tmp[2 * x + 1] = (unsigned char)((s0[x] * 3 + s0[x + 1] + 7) >> 4);
                }
                tmp[2 * cw - 2] = (unsigned char)((s0[cw - 1] * 3 + s0[cw - 2] + 8) >> 4); tmp[2 * cw - 1] = (unsigned char)((s0[cw - 1] * 4 + 7) >> 4);
            }
            memcpy(&full[(size_t)y * W], tmp.data(), (size_t)std::min(W, 2 * cw));
        }
    } else {                                        // int_upsample: replication
        for (int y = 0; y < H; ++y) {
            const unsigned char* in = row(y / vexp);
            unsigned char* o = &full[(size_t)y * W];
            for (int x = 0; x < W; ++x) o[x] = in[std::min(x / hexp, cw - 1)];
        }
    }
}

int load_file(const char* path, std::vector<unsigned char>& buf) {
    FILE* f = fopen(path, "rb");
    ISX_CHECK_ARG(f != nullptr, ISX_ERR_INVALID, "imread: cannot open %s", path);
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) { fclose(f); return fail(ISX_ERR_INVALID, "imread: %s is empty", path); }
    buf.resize((size_t)n);
    const size_t got = fread(buf.data(), 1, (size_t)n, f);
    fclose(f);
    ISX_CHECK_ARG(got == (size_t)n, ISX_ERR_INVALID, "imread: %s: read error", path);
    return ISX_OK;
}

}  // namespace

extern "C" {

// no C++ exception crosses the C boundary: an allocation that fails (a damaged header can still ask for a lot) is ISX_ERR_NOMEM
static int jpeg_size_impl(const char* path, int* rows, int* cols) {
    Jpeg j;
    ISX_TRY(load_file(path, j.data));
    ISX_TRY(parse(j, path, true));
    *rows = j.height; *cols = j.width;
    return ISX_OK;
}
static int jpeg_read_impl(const char* path, isx_mat* out);

int isx_jpeg_size(const char* path, int* rows, int* cols) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(path && rows && cols, ISX_ERR_INVALID, "isx_jpeg_size: null argument");
    try { return jpeg_size_impl(path, rows, cols); }
    catch (const std::bad_alloc&) { return fail(ISX_ERR_NOMEM, "imread: %s: out of host memory", path); }
    catch (...) { return fail(ISX_ERR_INVALID, "imread: %s: unexpected failure while decoding", path); }
} ISX_EXIT("isx_jpeg_size")

int isx_jpeg_read(const char* path, isx_mat* out) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(path != nullptr, ISX_ERR_INVALID, "imread: null path");
    try { return jpeg_read_impl(path, out); }
    catch (const std::bad_alloc&) { return fail(ISX_ERR_NOMEM, "imread: %s: out of host memory", path); }
    catch (...) { return fail(ISX_ERR_INVALID, "imread: %s: unexpected failure while decoding", path); }
} ISX_EXIT("isx_jpeg_read")

static int jpeg_read_impl(const char* path, isx_mat* out) {
    ISX_CHECK_ARG(path != nullptr, ISX_ERR_INVALID, "imread: null path");
    ISX_TRY(check_mat(out, "imread: out"));
    ISX_CHECK_ARG(out->type == ISX_8UC3, ISX_ERR_TYPE, "imread: out must be CV_8UC3 (IMREAD_COLOR), got %s", type_name(out->type));
    Jpeg j;
    ISX_TRY(load_file(path, j.data));
    ISX_TRY(parse(j, path, false));
    ISX_CHECK_ARG(out->rows == j.height && out->cols == j.width, ISX_ERR_SIZE, "imread: out is %dx%d, %s is %dx%d", out->cols, out->rows, path, j.width, j.height);
    if (j.progressive)      // every coefficient at full precision: otherwise libjpeg smooths the blocks from their neighbours' DC values (jdcoefct.c)
        for (int c = 0; c < j.ncomp; ++c)
            for (int i = 0; i < 64; ++i)
                ISX_CHECK_ARG(j.comp[c].coef_bits[i] == 0, ISX_ERR_UNSUPPORTED, "imread: %s: progressive file whose scans leave coefficient %d of component %d %s", path, i, c,
                              j.comp[c].coef_bits[i] < 0 ? "unsent" : "short of full precision");
    const int W = j.width, H = j.height;
    std::vector<unsigned char> full[3];
    for (int c = 0; c < j.ncomp; ++c) {
        Comp& k = j.comp[c];
        ISX_CHECK_ARG(!k.coef.empty() && j.have_qt[k.tq], ISX_ERR_INVALID, "imread: %s: component %d has no scan or no quantisation table", path, c);
        ISX_CHECK_ARG(j.hmax % k.h == 0 && j.vmax % k.v == 0, ISX_ERR_UNSUPPORTED, "imread: %s: fractional sampling ratios", path);
        k.pw = k.wblk * 8; k.ph = k.hblk * 8;
        k.plane.assign((size_t)k.pw * k.ph, 0);
        for (int by = 0; by < k.hblk; ++by)
            for (int bx = 0; bx < k.wblk; ++bx)
                idct_islow(&k.coef[((size_t)by * k.wblk + bx) * 64], j.qt[k.tq], &k.plane[(size_t)by * 8 * k.pw + bx * 8], k.pw);
        const int cw = (W * k.h + j.hmax - 1) / j.hmax, ch = (H * k.v + j.vmax - 1) / j.vmax;
        upsample(k, cw, ch, j.hmax / k.h, j.vmax / k.v, W, H, full[c]);
    }
    // colour conversion to BGR (jdcolor.c ycc_rgb_convert; grey: replicated; Adobe transform 0 or components 'R','G','B': RGB as stored)
    std::vector<unsigned char> host;
    const size_t dense = (size_t)W * 3;
    unsigned char* base = (unsigned char*)out->data;
    size_t step = out->step;
    if (out->device >= 0) { host.resize(dense * H); base = host.data(); step = dense; }
    const bool rgb = j.ncomp == 3 && ((j.adobe && j.adobe_transform == 0) || (j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B'));
    int cr_r[256], cb_b[256]; long cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
        const long x = i - 128;
        cr_r[i] = (int)((91881L * x + 32768L) >> 16);          // FIX(1.40200)
        cb_b[i] = (int)((116130L * x + 32768L) >> 16);         // FIX(1.77200)
        cr_g[i] = -46802L * x;                                  // FIX(0.71414)
        cb_g[i] = -22554L * x + 32768L;                         // FIX(0.34414) + ONE_HALF
    }
    for (int y = 0; y < H; ++y) {
        unsigned char* d = base + (size_t)y * step;
        const unsigned char* p0 = &full[0][(size_t)y * W];
        if (j.ncomp == 1) { for (int x = 0; x < W; ++x) { d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = p0[x]; } continue; }
        const unsigned char* p1 = &full[1][(size_t)y * W];
        const unsigned char* p2 = &full[2][(size_t)y * W];
        if (rgb) { for (int x = 0; x < W; ++x) { d[3 * x] = p2[x]; d[3 * x + 1] = p1[x]; d[3 * x + 2] = p0[x]; } continue; }
        for (int x = 0; x < W; ++x) {
            const int Y = p0[x], cb = p1[x], cr = p2[x];
            d[3 * x + 2] = range_limit(Y + cr_r[cr]);
            d[3 * x + 1] = range_limit(Y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
            d[3 * x] = range_limit(Y + cb_b[cb]);
        }
    }
    if (out->device >= 0) {
        ISX_HIP(hipSetDevice(out->device));
        ISX_HIP(hipMemcpy2D(out->data, out->step, host.data(), dense, dense, H, hipMemcpyHostToDevice));
    }
    return ISX_OK;
}

}  // extern "C"
