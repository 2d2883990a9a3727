// SURVEY §8(f) N4 — the on-disk format either side of the hot path: imread / imwrite of .bmp files
// (W:166 `imread(img_names[i])`, W:155-156,315 `imwrite("....bmp", ...)`; the reference's committed artefacts are BMPs).
// Host code only: uncompressed Windows bitmaps, 24-bit BGR and 8-bit paletted, bottom-up or top-down.
//   read : what cv::imread(path) (IMREAD_COLOR) returns — always 3 channels BGR, palettes expanded
//   write: what cv::imwrite does for CV_8UC3 (24-bit) and CV_8UC1 (8-bit, grey palette), rows padded to 4 bytes
// JPEG (pano.jpg) is out of scope.
#include <hip/hip_runtime.h>

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
    bi.height = hh < 0 ? -hh : hh;
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

}  // namespace

extern "C" {

int isx_bmp_size(const char* path, int* rows, int* cols) {
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
}

int isx_bmp_read(const char* path, isx_mat* out) {
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
}

int isx_bmp_write(const char* path, const isx_mat* img) {
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
}

}  // extern "C"
