// isx_internal.hpp — shared host-side plumbing of libimagestitch_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/imagestitch_hip.h"

namespace isx {

// ---- errors: thread-local message + status code, no exceptions across the C boundary ----------
int fail(int code, const char* fmt, ...);
void clear_error();
// The exception barrier of the C boundary (status codes, never a C++ exception - the reference's own errors are cv::Exceptions,
// W:94-96, which a C caller, a ctypes / cgo / JNI binding or a C++ caller built with another runtime cannot catch).  Every extern "C" entry
// is a function-try-block:   int isx_x(...) ISX_ENTRY { ... } ISX_EXIT("isx_x")
// on_exception() runs inside the handler: std::bad_alloc / std::length_error -> ISX_ERR_NOMEM, anything else -> ISX_ERR_INTERNAL, the
// message in isx_last_error(); it never throws itself (a fixed message when even the string cannot be stored).
int on_exception(const char* entry) noexcept;
#define ISX_ENTRY try
#define ISX_EXIT(entry) catch (...) { return ::isx::on_exception(entry); }

#define ISX_CHECK_ARG(cond, code, ...)                   \
    do {                                                 \
        if (!(cond)) return ::isx::fail(code, __VA_ARGS__); \
    } while (0)

#define ISX_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return ::isx::fail(e__ == hipErrorOutOfMemory ? ISX_ERR_NOMEM : ISX_ERR_HIP,       \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define ISX_TRY(expr)                \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != ISX_OK) return rc__; \
    } while (0)

// ---- cv::Mat type helpers --------------------------------------------------------------------
inline int mat_depth(int type) { return type & 7; }
inline int mat_cn(int type) { return (type >> 3) + 1; }
inline int mat_elem_size(int type) {
    static const int dsz[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return dsz[mat_depth(type)] * mat_cn(type);
}
const char* type_name(int type);

// ---- device buffer that only ever grows (no hipMalloc in the steady state) -------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // grows if needed (synchronous hipFree/hipMalloc), keeps contents undefined
    void release();
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Stages a host isx_mat through HBM (device == -1) or passes a device mat through.
// After use_in() `d` is a device mat with the same geometry; finish_out() copies results back.
struct MatStage {
    DevBuf buf;
    isx_mat d{};        // device view
    const isx_mat* host = nullptr;
    int use_in(const isx_mat* m, hipStream_t s, const char* what);   // H2D if needed
    int use_out(isx_mat* m, hipStream_t s, const char* what);        // allocate staging if needed
    int finish_out(hipStream_t s);                                   // D2H if needed (async)
    int finish_out_cols(hipStream_t s, int col0, int col1);          // D2H of columns [col0, col1) only: the rest of the caller's mat keeps its contents
};

int check_mat(const isx_mat* m, const char* what);

// ---- per-kernel profiler (HIP events on the launch stream) -------------------------------------
// A bracketed launch goes through hipExtLaunchKernelGGL with a start and a stop event: the two events then carry the
// kernel's own begin / end timestamps (what a kernel trace reports), not the times two separate hipEventRecord barrier
// packets completed - those include the dispatch latency either side of the kernel and cost the stream ~13 us of gaps.
struct ProfScope {
    ProfScope(const char* name, hipStream_t s, double alg_bytes);
    int slot = -1;                      // >= 0: this launch is bracketed with start / stop
    hipEvent_t start = nullptr, stop = nullptr;
};
bool profiling_enabled();

// Launch wrapper: brackets the launch when profiling asks for it, checks the launch error.
#define ISX_LAUNCH(name, alg_bytes, stream, kernel, grid, block, shmem, ...)                   \
    do {                                                                                       \
        ::isx::ProfScope ps__(name, stream, (double)(alg_bytes));                              \
        if (ps__.slot >= 0) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ps__.start, ps__.stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);              \
        hipError_t le__ = hipGetLastError();                                                   \
        if (le__ != hipSuccess)                                                                \
            return ::isx::fail(ISX_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(le__)); \
    } while (0)

// prep.hip: dilate(mask, MORPH_RECT kw x kh) [& other] between device buffers (one kernel, elements up to 33 a side)
int dilate_and_device(const unsigned char* mask, size_t mstep, const unsigned char* other, size_t ostep, int rows, int cols, int kw, int kh,
                      unsigned char* dst, size_t dstep, hipStream_t st);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// blocks of a 1-D launch in the XCD-aware order of isx_device.hpp's xcd_block: gy rows in groups of grp, the groups dealt to 8 XCDs
inline unsigned xcd_magic(int grp, int gx) { return 0xFFFFFFFFu / (unsigned)(grp * gx) + 1u; }
inline unsigned xcd_grid_blocks(int grp, int gx, int gy) { return (unsigned)(8 * cdiv(cdiv(gy, grp), 8) * grp) * (unsigned)gx; }

inline unsigned xcd_band_blocks(int grp, int gx, int gy) { return (unsigned)(8 * cdiv(cdiv(gy, 8), grp) * grp) * (unsigned)gx; }

}  // namespace isx
