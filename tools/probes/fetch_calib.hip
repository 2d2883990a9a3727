// What do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access shapes the pyramid kernels use?  (MI355X_MICROARCH.md "HBM": FETCH_SIZE is
// half the bytes of a wide streaming read on gfx950, "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count
// in your own access pattern".)  Every kernel below touches each byte of a 1 GiB buffer exactly once in one shape; a 512 MiB sweep of
// another buffer runs before each so that nothing is left in the 256 MiB Infinity Cache.  Run by tools/fetch_calib.sh under
// `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); counter / known bytes is the calibration factor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void flush_sweep(const uint4* p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
// reads ------------------------------------------------------------------------------------------------------------------
__global__ void rd16_stream(const uint4* p, size_t n, unsigned* sink) {       // 16 B per lane, contiguous across the wave (level records)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 v = p[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = v.x;
}
__global__ void rd12_window_stride8(const unsigned char* p, size_t nbytes, unsigned* sink) {   // 12-byte windows 8 bytes apart: the level-0 windows of the
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                                // last collapse step (2 pixels = 6 bytes of a CV_8UC3 row per
    const size_t off = i * 8;                                                                      // lane, fetched as an aligned 12-byte window: neighbours overlap)
    if (off + 12 > nbytes) return;
    const u32x3 v = *(const u32x3*)(p + off);
    if ((v.x ^ v.y ^ v.z) == 0x12345u) *sink = v.x;
}
__global__ void rd12_stream(const unsigned char* p, size_t nbytes, unsigned* sink) {   // 12 B per lane, back to back
    const size_t off = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 12;
    if (off + 12 > nbytes) return;
    const u32x3 v = *(const u32x3*)(p + off);
    if ((v.x ^ v.y ^ v.z) == 0x12345u) *sink = v.x;
}
__global__ void rd4_stream(const unsigned* p, size_t n, unsigned* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned v = p[i];
    if (v == 0x12345u) *sink = v;
}
__global__ void rd16_lds_dma(const uint4* p, size_t n, unsigned* sink) {      // global_load_lds_dwordx4: the coarse tiles of the collapse steps
    __shared__ uint4 buf[256];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(p + i),
                                     (void __attribute__((address_space(3)))*)(buf + (threadIdx.x & ~63u)), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint4 v = buf[threadIdx.x];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = v.x;
}
// writes -----------------------------------------------------------------------------------------------------------------
__global__ void wr16_stream(uint4* p, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(1, 2, 3, (unsigned)i); }
__global__ void wr12_stream(unsigned char* p, size_t nbytes) {               // three dword stores per lane: a pair of CV_16SC3 result pixels
    const size_t off = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 12;
    if (off + 12 > nbytes) return;
    unsigned* q = (unsigned*)(p + off);
    q[0] = 1; q[1] = 2; q[2] = (unsigned)off;
}
__global__ void wr2_stream(unsigned short* p, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned short)i; }   // mask pairs
__global__ void wr4_stream(unsigned* p, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main() {
    const size_t N = (size_t)1 << 30, F = (size_t)512 << 20;
    unsigned char *buf, *fl; unsigned* sink;
    CK(hipMalloc(&buf, N)); CK(hipMalloc(&fl, F)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, N)); CK(hipMemset(fl, 2, F));
    auto flush = [&]() { hipLaunchKernelGGL(flush_sweep, dim3(4096), dim3(256), 0, 0, (const uint4*)fl, F / 16, sink); };
    const unsigned B = 256;
    flush(); hipLaunchKernelGGL(rd16_stream, dim3((unsigned)(N / 16 / B)), dim3(B), 0, 0, (const uint4*)buf, N / 16, sink);
    flush(); hipLaunchKernelGGL(rd12_window_stride8, dim3((unsigned)(N / 8 / B)), dim3(B), 0, 0, buf, N, sink);
    flush(); hipLaunchKernelGGL(rd12_stream, dim3((unsigned)(N / 12 / B + 1)), dim3(B), 0, 0, buf, N, sink);
    flush(); hipLaunchKernelGGL(rd4_stream, dim3((unsigned)(N / 4 / B)), dim3(B), 0, 0, (const unsigned*)buf, N / 4, sink);
    flush(); hipLaunchKernelGGL(rd16_lds_dma, dim3((unsigned)(N / 16 / B)), dim3(B), 0, 0, (const uint4*)buf, N / 16, sink);
    flush(); hipLaunchKernelGGL(wr16_stream, dim3((unsigned)(N / 16 / B)), dim3(B), 0, 0, (uint4*)buf, N / 16);
    flush(); hipLaunchKernelGGL(wr12_stream, dim3((unsigned)(N / 12 / B + 1)), dim3(B), 0, 0, buf, N);
    flush(); hipLaunchKernelGGL(wr4_stream, dim3((unsigned)(N / 4 / B)), dim3(B), 0, 0, (unsigned*)buf, N / 4);
    flush(); hipLaunchKernelGGL(wr2_stream, dim3((unsigned)(N / 2 / B)), dim3(B), 0, 0, (unsigned short*)buf, N / 2);
    CK(hipDeviceSynchronize());
    printf("known bytes per measured kernel: %zu\n", N);
    return 0;
}
