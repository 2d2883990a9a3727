// valu_probe.hip — three hardware questions behind the round-3 rolling collapse kernel (run on the GPU box):
//   1. issue cost of v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 against v_add_f32 / v_fma_f32 and a DPP-modified v_add_f32
//   2. do byte-misaligned global_load_dwordx2 / dwordx3 / ushort return the right bytes, and at what rate against the aligned
//      12-byte window + v_alignbyte scheme the level-0 kernels use
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probes/valu_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k_valu(float* out, int iters, float seed) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const f32x2 c2 = {seed, seed * 0.5f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(seed));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
        } else if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
float time_valu(float* d, int waves_per_simd) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(64), 0, 0, d, 100, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int per_iter = (MODE >= 1 && MODE <= 3) ? 8 : 16;
    // ns per wave-instruction per SIMD
    return ms * 1e6f / ((float)iters * per_iter * waves_per_simd);
}

// ---- loads -------------------------------------------------------------------------------------------------------
struct U3 { unsigned x, y, z; };
struct U2 { unsigned x, y; };
// every lane reads the 6 bytes of "its" two CV_8UC3 pixels at byte offset mis + 6 * (global pixel-pair index)
template <int MODE>
__global__ __launch_bounds__(256) void k_load(const unsigned char* base, unsigned mis, size_t npairs, unsigned long long* sum) {
    unsigned long long acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npairs; i += stride) {
        const size_t off = (size_t)mis + 6 * i;
        unsigned lo, hi;
        if (MODE == 0) {          // aligned 12-byte window + alignbyte (the current scheme)
            const U3 v = *(const U3*)(base + (off & ~(size_t)3));
            lo = __builtin_amdgcn_alignbyte(v.y, v.x, (unsigned)off & 3u); hi = __builtin_amdgcn_alignbyte(v.z, v.y, (unsigned)off & 3u);
        } else if (MODE == 1) {   // misaligned 8-byte load
            U2 v;
            const unsigned char* p = base + off;
            asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            lo = v.x; hi = v.y;
        } else {                  // misaligned 2-byte + 4-byte... : ushort at odd offsets
            unsigned a, b;
            const unsigned char* p = base + off;
            asm volatile("global_load_dword %0, %2, off\n\tglobal_load_ushort %1, %2, off offset:4\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
            lo = a; hi = b;
        }
        acc += (unsigned long long)lo + ((unsigned long long)(hi & 0xffffu) << 32);
    }
    atomicAdd(sum, acc);
}

template <int MODE>
void run_load(const unsigned char* d, const std::vector<unsigned char>& h, size_t bytes, unsigned long long* dsum) {
    for (unsigned mis = 0; mis < 4; ++mis) {
        const size_t npairs = (bytes - 64) / 6;
        unsigned long long ref = 0;
        for (size_t i = 0; i < npairs; ++i) {
            const unsigned char* p = h.data() + mis + 6 * i;
            unsigned lo = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24), hi = p[4] | (p[5] << 8);
            ref += (unsigned long long)lo + ((unsigned long long)hi << 32);
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f; unsigned long long got = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(dsum, 0, 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_load<MODE>, dim3(256 * 16), dim3(256), 0, 0, d, mis, npairs, dsum);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost));
        }
        printf("load mode %d (%s) misalign %u: %s  %.3f ms  %.1f GB/s of useful bytes\n", MODE,
               MODE == 0 ? "aligned dwordx3 window + alignbyte" : MODE == 1 ? "misaligned dwordx2" : "misaligned dword + ushort", mis,
               got == ref ? "values OK" : "VALUES WRONG", best, 6.0 * npairs / best / 1e6);
    }
}

int main() {
    float* d; CK(hipMalloc(&d, 1024));
    const char* names[] = {"v_add_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_fma_f32", "v_add_f32_dpp wave_shr:1", "v_mov_b32_dpp wave_shr:1", "v_cvt_f32_ubyte1", "v_rcp_f32"};
    for (int w = 1; w <= 4; w *= 2) {
        float t[9] = {time_valu<0>(d, w), time_valu<1>(d, w), time_valu<2>(d, w), time_valu<3>(d, w), time_valu<4>(d, w), time_valu<5>(d, w), time_valu<6>(d, w), time_valu<7>(d, w), time_valu<8>(d, w)};
        for (int i = 0; i < 9; ++i) printf("waves/SIMD %d  %-26s %.3f ns per wave-instruction per SIMD (ratio to v_add_f32 %.2f)\n", w, names[i], t[i], t[i] / t[0]);
    }
    const size_t bytes = (size_t)256 << 20;
    std::vector<unsigned char> h(bytes);
    unsigned s = 12345u;
    for (size_t i = 0; i < bytes; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned char)(s >> 24); }
    unsigned char* db; CK(hipMalloc(&db, bytes)); CK(hipMemcpy(db, h.data(), bytes, hipMemcpyHostToDevice));
    unsigned long long* dsum; CK(hipMalloc(&dsum, 8));
    run_load<0>(db, h, bytes, dsum);
    run_load<1>(db, h, bytes, dsum);
    run_load<2>(db, h, bytes, dsum);
    return 0;
}
