// What does it cost the MAIN stream to let a side stream start behind one of its kernels?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/waitvalue_probe.hip -o /tmp/wv && timeout 60 /tmp/wv
// (a) nothing between the kernels; (b) hipEventRecord between them + hipStreamWaitEvent on the side stream (what blend() does for the
// ROI verification scans); (c) the first kernel's last block writes a flag in signal memory, the side stream waits with
// hipStreamWaitValue32: no packet on the main stream at all.  Reports the main stream's time per iteration of two dependent kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void busy(unsigned* sink, int iters, unsigned* done, unsigned* flag, unsigned value) {
    unsigned a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1664525u + 1013904223u;
    if (a == 12345u) *sink = a;
    if (flag) {   // last block out sets the flag
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(done, 1u) == gridDim.x - 1) { *done = 0; __threadfence(); __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}
__global__ void side_work(unsigned* sink, int iters) {
    unsigned a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 22695477u + 1u;
    if (a == 12345u) *sink = a;
}
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t main_s, side_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side_s, hipStreamNonBlocking));
    unsigned *sink, *done, *flag;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&done, 4)); CK(hipMemset(done, 0, 4));
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory)); CK(hipMemset(flag, 0, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 200, BLOCKS = 512, IT = 3000;
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && !can) break;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                const unsigned v = (unsigned)(rep * N + i + 1);
                hipLaunchKernelGGL(busy, dim3(BLOCKS), dim3(256), 0, main_s, sink, IT, done, mode == 2 ? flag : nullptr, v);
                if (mode == 1) { CK(hipEventRecord(ev, main_s)); CK(hipStreamWaitEvent(side_s, ev, 0)); }
                if (mode == 2) CK(hipStreamWaitValue32(side_s, flag, v, hipStreamWaitValueGte, 0xffffffffu));
                if (mode) hipLaunchKernelGGL(side_work, dim3(64), dim3(256), 0, side_s, sink, 2000);
                hipLaunchKernelGGL(busy, dim3(BLOCKS), dim3(256), 0, main_s, sink, IT, done, nullptr, 0u);
            }
            CK(hipStreamSynchronize(main_s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            if (rep) printf("mode %d (%s): %.2f us per pair of kernels on the main stream\n", mode,
                            mode == 0 ? "nothing between" : (mode == 1 ? "event record + side stream waits on the event" : "flag in signal memory + hipStreamWaitValue32"),
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
    }
    return 0;
}
