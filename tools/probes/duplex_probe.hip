// What a duplex host path can count on: pageable copies each way, the cost of registering a caller's buffer,
// both directions at once (registered + async on two streams; pageable from two host threads).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t NA = 24883200, NB = 29592000;      // a 4K CV_8UC3 tile up; its warped image + mask down
    char* A = (char*)malloc(NA); char* B = (char*)malloc(NB);
    memset(A, 1, NA); memset(B, 2, NB);
    char *dA, *dB;
    CK(hipMalloc(&dA, NA)); CK(hipMalloc(&dB, NB));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpy(dA, A, NA, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(B, dB, NB, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable: H2D %.3f ms (%.1f GB/s)  D2H %.3f ms (%.1f GB/s)  serial %.3f ms\n", t1 - t0, NA / (t1 - t0) / 1e6, t2 - t1, NB / (t2 - t1) / 1e6, t2 - t0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        std::thread th([&] { (void)hipMemcpy(B, dB, NB, hipMemcpyDeviceToHost); });
        CK(hipMemcpy(dA, A, NA, hipMemcpyHostToDevice));
        th.join();
        printf("pageable, two host threads at once: %.3f ms\n", now() - t0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipHostRegister(A, NA, hipHostRegisterDefault)); double t1 = now(); CK(hipHostRegister(B, NB, hipHostRegisterDefault)); double t2 = now();
        const int C = 8;
        for (int c = 0; c < C; ++c) {
            CK(hipMemcpyAsync(dA + NA / C * c, A + NA / C * c, NA / C, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(B + NB / C * c, dB + NB / C * c, NB / C, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        double t3 = now(); CK(hipHostUnregister(A)); CK(hipHostUnregister(B)); double t4 = now();
        printf("registered: register %.3f + %.3f ms, 8 + 8 chunks both ways at once %.3f ms, unregister %.3f ms, total %.3f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0);
    }
    // chunked pageable copies on one thread, alternating directions (what a single-threaded banded path would issue)
    for (int rep = 0; rep < 2; ++rep) {
        const int C = 8; double t0 = now();
        for (int c = 0; c < C; ++c) {
            CK(hipMemcpyAsync(dA + NA / C * c, A + NA / C * c, NA / C, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(B + NB / C * c, dB + NB / C * c, NB / C, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        printf("pageable, 8 + 8 async chunks from one thread: %.3f ms\n", now() - t0);
    }
    return 0;
}
