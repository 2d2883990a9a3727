// hipHostRegister / hipHostUnregister per call: the same buffer again and again against a fresh allocation each time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t NA = 24883200;
    char* dA; CK(hipMalloc(&dA, NA));
    char* A = (char*)malloc(NA); memset(A, 1, NA);
    CK(hipMemcpy(dA, A, NA, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 4; ++rep) {
        double t0 = now(); CK(hipHostRegister(A, NA, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(dA, A, NA, hipMemcpyHostToDevice, 0)); CK(hipStreamSynchronize(0)); double t2 = now();
        CK(hipHostUnregister(A)); double t3 = now();
        printf("same buffer : register %.3f ms  copy %.3f ms  unregister %.3f ms\n", t1 - t0, t2 - t1, t3 - t2);
    }
    for (int rep = 0; rep < 4; ++rep) {
        char* F = (char*)malloc(NA + 4096 * rep); memset(F, 3, NA);
        double t0 = now(); CK(hipHostRegister(F, NA, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(dA, F, NA, hipMemcpyHostToDevice, 0)); CK(hipStreamSynchronize(0)); double t2 = now();
        CK(hipHostUnregister(F)); double t3 = now();
        printf("fresh buffer: register %.3f ms  copy %.3f ms  unregister %.3f ms\n", t1 - t0, t2 - t1, t3 - t2);
        double t4 = now(); CK(hipMemcpy(dA, F, NA, hipMemcpyHostToDevice)); double t5 = now();
        printf("              pageable copy of the same buffer %.3f ms\n", t5 - t4);
        free(F);
    }
    return 0;
}
