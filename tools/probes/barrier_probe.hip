// Grid-barrier cost on MI355X: hipcc --offload-arch=gfx950 -O3 tools/probes/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
// A persistent kernel of G workgroups runs K device-wide barriers (arrive: one agent-scope atomic add; wait: spin on an agent-scope load);
// per-barrier cost = (t(K) - t(0)) / K.  Decides whether levels >= 2 of the pyramid can live in one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned* gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ void k_barriers(unsigned* counter, unsigned* gen, int K, float* data, int n) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        // a little work between barriers: each block touches a slice (written by another block in the previous phase)
        const int i = (blockIdx.x * blockDim.x + threadIdx.x + k * 7919) % n;
        acc += data[i];
        data[(i + 13) % n] = acc;
        grid_barrier(counter, gen, gridDim.x);
    }
    if (acc == 12345.f) data[0] = acc;
}

int main() {
    unsigned* ctr; float* data; const int n = 1 << 20;
    hipMalloc(&ctr, 8); hipMemset(ctr, 0, 8); hipMalloc(&data, n * 4); hipMemset(data, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {32, 64, 128, 256, 512}) {
        for (int threads : {256}) {
            float t[2];
            int Ks[2] = {0, 200};
            for (int j = 0; j < 2; ++j) {
                hipLaunchKernelGGL(k_barriers, dim3(G), dim3(threads), 0, 0, ctr, ctr + 1, Ks[j], data, n);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_barriers, dim3(G), dim3(threads), 0, 0, ctr, ctr + 1, Ks[j], data, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&t[j], e0, e1); t[j] /= 5;
            }
            printf("G=%4d threads=%d: empty kernel %.2f us, per barrier %.3f us\n", G, threads, t[0] * 1e3, (t[1] - t[0]) * 1e3 / Ks[1]);
        }
    }
    return 0;
}
