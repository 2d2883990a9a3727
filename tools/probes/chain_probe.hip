// Dependent phases in ONE launch against one launch per phase on MI355X:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
// The small pyramid levels of a 4K pair are six dependent launches of 20 - 490 workgroups (pyrDown 2->3->4->5, collapse 5->4->3->2): 39 us of a
// 211 us step at a few percent of the GPU.  Here phase p's blocks (a 1-D grid, phases in block-index order) wait until every block of phase
// p - 1 has published a per-block flag (a plain release store of the launch's epoch - no read-modify-write on a shared counter, which
// serialises at the memory side), then read what that phase wrote.  Blocks are dispatched in index order, so a waiting block's producers
// are always resident or done: no deadlock; the spin is bounded anyway.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NP = 6;
struct Phases { int first[NP + 1]; };

__device__ __forceinline__ void phase_work(const float* in, float* out, int blk, int nblk_prev, int elems) {
    // every block reads a slice the previous phase wrote (4 KB per block) and writes its own 4 KB
    float acc = 0.f;
    const int src = (blk * 7 + 3) % (nblk_prev > 0 ? nblk_prev : 1);
    for (int i = threadIdx.x; i < elems; i += blockDim.x) acc += in[(size_t)src * elems + i];
    for (int i = threadIdx.x; i < elems; i += blockDim.x) out[(size_t)blk * elems + i] = acc + 1.f;
}

__global__ __launch_bounds__(256) void k_chain(Phases ph, float* buf0, float* buf1, unsigned* flags, unsigned epoch, int* timeouts, int elems) {
    const int b = blockIdx.x;
    int p = 0;
    for (int i = 1; i < NP; ++i) p += b >= ph.first[i] ? 1 : 0;
    if (p > 0) {    // wait for every block of phase p - 1
        const int lo = ph.first[p - 1], hi = ph.first[p];
        int spins = 0;
        for (;;) {
            bool ok = true;
            for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) ok = ok && __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
            if (__syncthreads_and(ok)) break;
            if (++spins > (1 << 20)) { if (threadIdx.x == 0) atomicAdd(timeouts, 1); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float* in = (p & 1) ? buf0 : buf1;
    float* out = (p & 1) ? buf1 : buf0;
    phase_work(in, out, b - ph.first[p], p > 0 ? ph.first[p] - ph.first[p - 1] : 0, elems);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_phase(float* in, float* out, int nblk_prev, int elems) { phase_work(in, out, blockIdx.x, nblk_prev, elems); }

int main() {
    const int nb[NP] = {288, 72, 20, 36, 140, 490};
    Phases ph; ph.first[0] = 0;
    for (int i = 0; i < NP; ++i) ph.first[i + 1] = ph.first[i] + nb[i];
    const int total = ph.first[NP], elems = 1024;
    float *b0, *b1; unsigned* flags; int* to;
    hipMalloc(&b0, (size_t)512 * elems * 4); hipMalloc(&b1, (size_t)512 * elems * 4); hipMemset(b0, 0, (size_t)512 * elems * 4); hipMemset(b1, 0, (size_t)512 * elems * 4);
    hipMalloc(&flags, total * 4); hipMemset(flags, 0, total * 4); hipMalloc(&to, 4); hipMemset(to, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned epoch = 0;
    const int R = 200;
    float t;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_chain, dim3(total), dim3(256), 0, 0, ph, b0, b1, flags, ++epoch, to, elems);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1);
        int h = 0; hipMemcpy(&h, to, 4, hipMemcpyDeviceToHost);
        printf("one launch, %d phases by flags : %.2f us per chain (timeouts %d)\n", NP, t * 1e3 / R, h);
        hipEventRecord(e0);
        for (int r = 0; r < R; ++r)
            for (int p = 0; p < NP; ++p) hipLaunchKernelGGL(k_phase, dim3(nb[p]), dim3(256), 0, 0, (p & 1) ? b0 : b1, (p & 1) ? b1 : b0, p > 0 ? nb[p - 1] : 0, elems);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1);
        printf("one launch per phase           : %.2f us per chain\n", t * 1e3 / R);
    }
    // the chain with every block of the launch on ONE XCD (8 x the blocks, those with index % 8 != 0 leave at once): the flags and the data then
    // never cross an L2
    return 0;
}
