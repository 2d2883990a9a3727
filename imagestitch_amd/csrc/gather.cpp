// gather.cpp — assembling a batch of blended mosaics across the GPUs of one node (BASELINE config 4).
// Nothing in the reference corresponds to this (no NCCL / MPI anywhere); the contract is north_star's: the independent
// pairs are partitioned across ranks, no data-path collective while blending, and the finished mosaics are assembled on every rank
// by all-gather over xGMI.  This file gives a C / C++ pipeline built on imagestitch_hip.h the collective without torch: it owns an
// RCCL communicator (one process per GPU) and offers
//   isx_gather_all    ONE ncclAllGather of every rank's packed block (the single-collective form), and
//   isx_gather_chunk  the same block gathered chunk by chunk (one chunk = one pair's mosaic): each chunk's all-gather is enqueued on
//                     the handle's communication stream behind an event the caller recorded when that pair's blend was enqueued, so
//                     with P pairs per rank (P - 1) / P of the transfer runs under the blends that follow (xGMI is point to point:
//                     a rank's block crosses each of its links once whatever the schedule, so hiding it is the only lever).
// RCCL is loaded with dlopen at first use: a process that never gathers never needs it, and inside a torch process the copy torch
// already loaded (same SONAME) is the one that is used.
#include "isx_internal.hpp"

#include <dlfcn.h>

#include <new>

using namespace isx;

namespace {

// the few RCCL entry points used, with rccl.h's signatures (ncclResult_t / ncclDataType_t are ints; ncclUniqueId is 128 opaque bytes)
struct UniqueId { char internal[128]; };
typedef void* Comm;
struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_UINT8 = 1;   // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

int load_rccl(Rccl** out) {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (r.so) {
            r.GetUniqueId = (int (*)(UniqueId*))dlsym(r.so, "ncclGetUniqueId");
            r.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(r.so, "ncclCommInitRank");
            r.CommDestroy = (int (*)(Comm))dlsym(r.so, "ncclCommDestroy");
            r.AllGather = (int (*)(const void*, void*, size_t, int, Comm, hipStream_t))dlsym(r.so, "ncclAllGather");
            r.GroupStart = (int (*)())dlsym(r.so, "ncclGroupStart");
            r.GroupEnd = (int (*)())dlsym(r.so, "ncclGroupEnd");
            r.GetErrorString = (const char* (*)(int))dlsym(r.so, "ncclGetErrorString");
        }
    }
    ISX_CHECK_ARG(r.so != nullptr, ISX_ERR_UNSUPPORTED, "isx_gather: librccl.so.1 could not be loaded (%s)", dlerror());
    ISX_CHECK_ARG(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd, ISX_ERR_UNSUPPORTED,
                  "isx_gather: librccl lacks an expected entry point");
    *out = &r;
    return ISX_OK;
}

#define ISX_NCCL(R, expr)                                                                                                   \
    do {                                                                                                                    \
        int rc__ = (expr);                                                                                                  \
        if (rc__ != 0) return ::isx::fail(ISX_ERR_HIP, "%s failed: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(rc__) : "rccl error"); \
    } while (0)

}  // namespace

struct isx_gather {
    Rccl* r = nullptr;
    Comm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t comm_stream = nullptr;     // chunks run here, behind the caller's events
    hipEvent_t done = nullptr;             // recorded behind the last chunk / gather enqueued on comm_stream
};

extern "C" {

int isx_gather_unique_id(unsigned char id[128]) {
    clear_error();
    ISX_CHECK_ARG(id != nullptr, ISX_ERR_INVALID, "isx_gather_unique_id: null id");
    Rccl* r = nullptr;
    ISX_TRY(load_rccl(&r));
    UniqueId u;
    ISX_NCCL(r, r->GetUniqueId(&u));
    std::memcpy(id, u.internal, 128);
    return ISX_OK;
}

int isx_gather_create(int world, int rank, const unsigned char id[128], int device, isx_gather** out) {
    clear_error();
    ISX_CHECK_ARG(out != nullptr && id != nullptr, ISX_ERR_INVALID, "isx_gather_create: null argument");
    *out = nullptr;
    ISX_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, ISX_ERR_INVALID, "isx_gather_create: rank %d of %d", rank, world);
    Rccl* r = nullptr;
    ISX_TRY(load_rccl(&r));
    ISX_HIP(hipSetDevice(device));
    isx_gather* g = new (std::nothrow) isx_gather();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_NOMEM, "isx_gather_create: out of host memory");
    g->r = r; g->world = world; g->rank = rank; g->device = device;
    UniqueId u;
    std::memcpy(u.internal, id, 128);
    int rc = r->CommInitRank(&g->comm, world, u, rank);
    if (rc != 0) { delete g; return fail(ISX_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "rccl error"); }
    hipError_t e = hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
    if (e != hipSuccess) { (void)r->CommDestroy(g->comm); delete g; return fail(ISX_ERR_HIP, "isx_gather_create: %s", hipGetErrorString(e)); }
    *out = g;
    return ISX_OK;
}

int isx_gather_destroy(isx_gather* g) {
    if (!g) return ISX_OK;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->comm_stream);
    if (g->comm) (void)g->r->CommDestroy(g->comm);
    (void)hipEventDestroy(g->done);
    (void)hipStreamDestroy(g->comm_stream);
    delete g;
    return ISX_OK;
}

int isx_gather_all(isx_gather* g, const void* send, size_t bytes, void* recv, void* hip_stream) {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && send != nullptr && recv != nullptr && bytes > 0, ISX_ERR_INVALID, "isx_gather_all: bad argument");
    ISX_HIP(hipSetDevice(g->device));
    ISX_NCCL(g->r, g->r->AllGather(send, recv, bytes, NCCL_UINT8, g->comm, (hipStream_t)hip_stream));
    return ISX_OK;
}

int isx_gather_chunk(isx_gather* g, const void* send_base, size_t block_bytes, size_t offset, size_t bytes, void* recv_base, void* ready_event) {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && send_base != nullptr && recv_base != nullptr, ISX_ERR_INVALID, "isx_gather_chunk: null argument");
    ISX_CHECK_ARG(bytes > 0 && offset + bytes <= block_bytes, ISX_ERR_INVALID, "isx_gather_chunk: chunk [%zu, %zu) outside the %zu-byte block", offset,
                  offset + bytes, block_bytes);
    ISX_HIP(hipSetDevice(g->device));
    if (ready_event) ISX_HIP(hipStreamWaitEvent(g->comm_stream, (hipEvent_t)ready_event, 0));
    // Layout on every rank: recv_base holds `world` regions of chunk_bytes each PER CHUNK, chunk c of rank r at
    //   recv_base + world * offset + r * bytes
    // i.e. the chunks of all ranks are gathered contiguously per chunk (an all-gather writes rank-major); isx_gather_chunk_ptr gives
    // the address of (rank, chunk) so that callers need not know.
    unsigned char* dst = (unsigned char*)recv_base + (size_t)g->world * offset;
    ISX_NCCL(g->r, g->r->AllGather((const unsigned char*)send_base + offset, dst, bytes, NCCL_UINT8, g->comm, g->comm_stream));
    ISX_HIP(hipEventRecord(g->done, g->comm_stream));
    return ISX_OK;
}

int isx_gather_chunk_ptr(const isx_gather* g, void* recv_base, size_t offset, size_t bytes, int rank, void** ptr) {
    ISX_CHECK_ARG(g != nullptr && recv_base != nullptr && ptr != nullptr && rank >= 0 && rank < g->world, ISX_ERR_INVALID, "isx_gather_chunk_ptr: bad argument");
    *ptr = (unsigned char*)recv_base + (size_t)g->world * offset + (size_t)rank * bytes;
    return ISX_OK;
}

int isx_gather_wait(isx_gather* g, void* hip_stream) {
    clear_error();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_wait: null gather");
    ISX_HIP(hipSetDevice(g->device));
    ISX_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, g->done, 0));   // the stream's next work sees every chunk enqueued so far
    return ISX_OK;
}

int isx_gather_synchronize(isx_gather* g) {
    clear_error();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_synchronize: null gather");
    ISX_HIP(hipSetDevice(g->device));
    ISX_HIP(hipStreamSynchronize(g->comm_stream));
    return ISX_OK;
}

int isx_gather_info(const isx_gather* g, int* world, int* rank) {
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_info: null gather");
    if (world) *world = g->world;
    if (rank) *rank = g->rank;
    return ISX_OK;
}

}  // extern "C"
