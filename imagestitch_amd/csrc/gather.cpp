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
#include <string>
#include <vector>

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
    // one-time dlopen / dlsym, safe against concurrent first calls (function-local static: initialised once, by one thread);
    // the dlerror text of a failed load is captured at load time (dlerror() is only meaningful right after the failing call)
    struct Loaded { Rccl r; std::string err; };
    static const Loaded L = [] {
        Loaded l;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            l.r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (l.r.so) break;
            const char* e = dlerror();
            l.err = e ? e : "unknown dlopen error";
        }
        if (l.r.so) {
            l.r.GetUniqueId = (int (*)(UniqueId*))dlsym(l.r.so, "ncclGetUniqueId");
            l.r.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(l.r.so, "ncclCommInitRank");
            l.r.CommDestroy = (int (*)(Comm))dlsym(l.r.so, "ncclCommDestroy");
            l.r.AllGather = (int (*)(const void*, void*, size_t, int, Comm, hipStream_t))dlsym(l.r.so, "ncclAllGather");
            l.r.GroupStart = (int (*)())dlsym(l.r.so, "ncclGroupStart");
            l.r.GroupEnd = (int (*)())dlsym(l.r.so, "ncclGroupEnd");
            l.r.GetErrorString = (const char* (*)(int))dlsym(l.r.so, "ncclGetErrorString");
        }
        return l;
    }();
    ISX_CHECK_ARG(L.r.so != nullptr, ISX_ERR_UNSUPPORTED, "isx_gather: librccl.so.1 could not be loaded (%s)", L.err.c_str());
    ISX_CHECK_ARG(L.r.GetUniqueId && L.r.CommInitRank && L.r.CommDestroy && L.r.AllGather && L.r.GroupStart && L.r.GroupEnd, ISX_ERR_UNSUPPORTED,
                  "isx_gather: librccl lacks an expected entry point");
    *out = const_cast<Rccl*>(&L.r);
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
    // direct schedule (isx_gather_p2p_*): every rank's receive buffer mapped here, one copy stream per destination
    void* p2p_local = nullptr;             // this rank's receive buffer (isx_gather_p2p_alloc)
    size_t p2p_bytes = 0;
    std::vector<void*> peer;               // peer[r] = rank r's receive buffer in this process's address space ([rank] = p2p_local)
    std::vector<hipStream_t> pstream;      // one per destination rank: the copies to different peers cross different xGMI links
    std::vector<hipEvent_t> pdone;
};

extern "C" {

int isx_gather_unique_id(unsigned char id[128]) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(id != nullptr, ISX_ERR_INVALID, "isx_gather_unique_id: null id");
    Rccl* r = nullptr;
    ISX_TRY(load_rccl(&r));
    UniqueId u;
    ISX_NCCL(r, r->GetUniqueId(&u));
    std::memcpy(id, u.internal, 128);
    return ISX_OK;
} ISX_EXIT("isx_gather_unique_id")

int isx_gather_create(int world, int rank, const unsigned char id[128], int device, isx_gather** out) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(out != nullptr, ISX_ERR_INVALID, "isx_gather_create: null argument");
    *out = nullptr;
    ISX_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, ISX_ERR_INVALID, "isx_gather_create: rank %d of %d", rank, world);
    Rccl* r = nullptr;
    if (id) ISX_TRY(load_rccl(&r));      // id == NULL: no communicator, only the direct schedule (isx_gather_p2p_*) is available
    ISX_HIP(hipSetDevice(device));
    isx_gather* g = new (std::nothrow) isx_gather();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_NOMEM, "isx_gather_create: out of host memory");
    g->r = r; g->world = world; g->rank = rank; g->device = device;
    if (id) {
        UniqueId u;
        std::memcpy(u.internal, id, 128);
        int rc = r->CommInitRank(&g->comm, world, u, rank);
        if (rc != 0) { delete g; return fail(ISX_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "rccl error"); }
    }
    hipError_t e = hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
    if (e != hipSuccess) { if (g->comm) (void)r->CommDestroy(g->comm); delete g; return fail(ISX_ERR_HIP, "isx_gather_create: %s", hipGetErrorString(e)); }
    *out = g;
    return ISX_OK;
} ISX_EXIT("isx_gather_create")

int isx_gather_destroy(isx_gather* g) ISX_ENTRY {
    if (!g) return ISX_OK;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->comm_stream);
    for (size_t r = 0; r < g->pstream.size(); ++r) {
        (void)hipStreamSynchronize(g->pstream[r]);
        (void)hipStreamDestroy(g->pstream[r]);
        (void)hipEventDestroy(g->pdone[r]);
    }
    for (size_t r = 0; r < g->peer.size(); ++r)
        if ((int)r != g->rank && g->peer[r]) (void)hipIpcCloseMemHandle(g->peer[r]);
    if (g->p2p_local) (void)hipFree(g->p2p_local);
    if (g->comm) (void)g->r->CommDestroy(g->comm);
    (void)hipEventDestroy(g->done);
    (void)hipStreamDestroy(g->comm_stream);
    if (have_prev) (void)hipSetDevice(prev);
    delete g;
    return ISX_OK;
} ISX_EXIT("isx_gather_destroy")

int isx_gather_all(isx_gather* g, const void* send, size_t bytes, void* recv, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && send != nullptr && recv != nullptr && bytes > 0, ISX_ERR_INVALID, "isx_gather_all: bad argument");
    ISX_CHECK_ARG(g->comm != nullptr, ISX_ERR_STATE, "isx_gather_all: this handle was created without a communicator (id == NULL)");
    ISX_HIP(hipSetDevice(g->device));
    ISX_NCCL(g->r, g->r->AllGather(send, recv, bytes, NCCL_UINT8, g->comm, (hipStream_t)hip_stream));
    return ISX_OK;
} ISX_EXIT("isx_gather_all")

int isx_gather_chunk(isx_gather* g, const void* send_base, size_t block_bytes, size_t offset, size_t bytes, void* recv_base, void* ready_event) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && send_base != nullptr && recv_base != nullptr, ISX_ERR_INVALID, "isx_gather_chunk: null argument");
    ISX_CHECK_ARG(g->comm != nullptr, ISX_ERR_STATE, "isx_gather_chunk: this handle was created without a communicator (id == NULL)");
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
} ISX_EXIT("isx_gather_chunk")

int isx_gather_chunk_ptr(const isx_gather* g, void* recv_base, size_t offset, size_t bytes, int rank, void** ptr) ISX_ENTRY {
    ISX_CHECK_ARG(g != nullptr && recv_base != nullptr && ptr != nullptr && rank >= 0 && rank < g->world, ISX_ERR_INVALID, "isx_gather_chunk_ptr: bad argument");
    *ptr = (unsigned char*)recv_base + (size_t)g->world * offset + (size_t)rank * bytes;
    return ISX_OK;
} ISX_EXIT("isx_gather_chunk_ptr")

int isx_gather_wait(isx_gather* g, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_wait: null gather");
    ISX_HIP(hipSetDevice(g->device));
    ISX_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, g->done, 0));   // the stream's next work sees every chunk enqueued so far
    return ISX_OK;
} ISX_EXIT("isx_gather_wait")

int isx_gather_synchronize(isx_gather* g) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_synchronize: null gather");
    ISX_HIP(hipSetDevice(g->device));
    ISX_HIP(hipStreamSynchronize(g->comm_stream));
    return ISX_OK;
} ISX_EXIT("isx_gather_synchronize")

// ---- the direct schedule: every chunk copied straight into every rank's receive buffer ----------------------------------------------
// xGMI is point to point: a rank's block has to cross each of its world - 1 links once whatever the schedule.  An all-gather leaves
// the choice of rings / trees / channels to RCCL; here the rank issues world - 1 plain device-to-device copies on world - 1 streams,
// one per link, into buffers the peers exported through HIP IPC (dmabuf).  Same layout as isx_gather_chunk (rank-major per chunk,
// isx_gather_chunk_ptr), so the two schedules are interchangeable - and comparable: bench.py --gather-backend p2p against torch / isx
// tells "RCCL's schedule" from "the links".  Arrival on the DESTINATION rank is not signalled by the copy itself: consumers
// synchronise across ranks as they would after any one-sided put (bench.py: the barrier that closes the timed region).
int isx_gather_p2p_alloc(isx_gather* g, size_t bytes, void** ptr, unsigned char handle[64]) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && ptr != nullptr && handle != nullptr && bytes > 0, ISX_ERR_INVALID, "isx_gather_p2p_alloc: bad argument");
    ISX_CHECK_ARG(g->p2p_local == nullptr, ISX_ERR_STATE, "isx_gather_p2p_alloc: the receive buffer exists already");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    ISX_HIP(hipSetDevice(g->device));
    ISX_HIP(hipMalloc(&g->p2p_local, bytes));        // its own allocation: an IPC handle names a whole hipMalloc, not a sub-range of a caching allocator
    g->p2p_bytes = bytes;
    hipIpcMemHandle_t h;
    ISX_HIP(hipIpcGetMemHandle(&h, g->p2p_local));
    std::memcpy(handle, &h, 64);
    *ptr = g->p2p_local;
    return ISX_OK;
} ISX_EXIT("isx_gather_p2p_alloc")

// undo a partly completed isx_gather_p2p_open: close the peers' mappings opened so far, destroy the streams / events created so far and
// leave the gather as it was before the call, so that the call can be repeated
static void p2p_rollback(isx_gather* g) {
    for (int r = 0; r < (int)g->peer.size(); ++r)
        if (r != g->rank && g->peer[r]) (void)hipIpcCloseMemHandle(g->peer[r]);
    g->peer.clear();
    for (hipStream_t st : g->pstream) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : g->pdone) (void)hipEventDestroy(ev);
    g->pstream.clear(); g->pdone.clear();
}

int isx_gather_p2p_open(isx_gather* g, const unsigned char* handles /* world x 64 bytes, by rank */) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && handles != nullptr, ISX_ERR_INVALID, "isx_gather_p2p_open: bad argument");
    ISX_CHECK_ARG(g->p2p_local != nullptr, ISX_ERR_STATE, "isx_gather_p2p_open: isx_gather_p2p_alloc first");
    ISX_CHECK_ARG(g->peer.empty(), ISX_ERR_STATE, "isx_gather_p2p_open: already open");
    ISX_HIP(hipSetDevice(g->device));
    g->peer.assign((size_t)g->world, nullptr);
    // a failure part of the way leaves nothing behind (no half-filled peer table that a later chunk would dereference, no "already open")
#define P2P_OPEN_HIP(expr)                                                                                                       \
    do {                                                                                                                         \
        hipError_t e__ = (expr);                                                                                                 \
        if (e__ != hipSuccess) { p2p_rollback(g); return fail(ISX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); } \
    } while (0)
    for (int r = 0; r < g->world; ++r) {
        if (r == g->rank) { g->peer[r] = g->p2p_local; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        P2P_OPEN_HIP(hipIpcOpenMemHandle(&g->peer[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    for (int r = 0; r < g->world; ++r) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        P2P_OPEN_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        g->pstream.push_back(st);
        P2P_OPEN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->pdone.push_back(ev);
    }
#undef P2P_OPEN_HIP
    return ISX_OK;
} ISX_EXIT("isx_gather_p2p_open")

int isx_gather_p2p_chunk(isx_gather* g, const void* send_base, size_t block_bytes, size_t offset, size_t bytes, void* ready_event) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr && send_base != nullptr, ISX_ERR_INVALID, "isx_gather_p2p_chunk: null argument");
    ISX_CHECK_ARG(!g->peer.empty(), ISX_ERR_STATE, "isx_gather_p2p_chunk: isx_gather_p2p_open first");
    ISX_CHECK_ARG(bytes > 0 && offset + bytes <= block_bytes && (size_t)g->world * block_bytes <= g->p2p_bytes, ISX_ERR_INVALID,
                  "isx_gather_p2p_chunk: chunk [%zu, %zu) of a %zu-byte block does not fit the %zu-byte receive buffers of %d ranks", offset,
                  offset + bytes, block_bytes, g->p2p_bytes, g->world);
    ISX_HIP(hipSetDevice(g->device));
    for (int i = 0; i < g->world; ++i) {
        const int r = (g->rank + 1 + i) % g->world;         // every rank starts with its right-hand neighbour: no link is everyone's first
        if (ready_event) ISX_HIP(hipStreamWaitEvent(g->pstream[r], (hipEvent_t)ready_event, 0));
        unsigned char* dst = (unsigned char*)g->peer[r] + (size_t)g->world * offset + (size_t)g->rank * bytes;
        ISX_HIP(hipMemcpyAsync(dst, (const unsigned char*)send_base + offset, bytes, hipMemcpyDeviceToDevice, g->pstream[r]));
        ISX_HIP(hipEventRecord(g->pdone[r], g->pstream[r]));
    }
    return ISX_OK;
} ISX_EXIT("isx_gather_p2p_chunk")

// makes hip_stream wait (without blocking the host) for every copy this rank has enqueued so far (its send block may then be rewritten)
int isx_gather_p2p_wait(isx_gather* g, void* hip_stream) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_p2p_wait: null gather");
    ISX_HIP(hipSetDevice(g->device));
    for (size_t r = 0; r < g->pdone.size(); ++r) ISX_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, g->pdone[r], 0));
    return ISX_OK;
} ISX_EXIT("isx_gather_p2p_wait")

int isx_gather_p2p_synchronize(isx_gather* g) ISX_ENTRY {
    clear_error();
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_p2p_synchronize: null gather");
    ISX_HIP(hipSetDevice(g->device));
    for (size_t r = 0; r < g->pstream.size(); ++r) ISX_HIP(hipStreamSynchronize(g->pstream[r]));
    return ISX_OK;
} ISX_EXIT("isx_gather_p2p_synchronize")

int isx_gather_info(const isx_gather* g, int* world, int* rank) ISX_ENTRY {
    ISX_CHECK_ARG(g != nullptr, ISX_ERR_INVALID, "isx_gather_info: null gather");
    if (world) *world = g->world;
    if (rank) *rank = g->rank;
    return ISX_OK;
} ISX_EXIT("isx_gather_info")

}  // extern "C"
