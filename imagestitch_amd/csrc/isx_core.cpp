// isx_core.cpp — error strings, buffers, host<->device mat staging, per-kernel profiler.
#include "isx_internal.hpp"

#include <map>
#include <mutex>
#include <new>
#include <stdexcept>

namespace isx {

static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
void clear_error() { g_err.clear(); }

int on_exception(const char* entry) noexcept {
    int code = ISX_ERR_INTERNAL;
    const char* what = "an exception that is not a std::exception";
    char buf[512];
    try { throw; }
    catch (const std::bad_alloc&) { code = ISX_ERR_NOMEM; what = "out of host memory (std::bad_alloc)"; }
    catch (const std::length_error& e) { code = ISX_ERR_NOMEM; snprintf(buf, sizeof(buf), "a container was asked for more than it can hold (std::length_error: %s)", e.what()); what = buf; }
    catch (const std::exception& e) { snprintf(buf, sizeof(buf), "%s", e.what()); what = buf; }
    catch (...) { }
    try { return fail(code, "%s: stopped a C++ exception at the C boundary: %s", entry, what); }
    catch (...) {        // the message itself could not be stored: keep whatever isx_last_error() held, the status code still says what happened
        return code;
    }
}

const char* type_name(int type) {
    switch (type) {
        case ISX_8UC1: return "CV_8UC1";
        case ISX_8UC3: return "CV_8UC3";
        case ISX_16SC3: return "CV_16SC3";
        case ISX_32SC1: return "CV_32SC1";
        case ISX_32FC1: return "CV_32FC1";
        case ISX_32FC3: return "CV_32FC3";
        default: return "unsupported type";
    }
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return ISX_OK;
    if (p) { ISX_HIP(hipFree(p)); p = nullptr; cap = 0; }
    // round up so that small geometry changes do not reallocate
    size_t want = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    ISX_HIP(hipMalloc(&p, want));
    cap = want;
    return ISX_OK;
}
void DevBuf::release() {
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
}

int check_mat(const isx_mat* m, const char* what) {
    ISX_CHECK_ARG(m != nullptr, ISX_ERR_INVALID, "%s: null isx_mat", what);
    ISX_CHECK_ARG(m->data != nullptr, ISX_ERR_INVALID, "%s: null data pointer", what);
    ISX_CHECK_ARG(m->rows > 0 && m->cols > 0, ISX_ERR_INVALID, "%s: empty mat (%d x %d)", what, m->rows, m->cols);
    ISX_CHECK_ARG(m->step >= (size_t)m->cols * mat_elem_size(m->type), ISX_ERR_INVALID,
                  "%s: step %zu smaller than a row (%d cols of %s)", what, m->step, m->cols, type_name(m->type));
    return ISX_OK;
}

static int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);     // the staging buffer lives on the device the caller selected (hipSetDevice precedes every use)
    return dev;
}

int MatStage::use_in(const isx_mat* m, hipStream_t s, const char* what) {
    ISX_TRY(check_mat(m, what));
    host = nullptr;
    if (m->device >= 0) { d = *m; return ISX_OK; }
    size_t row = (size_t)m->cols * mat_elem_size(m->type);
    ISX_TRY(buf.reserve(row * m->rows));
    d = *m; d.data = buf.p; d.step = row; d.device = current_device();
    // a continuous mat (cv::Mat::isContinuous(), what imread / create produce) is ONE linear copy: the runtime's pitched 2-D copy
    // moves a 4K CV_8UC3 image row by row at about 1.6 GB/s, the linear one at the link's rate
    if (m->step == row) ISX_HIP(hipMemcpyAsync(buf.p, m->data, row * (size_t)m->rows, hipMemcpyHostToDevice, s));
    else ISX_HIP(hipMemcpy2DAsync(buf.p, row, m->data, m->step, row, m->rows, hipMemcpyHostToDevice, s));
    // cv::Mat semantics: the caller may free or overwrite a host mat as soon as the call returns
    // (W:305-308 clears the fed images before blend()), so the copy must have consumed it by then
    ISX_HIP(hipStreamSynchronize(s));
    return ISX_OK;
}
int MatStage::use_out(isx_mat* m, hipStream_t s, const char* what) {
    (void)s;
    ISX_TRY(check_mat(m, what));
    host = nullptr;
    if (m->device >= 0) { d = *m; return ISX_OK; }
    size_t row = (size_t)m->cols * mat_elem_size(m->type);
    ISX_TRY(buf.reserve(row * m->rows));
    d = *m; d.data = buf.p; d.step = row; d.device = current_device();
    host = m;
    return ISX_OK;
}
int MatStage::finish_out(hipStream_t s) {
    if (!host) return ISX_OK;
    size_t row = (size_t)host->cols * mat_elem_size(host->type);
    if (host->step == row && d.step == row) ISX_HIP(hipMemcpyAsync(host->data, d.data, row * (size_t)host->rows, hipMemcpyDeviceToHost, s));
    else ISX_HIP(hipMemcpy2DAsync(host->data, host->step, d.data, d.step, row, host->rows, hipMemcpyDeviceToHost, s));
    ISX_HIP(hipStreamSynchronize(s));
    return ISX_OK;
}

int MatStage::finish_out_cols(hipStream_t s, int col0, int col1) {
    if (!host) return ISX_OK;
    col0 = col0 < 0 ? 0 : col0; col1 = col1 > host->cols ? host->cols : col1;
    if (col1 <= col0) return ISX_OK;
    const size_t es = (size_t)mat_elem_size(host->type);
    ISX_HIP(hipMemcpy2DAsync((char*)host->data + (size_t)col0 * es, host->step, (const char*)d.data + (size_t)col0 * es, d.step, (size_t)(col1 - col0) * es,
                             host->rows, hipMemcpyDeviceToHost, s));
    ISX_HIP(hipStreamSynchronize(s));
    return ISX_OK;
}

// ---- profiler ---------------------------------------------------------------------------------
struct ProfEntry {
    std::string name;
    long long launches = 0;
    double ms = 0.0;
    double bytes = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
static std::mutex g_pm;
static bool g_prof = false;
static std::string g_filter;   // empty = every kernel
static int g_every = 1;        // bracket every g_every-th matching launch
static long long g_seen = 0;
static std::vector<ProfEntry> g_entries;
static std::map<std::string, int> g_index;
static std::vector<hipEvent_t> g_pool;

bool profiling_enabled() { return g_prof; }

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(const char* name, hipStream_t, double alg_bytes) {
    if (!g_prof) return;
    std::lock_guard<std::mutex> lk(g_pm);
    if (!g_filter.empty() && g_filter != name) return;
    if (g_every > 1 && (g_seen++ % g_every) != 0) return;
    auto it = g_index.find(name);
    if (it == g_index.end()) {
        g_entries.emplace_back();
        g_entries.back().name = name;
        slot = (int)g_entries.size() - 1;
        g_index[name] = slot;
    } else slot = it->second;
    ProfEntry& e = g_entries[slot];
    e.launches++;
    e.bytes += alg_bytes;
    start = get_event(); stop = get_event();
    e.pending.emplace_back(start, stop);
}

}  // namespace isx

using namespace isx;

extern "C" {

const char* isx_last_error(void) { return g_err.c_str(); }
const char* isx_version(void) { return "imagestitch_hip 0.1 (gfx950)"; }

int isx_device_count(int* count) ISX_ENTRY {
    ISX_CHECK_ARG(count != nullptr, ISX_ERR_INVALID, "isx_device_count: null pointer");
    ISX_HIP(hipGetDeviceCount(count));
    return ISX_OK;
} ISX_EXIT("isx_device_count")

int isx_profile_enable(int on) { g_prof = on != 0; return ISX_OK; }

// The barrier's own known-answer test (tests/test_abi_and_host.py, no GPU needed): throws `kind` from inside a guarded entry.
// 0 std::bad_alloc, 1 a real std::length_error (vector::reserve beyond max_size), 2 a real allocation failure (a vector of 2^62 bytes),
// 3 std::runtime_error, 4 a thrown int, 5 std::out_of_range from vector::at; anything else returns ISX_OK.
int isx_selftest_exception_barrier(int kind) ISX_ENTRY {
    clear_error();
    std::vector<char> v;
    switch (kind) {
        case 0: throw std::bad_alloc();
        case 1: v.reserve(v.max_size() + 1); break;
        case 2: v.resize((size_t)1 << 62); break;
        case 3: throw std::runtime_error("selftest");
        case 4: throw 42;
        case 5: return (int)v.at(7);
        default: break;
    }
    return ISX_OK;
} ISX_EXIT("isx_selftest_exception_barrier")

int isx_profile_filter(const char* kernel_name) ISX_ENTRY {
    std::lock_guard<std::mutex> lk(g_pm);
    g_filter = kernel_name ? kernel_name : "";
    return ISX_OK;
} ISX_EXIT("isx_profile_filter")

int isx_profile_sample(int every) ISX_ENTRY {
    ISX_CHECK_ARG(every >= 1, ISX_ERR_INVALID, "isx_profile_sample: every = %d", every);
    std::lock_guard<std::mutex> lk(g_pm);
    g_every = every;
    g_seen = 0;
    return ISX_OK;
} ISX_EXIT("isx_profile_sample")

int isx_profile_collect(void) ISX_ENTRY {
    ISX_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_pm);
    for (auto& e : g_entries) {
        for (auto& pr : e.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) e.ms += ms;
            g_pool.push_back(pr.first);
            g_pool.push_back(pr.second);
        }
        e.pending.clear();
    }
    return ISX_OK;
} ISX_EXIT("isx_profile_collect")

int isx_profile_reset(void) ISX_ENTRY {
    ISX_TRY(isx_profile_collect());
    std::lock_guard<std::mutex> lk(g_pm);
    g_entries.clear();
    g_index.clear();
    return ISX_OK;
} ISX_EXIT("isx_profile_reset")

int isx_profile_count(int* n) ISX_ENTRY {
    ISX_CHECK_ARG(n != nullptr, ISX_ERR_INVALID, "isx_profile_count: null pointer");
    std::lock_guard<std::mutex> lk(g_pm);
    *n = (int)g_entries.size();
    return ISX_OK;
} ISX_EXIT("isx_profile_count")

int isx_profile_entry(int index, const char** name, long long* launches, double* total_ms, double* alg_bytes) ISX_ENTRY {
    std::lock_guard<std::mutex> lk(g_pm);
    ISX_CHECK_ARG(index >= 0 && index < (int)g_entries.size(), ISX_ERR_INVALID, "isx_profile_entry: index %d out of range", index);
    const ProfEntry& e = g_entries[index];
    if (name) *name = e.name.c_str();
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.ms;
    if (alg_bytes) *alg_bytes = e.bytes;
    return ISX_OK;
} ISX_EXIT("isx_profile_entry")

}  // extern "C"
