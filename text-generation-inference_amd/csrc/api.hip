// Library info, error state and optional per-op HIP-event timing for libtgis_hip.so.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void tgis_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tgis_version(void) { return "tgis_hip 0.5 (gfx950)"; }
extern "C" const char* tgis_arch(void) { return "gfx950"; }
extern "C" const char* tgis_last_error(void) { return g_err; }
extern "C" void tgis_clear_error(void) {
    (void)hipGetLastError();  // the runtime's sticky per-thread error (e.g. left behind by an aborted graph capture)
    g_err[0] = 0;
}

extern "C" int tgis_device_info(int device, int* num_cus, int64_t* hbm_bytes, char* name, int name_len) {
    hipDeviceProp_t p;
    TGIS_CHECK_HIP(hipGetDeviceProperties(&p, device));
    if (num_cus) *num_cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return TGIS_OK;
}

// ---- timing -------------------------------------------------------------------------------------
struct EventPair {
    hipEvent_t a, b;
};
static std::mutex g_tmu;
static bool g_timing = false;
static std::vector<EventPair> g_pairs[TGIS_OP_COUNT];
static std::vector<EventPair> g_free;

TgisTimedScope::TgisTimedScope(int op_, hipStream_t s) : op(op_), stream(s), slot(nullptr) {
    if (!g_timing) return;
    std::lock_guard<std::mutex> lk(g_tmu);
    EventPair* p = new EventPair;
    if (!g_free.empty()) {
        *p = g_free.back();
        g_free.pop_back();
    } else {
        if (hipEventCreate(&p->a) != hipSuccess || hipEventCreate(&p->b) != hipSuccess) {
            delete p;
            return;
        }
    }
    (void)hipEventRecord(p->a, stream);
    slot = p;
}
TgisTimedScope::~TgisTimedScope() {
    if (!slot) return;
    EventPair* p = (EventPair*)slot;
    (void)hipEventRecord(p->b, stream);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_pairs[op].push_back(*p);
    delete p;
}

extern "C" int tgis_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timing = on != 0;
    return TGIS_OK;
}
extern "C" int tgis_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (int i = 0; i < TGIS_OP_COUNT; ++i) {
        for (auto& p : g_pairs[i]) g_free.push_back(p);
        g_pairs[i].clear();
    }
    return TGIS_OK;
}
extern "C" int tgis_timing_read(int op, int64_t* count, double* total_ms) {
    TGIS_CHECK_ARG(op >= 0 && op < TGIS_OP_COUNT, "tgis_timing_read: bad op %d", op);
    std::lock_guard<std::mutex> lk(g_tmu);
    double tot = 0;
    for (auto& p : g_pairs[op]) {
        TGIS_CHECK_HIP(hipEventSynchronize(p.b));
        float ms = 0;
        TGIS_CHECK_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        tot += ms;
    }
    if (count) *count = (int64_t)g_pairs[op].size();
    if (total_ms) *total_ms = tot;
    return TGIS_OK;
}
