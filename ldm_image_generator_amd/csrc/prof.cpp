// hipEvent timing of the MFMA kernels for bench.py (include/ldm_hip.h, ldm_prof_*): when enabled, the kernel launch of every
// profiled entry point carries a (start, stop) event pair bound to its dispatch on ITS stream (hipExtLaunchKernelGGL: common.h,
// ldm_launch); reading synchronises the events and sums kernel time and algorithmic FLOPs per kernel class.  Host code only.
#include "common.h"
#include <mutex>
#include <vector>

namespace {

struct ProfRec {
    hipEvent_t start, stop;
    double flops, bytes;
    int cls;
};
std::mutex g_mu;
bool g_on = false;
std::vector<ProfRec> g_pool;
size_t g_used = 0;
constexpr int kVoid = -1000;                 // class of a record whose entry point launched nothing: never summed, never dumped

}  // namespace

void *ldm_prof_begin(int cls, double flops, hipStream_t st, double bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return nullptr;
    if (g_used == g_pool.size()) {
        ProfRec r{};
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return nullptr;
        g_pool.push_back(r);
    }
    ProfRec *rec = &g_pool[g_used++];
    rec->flops = flops;
    rec->bytes = bytes;
    rec->cls = cls;
    (void)st;
    LdmProfPending &pp = ldm_prof_pending();  // the entry's first ldm_launch binds the pair to its dispatch (common.h)
    pp.start = rec->start;
    pp.stop = rec->stop;
    pp.armed = true;
    return (void *)(size_t)(g_used);          // index + 1: the pool may reallocate
}

void ldm_prof_end(void *h, hipStream_t st)
{
    if (!h) return;
    (void)st;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t i = (size_t)h - 1;
    LdmProfPending &pp = ldm_prof_pending();
    if (pp.armed) {                           // the entry returned without launching (an argument error after the record was opened)
        pp.armed = false;
        if (i < g_used) g_pool[i].cls = kVoid;
    }
}

LdmProfPending &ldm_prof_pending()
{
    static thread_local LdmProfPending pp = {nullptr, nullptr, false};
    return pp;
}

static int prof_sum(int cls, long long *launches, double *ms, double *flops, double *bytes = nullptr)
{
    double tms = 0.0, tf = 0.0, tb = 0.0;
    long long n = 0;
    for (size_t i = 0; i < g_used; ++i) {
        if (g_pool[i].cls == kVoid || (cls >= 0 && g_pool[i].cls != cls)) continue;
        float e = 0.f;
        if (hipEventSynchronize(g_pool[i].stop) != hipSuccess || hipEventElapsedTime(&e, g_pool[i].start, g_pool[i].stop) != hipSuccess) {
            ldm_set_error("ldm_prof_read: event %zu not readable", i);
            return LDM_ELAUNCH;
        }
        tms += e;
        tf += g_pool[i].flops;
        tb += g_pool[i].bytes;
        ++n;
    }
    if (launches) *launches = n;
    if (ms) *ms = tms;
    if (flops) *flops = tf;
    if (bytes) *bytes = tb;
    return LDM_OK;
}

extern "C" int ldm_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    g_used = 0;
    return LDM_OK;
}

extern "C" int ldm_prof_read_class(int cls, long long *launches, double *ms, double *flops)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return prof_sum(cls, launches, ms, flops);
}

extern "C" int ldm_prof_read_bytes(int cls, double *bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return prof_sum(cls, nullptr, nullptr, nullptr, bytes);
}

extern "C" int ldm_prof_read(long long *launches, double *ms, double *flops)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const int rc = prof_sum(-1, launches, ms, flops);
    g_used = 0;
    return rc;
}

// every record of the pool in launch order: out[4 i .. 4 i + 3] = (class, kernel ms, algorithmic FLOPs, algorithmic bytes); returns the
// number of records written (at most max_records), negative on error.  Does not clear (tools: per-shape census of a forward).
extern "C" long long ldm_prof_dump(double *out, long long max_records)
{
    std::lock_guard<std::mutex> lk(g_mu);
    long long n = 0;
    for (size_t i = 0; i < g_used && n < max_records; ++i) {
        if (g_pool[i].cls == kVoid) continue;
        float e = 0.f;
        if (hipEventSynchronize(g_pool[i].stop) != hipSuccess || hipEventElapsedTime(&e, g_pool[i].start, g_pool[i].stop) != hipSuccess) {
            ldm_set_error("ldm_prof_dump: event %zu not readable", i);
            return LDM_ELAUNCH;
        }
        out[4 * n] = g_pool[i].cls;
        out[4 * n + 1] = e;
        out[4 * n + 2] = g_pool[i].flops;
        out[4 * n + 3] = g_pool[i].bytes;
        ++n;
    }
    return n;
}
