// Library-internal device scratch for fixed-order partial sums.
//
// The reductions of the training step (loss scalars, bias / column sums, stem / head weight gradients) used to finish with float
// atomics: fast to write, but the order of arrival decides the rounding, so two runs of the same step differed in the last bits
// (the reference on CPU is bit-reproducible).  They now write per-workgroup partials and a second kernel adds them in a
// FIXED order.  The partials need a few KiB ... MiB of device memory that no caller of the C ABI should have to size, so the library
// keeps ONE grow-only buffer per (device, stream): kernels enqueued on one stream run in order, hence a buffer per stream is never
// used by two launches at once.  Growth calls hipMalloc (synchronous; never inside a graph capture after the first, warm, step);
// the old buffer is freed with the stream drained first.  ldm_scratch_release() frees everything (tests, shutdown).
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace {
struct Buf {
    void *ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, Buf> g_bufs;
}  // namespace

void *ldm_scratch(hipStream_t st, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_mu);
    Buf &b = g_bufs[std::make_pair(dev, st)];
    if (b.bytes >= bytes && b.ptr) return b.ptr;
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    want = (want + (want >> 2) + 4095) & ~(size_t)4095;                 // 25 % head room: shapes vary a little from layer to layer
    if (b.ptr) {
        (void)hipStreamSynchronize(st);                                 // launches still reading the old buffer
        (void)hipFree(b.ptr);
        b.ptr = nullptr;
        b.bytes = 0;
    }
    if (hipMalloc(&b.ptr, want) != hipSuccess) {
        (void)hipGetLastError();
        b.ptr = nullptr;
        ldm_set_error("ldm_scratch: hipMalloc of %zu bytes failed", want);
        return nullptr;
    }
    b.bytes = want;
    return b.ptr;
}

extern "C" int ldm_scratch_release(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_bufs) {
        if (kv.second.ptr) {
            (void)hipStreamSynchronize(kv.first.second);
            (void)hipFree(kv.second.ptr);
        }
    }
    g_bufs.clear();
    return LDM_OK;
}
