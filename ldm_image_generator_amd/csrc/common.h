// Shared helpers for the gfx950 kernels (host side error plumbing + device utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdarg>
#include <atomic>
#include "../../include/ldm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ldm_set_error(const char *fmt, ...);

// hipEvent profiler (prof.cpp): kernel classes of ldm_prof_read_class
#define LDM_PROF_GEMM      0   /* ldm_gemm_f32 (NT, exact fp32 / split schedule, grouped conv forward + data gradient) */
#define LDM_PROF_GEMM_TN   1   /* ldm_gemm_tn_f32 (weight gradients)                                                   */
#define LDM_PROF_GCONV_WG  2   /* ldm_gconv3x3_wgrad_f32                                                               */
#define LDM_PROF_GEMM_BF16 3   /* ldm_gemm_bf16 (NT, v_mfma_f32_32x32x16_bf16)                                         */
#define LDM_PROF_TN_BF16   4   /* ldm_gemm_tn_bf16                                                                     */
#define LDM_PROF_GCONV_BF16 5  /* grouped conv, bf16 operands                                                          */
void *ldm_prof_begin(int cls, double flops, hipStream_t st, double bytes = 0.0);      // NULL when profiling is off; bytes = algorithmic HBM bytes
void ldm_prof_end(void *h, hipStream_t st);
// How a profiled launch is timed.  ldm_prof_begin arms a (start, stop) event pair; the FIRST ldm_launch inside the entry point hands
// the pair to hipExtLaunchKernelGGL, which binds both events to the dispatch itself: hipEventElapsedTime then reads the kernel's own
// begin / end timestamps and no marker packet enters the queue.  Measured (tools/event_overhead.hip, 109-us kernels back to back):
// hipEventRecord before and after every launch costs 7.8 us per launch and reports 2.9 us too much; the bound pair costs 5.0 us
// and reports the kernel time.  Further launches of the same entry (a split-K epilogue) run untimed.
struct LdmProfPending {
    hipEvent_t start, stop;
    bool armed;
};
LdmProfPending &ldm_prof_pending();          // thread-local (prof.cpp)
template <typename K, typename... A>
inline void ldm_launch(K kern, dim3 grid, dim3 block, size_t smem, hipStream_t st, A... args)
{
    LdmProfPending &pp = ldm_prof_pending();
    if (pp.armed) {
        pp.armed = false;
        hipExtLaunchKernelGGL(kern, grid, block, smem, st, pp.start, pp.stop, 0, args...);
    } else {
        hipLaunchKernelGGL(kern, grid, block, smem, st, args...);
    }
}

// grow-only device scratch of this (device, stream) for fixed-order partial sums (scratch.cpp); NULL (and ldm_last_error set) on failure
void *ldm_scratch(hipStream_t st, size_t bytes);

#define LDM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ldm_set_error(__VA_ARGS__);        \
            return LDM_EINVAL;                 \
        }                                      \
    } while (0)

#define LDM_CHECK_LAUNCH(what)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            ldm_set_error("%s: %s", what, hipGetErrorString(e_));                \
            return LDM_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// Per-device one-time setup.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only, so a process
// that drives several GPUs must repeat it per device; the masks are atomics (bit = device ordinal), so concurrent first calls
// at worst both set the attribute.  One process per GPU (bench.py, torchrun) only ever sets bit `local rank`.
struct LdmLdsOptIn {
    std::atomic<unsigned long long> ok{0}, bad{0};
    // true when the current device accepts `bytes` of dynamic LDS for `kern`
    bool operator()(const void *kern, size_t bytes)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (ok.load(std::memory_order_acquire) & bit) return true;
        if (bad.load(std::memory_order_acquire) & bit) return false;
        const bool good = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
        if (!good) (void)hipGetLastError();
        (good ? ok : bad).fetch_or(bit, std::memory_order_release);
        return good;
    }
};

// compute units of the current device (cached per device ordinal)
static inline int ldm_cu_count()
{
    static std::atomic<int> cache[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

static inline bool ldm_aligned16(const void *p) { return (((unsigned long long)p) & 15ull) == 0; }

// Bijective XCD-aware remap of a linear workgroup id (blocks b and b+8 share an
// XCD under round-robin dispatch): every XCD gets a contiguous run of logical
// tiles, so neighbouring tiles that share operand panels hit the same L2.
__device__ __forceinline__ int xcd_remap(int id, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = id & 7, within = id >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}
