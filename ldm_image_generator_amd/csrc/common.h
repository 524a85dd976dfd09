// Shared helpers for the gfx950 kernels (host side error plumbing + device utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/ldm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ldm_set_error(const char *fmt, ...);

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
