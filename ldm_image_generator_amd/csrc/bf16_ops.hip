// HBM-bound kernels of the bf16 training step (BASELINE cfg 5): the producers that round GEMM operands ONCE to bf16, and the
// elementwise pieces whose inputs and outputs are all GEMM operands.  16-byte accesses per lane (8 bf16 / 4 fp32), wave-shuffle
// reductions, fp32 arithmetic inside.  The residual stream, its gradient, FiLM rows and all parameter gradients stay fp32.
#include "common.h"
#include <cmath>
#include <vector>

namespace {

constexpr int kMaxV = 8;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float lo, float hi)          // RNE, NaN stays NaN (v_cvt_pk_bf16_f32)
{
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo16(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ u32x2 pack4(const f32x4 &v) { return u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])}; }

__device__ __forceinline__ float group_sum(float v, int lpr)
{
    for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }
inline int pow2_lanes(int c4n)
{
    int lpr = 1;
    while (lpr < c4n && lpr < 64) lpr <<= 1;
    return lpr;
}

// ---- casts ---------------------------------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const f32x4 *__restrict__ x, u32x2 *__restrict__ out, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = pack4(x[i]);
}

__global__ void uncast_bf16_kernel(const u32x2 *__restrict__ x, f32x4 *__restrict__ out, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const u32x2 v = x[i];
    out[i] = f32x4{lo16(v[0]), hi16(v[0]), lo16(v[1]), hi16(v[1])};
}

// out[c][r] = bf16(x[r][c]): 64 x 64 tiles through LDS (padded rows), coalesced on both sides
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ out, long long R, int C)
{
    __shared__ float tile[64][65];
    const long long r0 = (long long)blockIdx.y * 64;
    const int c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const long long r = r0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? x[r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const long long r = r0 + tx;
        if (c < C && r < R) out[(long long)c * R + r] = (unsigned short)(pack2(tile[tx][i], 0.f) & 0xFFFFu);
    }
}

// ---- gate / relu, all operands bf16 ------------------------------------------------------------------------------------------
// hid = a * relu(b)   (modules.py:15), 8 elements per lane
__global__ void gate_fwd_bf16_kernel(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *__restrict__ out, long long n8)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const u32x4 av = a[i], bv = b[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2(lo16(av[e]) * fmaxf(lo16(bv[e]), 0.f), hi16(av[e]) * fmaxf(hi16(bv[e]), 0.f));
    out[i] = o;
}

__global__ void gate_bwd_bf16_kernel(const u32x4 *__restrict__ dh, const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *__restrict__ da,
                                     u32x4 *__restrict__ db, long long n8)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const u32x4 g = dh[i], av = a[i], bv = b[i];
    u32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g0 = lo16(g[e]), g1 = hi16(g[e]), a0 = lo16(av[e]), a1 = hi16(av[e]), b0 = lo16(bv[e]), b1 = hi16(bv[e]);
        oa[e] = pack2(g0 * fmaxf(b0, 0.f), g1 * fmaxf(b1, 0.f));
        ob[e] = pack2(b0 > 0.f ? g0 * a0 : 0.f, b1 > 0.f ? g1 * a1 : 0.f);
    }
    da[i] = oa;
    db[i] = ob;
}

__global__ void relu_bwd_bf16_kernel(const u32x4 *__restrict__ dy, const u32x4 *__restrict__ y, u32x4 *__restrict__ dx, long long n8)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const u32x4 g = dy[i], yv = y[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned keep = ((yv[e] << 16) != 0u && !(yv[e] & 0x8000u) ? 0x0000FFFFu : 0u) |      // y > 0 (a relu output: never negative, +0 or > 0)
                              ((yv[e] & 0xFFFF0000u) != 0u && !(yv[e] & 0x80000000u) ? 0xFFFF0000u : 0u);
        o[e] = g[e] & keep;
    }
    dx[i] = o;
}

// ---- ChannelNorm + FiLM, forward: fp32 in, fp32 and/or bf16 out ----------------------------------------------------------------
// NV float4 per lane and R row groups per wave (see channelnorm_film_rows_kernel in elementwise.hip: the loads of all R rows are in
// flight before the first reduction; arithmetic and its order do not depend on NV / R)
// four FiLM values (chunk c4 of a row) from fp32 rows or -- FBF: the bf16 training step keeps its per-(sample, pixel) FiLM rows in bf16 --
// from bf16 rows widened exactly
template <bool FBF>
__device__ __forceinline__ f32x4 film4(const void *row, int c4)
{
    if constexpr (FBF) {
        const u32x2 w = ((const u32x2 *)row)[c4];
        return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u)};
    } else {
        return ((const f32x4 *)row)[c4];
    }
}

template <int NV, int R, bool FBF = false>
__global__ __launch_bounds__(256) void channelnorm_film_mp_kernel(const float *__restrict__ x, const void *__restrict__ film, const int *__restrict__ slot,
                                                                  float *__restrict__ out32, unsigned short *__restrict__ out16, long long rows, int HW,
                                                                  int C, float eps, int lpr)
{
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int sub = lane % lpr;
    const int c4n = C >> 2;
    f32x4 v[R][NV];
    long long row[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        row[r] = (wave * R + r) * rpw + lane / lpr;
        live[r] = row[r] < rows;
        const f32x4 *xr = (const f32x4 *)(x + (live[r] ? row[r] : 0) * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = sub + i * lpr;
            v[r][i] = (live[r] && c4 < c4n) ? xr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        const float mean = group_sum(s, lpr) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (sub + i * lpr < c4n)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[r][i][e] - mean;
                    ss += d * d;
                }
        const float den = sqrtf(group_sum(ss, lpr) / (float)(C - 1) + eps);     // unbiased, modules.py:24
        if (!live[r]) continue;
        const int b = (int)(row[r] / HW), pix = (int)(row[r] - (long long)b * HW);
        const int sl = slot ? slot[b] : 0;
        const void *fr = (const char *)film + ((long long)sl * HW + pix) * 2 * C * (FBF ? 2 : 4);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = sub + i * lpr;
            if (c4 < c4n) {
                const f32x4 mu = film4<FBF>(fr, c4), bi = film4<FBF>(fr, c4n + c4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn((v[r][i][e] - mean) / den, mu[e]), bi[e]);
                if (out32) ((f32x4 *)(out32 + row[r] * C))[c4] = o;
                if (out16) ((u32x2 *)(out16 + row[r] * C))[c4] = pack4(o);
            }
        }
    }
}

// backward (one FiLM slot per sample: every (slot, pixel) row of dfilm has exactly one writer):
//   dfilm = (dxf * xn | dxf) as bf16;  dx = dres + (dxn - mean(dxn) - xn * sum(dxn * xn) / (C - 1)) / den, dxn = dxf * mul;  dx16 = bf16(dx)
template <bool FBF>
__global__ __launch_bounds__(256) void channelnorm_film_bwd_mp_kernel(const float *__restrict__ x, const void *__restrict__ film, const int *__restrict__ slot,
                                                                      const float *__restrict__ dxf, const float *__restrict__ dres, float *__restrict__ dx,
                                                                      unsigned short *__restrict__ dx16, unsigned short *__restrict__ dfilm16, long long rows,
                                                                      int HW, int C, float eps, int lpr)
{
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long row = wave * rpw + lane / lpr;
    const int sub = lane % lpr;
    const int c4n = C >> 2;
    const bool live = row < rows;
    const long long rr = live ? row : 0;
    const f32x4 *xr = (const f32x4 *)(x + rr * C);
    const f32x4 *gr = (const f32x4 *)(dxf + rr * C);
    const int b = (int)(rr / HW), pix = (int)(rr - (long long)b * HW);
    const int sl = slot ? slot[b] : 0;
    const long long frow = ((long long)sl * HW + pix) * 2 * C;
    const void *fr = (const char *)film + frow * (FBF ? 2 : 4);
    f32x4 v[kMaxV], g[kMaxV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
        const int c4 = sub + i * lpr;
        const bool ok = live && c4 < c4n;
        v[i] = ok ? xr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        g[i] = ok ? gr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = group_sum(s, lpr) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i)
        if (sub + i * lpr < c4n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                ss += d * d;
            }
    const float den = sqrtf(group_sum(ss, lpr) / (float)(C - 1) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
        const int c4 = sub + i * lpr;
        if (c4 < c4n) {
            const f32x4 mu = live ? film4<FBF>(fr, c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 gm;
            const f32x4 gx4 = g[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xn = (v[i][e] - mean) / den;
                const float gx = g[i][e];
                gm[e] = gx * xn;
                const float dxn = gx * mu[e];
                v[i][e] = xn;
                g[i][e] = dxn;
                s1 += dxn;
                s2 += dxn * xn;
            }
            if (live) {
                *(u32x2 *)(dfilm16 + frow + 4 * c4) = pack4(gm);
                *(u32x2 *)(dfilm16 + frow + C + 4 * c4) = pack4(gx4);
            }
        }
    }
    s1 = group_sum(s1, lpr) / (float)C;
    s2 = group_sum(s2, lpr) / (float)(C - 1);
    if (!live) return;
    const f32x4 *rres = dres ? (const f32x4 *)(dres + row * C) : nullptr;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
        const int c4 = sub + i * lpr;
        if (c4 < c4n) {
            f32x4 o = rres ? rres[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += (g[i][e] - s1 - v[i][e] * s2) / den;
            ((f32x4 *)(dx + row * C))[c4] = o;
            if (dx16) ((u32x2 *)(dx16 + row * C))[c4] = pack4(o);
        }
    }
}

}  // namespace

extern "C" int ldm_cast_bf16(const float *x, void *out, long long n, void *stream)
{
    LDM_REQUIRE(x && out && n > 0 && n % 4 == 0 && ldm_aligned16(x) && (((size_t)out) & 7) == 0, "ldm_cast_bf16: n=%lld must be a multiple of 4, pointers aligned", n);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (u32x2 *)out, n / 4);
    LDM_CHECK_LAUNCH("ldm_cast_bf16");
    return LDM_OK;
}

extern "C" int ldm_uncast_bf16(const void *x, float *out, long long n, void *stream)
{
    LDM_REQUIRE(x && out && n > 0 && n % 4 == 0 && ldm_aligned16(out) && (((size_t)x) & 7) == 0, "ldm_uncast_bf16: n=%lld must be a multiple of 4, pointers aligned", n);
    hipLaunchKernelGGL(uncast_bf16_kernel, dim3(blocks_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x2 *)x, (f32x4 *)out, n / 4);
    LDM_CHECK_LAUNCH("ldm_uncast_bf16");
    return LDM_OK;
}

extern "C" int ldm_transpose_cast_bf16(const float *x, void *out, long long R, int C, void *stream)
{
    LDM_REQUIRE(x && out && R > 0 && C > 0 && (R + 63) / 64 <= 65535, "ldm_transpose_cast_bf16: bad shape R=%lld C=%d", R, C);
    hipLaunchKernelGGL(transpose_cast_bf16_kernel, dim3((C + 63) / 64, (unsigned)((R + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short *)out, R, C);
    LDM_CHECK_LAUNCH("ldm_transpose_cast_bf16");
    return LDM_OK;
}

#define LDM_BF16_ELEMENTWISE(name, kern, ...)                                                                                                   \
    LDM_REQUIRE(n > 0 && n % 8 == 0, name ": n=%lld must be a positive multiple of 8", n);                                                      \
    hipLaunchKernelGGL(kern, dim3(blocks_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                                     \
    LDM_CHECK_LAUNCH(name);                                                                                                                     \
    return LDM_OK;

extern "C" int ldm_gate_fwd_bf16(const void *a, const void *b, void *out, long long n, void *stream)
{
    LDM_REQUIRE(a && b && out && ldm_aligned16(a) && ldm_aligned16(b) && ldm_aligned16(out), "ldm_gate_fwd_bf16: null / unaligned pointer");
    LDM_BF16_ELEMENTWISE("ldm_gate_fwd_bf16", gate_fwd_bf16_kernel, (const u32x4 *)a, (const u32x4 *)b, (u32x4 *)out, n / 8)
}

extern "C" int ldm_gate_bwd_bf16(const void *dh, const void *a, const void *b, void *da, void *db, long long n, void *stream)
{
    LDM_REQUIRE(dh && a && b && da && db && ldm_aligned16(dh) && ldm_aligned16(a) && ldm_aligned16(b) && ldm_aligned16(da) && ldm_aligned16(db),
                "ldm_gate_bwd_bf16: null / unaligned pointer");
    LDM_BF16_ELEMENTWISE("ldm_gate_bwd_bf16", gate_bwd_bf16_kernel, (const u32x4 *)dh, (const u32x4 *)a, (const u32x4 *)b, (u32x4 *)da, (u32x4 *)db, n / 8)
}

extern "C" int ldm_relu_bwd_bf16(const void *dy, const void *y, void *dx, long long n, void *stream)
{
    LDM_REQUIRE(dy && y && dx && ldm_aligned16(dy) && ldm_aligned16(y) && ldm_aligned16(dx), "ldm_relu_bwd_bf16: null / unaligned pointer");
    LDM_BF16_ELEMENTWISE("ldm_relu_bwd_bf16", relu_bwd_bf16_kernel, (const u32x4 *)dy, (const u32x4 *)y, (u32x4 *)dx, n / 8)
}

static int channelnorm_film_bf16_impl(const float *x, const void *film, bool film16, const int *slot, float *out_f32, void *out_bf16, int B, int HW, int C, float eps,
                                      void *stream);

extern "C" int ldm_channelnorm_film_bf16(const float *x, const float *film, const int *slot, float *out_f32, void *out_bf16, int B, int HW, int C, float eps,
                                         void *stream)
{
    return channelnorm_film_bf16_impl(x, film, false, slot, out_f32, out_bf16, B, HW, C, eps, stream);
}

// the same with the FiLM rows themselves in bf16 ([nslot, HW, 2C] bf16: the bf16 training step's per-(sample, pixel) rows)
extern "C" int ldm_channelnorm_film16_bf16(const float *x, const void *film_bf16, const int *slot, float *out_f32, void *out_bf16, int B, int HW, int C, float eps,
                                           void *stream)
{
    return channelnorm_film_bf16_impl(x, film_bf16, true, slot, out_f32, out_bf16, B, HW, C, eps, stream);
}

static int channelnorm_film_bf16_impl(const float *x, const void *film, bool film16, const int *slot, float *out_f32, void *out_bf16, int B, int HW, int C, float eps,
                                      void *stream)
{
    LDM_REQUIRE(x && film && (out_f32 || out_bf16), "ldm_channelnorm_film_bf16: null pointer");
    LDM_REQUIRE(B > 0 && HW > 0 && C >= 8 && C % 4 == 0 && C <= 64 * 4 * kMaxV, "ldm_channelnorm_film_bf16: bad shape B=%d HW=%d C=%d", B, HW, C);
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(film) && ldm_aligned16(out_f32) && (((size_t)out_bf16) & 7) == 0, "ldm_channelnorm_film_bf16: unaligned pointer");
    const int lpr = pow2_lanes(C / 4);
    const long long rows = (long long)B * HW;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    const int nv = (C / 4 + lpr - 1) / lpr;
    const int r = (nv <= 2 && waves >= 4 * 16384) ? 4 : 1;                 // as launch_channelnorm_film (elementwise.hip)
    const dim3 grid(blocks_for((waves + r - 1) / r, 4));
#define LDM_CNF16_LAUNCH(NV_, R_)                                                                                                      \
    do {                                                                                                                               \
        if (film16)                                                                                                                    \
            hipLaunchKernelGGL((channelnorm_film_mp_kernel<NV_, R_, true>), grid, dim3(256), 0, (hipStream_t)stream, x, film, slot, out_f32, \
                               (unsigned short *)out_bf16, rows, HW, C, eps, lpr);                                                     \
        else                                                                                                                           \
            hipLaunchKernelGGL((channelnorm_film_mp_kernel<NV_, R_, false>), grid, dim3(256), 0, (hipStream_t)stream, x, film, slot, out_f32, \
                               (unsigned short *)out_bf16, rows, HW, C, eps, lpr);                                                     \
    } while (0)
    if (nv == 1 && r == 4) LDM_CNF16_LAUNCH(1, 4);
    else if (nv == 2 && r == 4) LDM_CNF16_LAUNCH(2, 4);
    else if (nv == 1) LDM_CNF16_LAUNCH(1, 1);
    else if (nv == 2) LDM_CNF16_LAUNCH(2, 1);
    else if (nv <= 4) LDM_CNF16_LAUNCH(4, 1);
    else LDM_CNF16_LAUNCH(kMaxV, 1);
#undef LDM_CNF16_LAUNCH
    LDM_CHECK_LAUNCH("ldm_channelnorm_film_bf16");
    return LDM_OK;
}

static int channelnorm_film_bwd_bf16_impl(const float *x, const void *film, bool film16, const int *slot, const float *dxf, const float *dres, float *dx,
                                          void *dx_bf16, void *dfilm_bf16, int B, int HW, int C, float eps, void *stream);

extern "C" int ldm_channelnorm_film_bwd_bf16(const float *x, const float *film, const int *slot, const float *dxf, const float *dres, float *dx, void *dx_bf16,
                                             void *dfilm_bf16, int B, int HW, int C, float eps, void *stream)
{
    return channelnorm_film_bwd_bf16_impl(x, film, false, slot, dxf, dres, dx, dx_bf16, dfilm_bf16, B, HW, C, eps, stream);
}

extern "C" int ldm_channelnorm_film16_bwd_bf16(const float *x, const void *film_bf16, const int *slot, const float *dxf, const float *dres, float *dx, void *dx_bf16,
                                               void *dfilm_bf16, int B, int HW, int C, float eps, void *stream)
{
    return channelnorm_film_bwd_bf16_impl(x, film_bf16, true, slot, dxf, dres, dx, dx_bf16, dfilm_bf16, B, HW, C, eps, stream);
}

static int channelnorm_film_bwd_bf16_impl(const float *x, const void *film, bool film16, const int *slot, const float *dxf, const float *dres, float *dx,
                                          void *dx_bf16, void *dfilm_bf16, int B, int HW, int C, float eps, void *stream)
{
    LDM_REQUIRE(x && film && dxf && dx && dfilm_bf16, "ldm_channelnorm_film_bwd_bf16: null pointer");
    LDM_REQUIRE(B > 0 && HW > 0 && C >= 8 && C % 4 == 0 && C <= 64 * 4 * kMaxV, "ldm_channelnorm_film_bwd_bf16: bad shape");
    const int lpr = pow2_lanes(C / 4);
    const long long rows = (long long)B * HW;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    if (film16)
        hipLaunchKernelGGL(channelnorm_film_bwd_mp_kernel<true>, dim3(blocks_for(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, film, slot, dxf, dres, dx,
                           (unsigned short *)dx_bf16, (unsigned short *)dfilm_bf16, rows, HW, C, eps, lpr);
    else
        hipLaunchKernelGGL(channelnorm_film_bwd_mp_kernel<false>, dim3(blocks_for(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, film, slot, dxf, dres, dx,
                           (unsigned short *)dx_bf16, (unsigned short *)dfilm_bf16, rows, HW, C, eps, lpr);
    LDM_CHECK_LAUNCH("ldm_channelnorm_film_bwd_bf16");
    return LDM_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Encodings.proj1 in separable form (unet.py:18-20, training: one timestep per sample).  The input of proj1 is
// cat[pe(pixel), te(t_b)], so  proj1(cat) = W1[:, :C] pe(pixel) + W1[:, C:] te(t_b) + b1 = P[pixel] + T[b]:  two small GEMMs
// (HW and B rows) instead of one over B*HW rows -- 16 of the ~55 M*C^2 multiply-adds per pixel of a SwinBlock's forward.
//   forward : hid[b, pixel, :] = relu(P[pixel, :] + T[b, :])                                            (ldm_film_hidden)
//   backward: dhm = dh * (hid > 0);  dP[pixel] = sum_b dhm;  dT[b] = sum_pixel dhm                       (ldm_film_hidden_bwd)
// ------------------------------------------------------------------------------------------------------------------------
namespace {

template <bool OBF>
__global__ void film_hidden_kernel(const f32x4 *__restrict__ P, const f32x4 *__restrict__ Tt, void *__restrict__ out, int B, int HW, int n8)
{
    // one thread = 8 consecutive columns of one (sample, pixel) row
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * HW * n8;
    if (idx >= total) return;
    const int c8 = (int)(idx % n8);
    const long long row = idx / n8;
    const int pix = (int)(row % HW), b = (int)(row / HW);
    const f32x4 p0 = P[((long long)pix * n8 + c8) * 2], p1 = P[((long long)pix * n8 + c8) * 2 + 1];
    const f32x4 t0 = Tt[((long long)b * n8 + c8) * 2], t1 = Tt[((long long)b * n8 + c8) * 2 + 1];
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o0[e] = fmaxf(p0[e] + t0[e], 0.f);
        o1[e] = fmaxf(p1[e] + t1[e], 0.f);
    }
    if constexpr (OBF) {
        ((u32x4 *)out)[idx] = u32x4{pack2(o0[0], o0[1]), pack2(o0[2], o0[3]), pack2(o1[0], o1[1]), pack2(o1[2], o1[3])};
    } else {
        ((f32x4 *)out)[2 * idx] = o0;
        ((f32x4 *)out)[2 * idx + 1] = o1;
    }
}

// One WAVE owns 32 pixels x 64 columns and a range of samples: lane = (pixel & 7) * 8 + column chunk (8 columns), four
// sub-tiles of 8 pixels.  dP accumulates in registers over the samples; dT over the wave's 32 pixels (4 sub-tiles in
// registers, then 3 xor-shuffle steps over the pixel lanes).  No barriers, no atomics; partial planes
//   dP_part[z][HW][N] (z = sample chunk),  dT_part[pixel tile][B][N]   are summed by the caller in a fixed order.
template <bool BF>
__global__ __launch_bounds__(256) void film_hidden_bwd_kernel(const void *__restrict__ dh, const void *__restrict__ hid, float *__restrict__ dP_part,
                                                              float *__restrict__ dT_part, int B, int HW, int N, int ptiles, int bchunk)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);            // global wave id over (pixel tile, column group)
    const int ngroups = N >> 6;
    if (gw >= ptiles * ngroups) return;
    const int tile = gw / ngroups, grp = gw - tile * ngroups;
    const int z = blockIdx.y;
    const int b0 = z * bchunk, b1 = (b0 + bchunk < B) ? b0 + bchunk : B;
    const int psub = lane >> 3, chunk = lane & 7;
    const int col = grp * 64 + chunk * 8;
    float accP[4][8];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) accP[s][e] = 0.f;
    for (int b = b0; b < b1; ++b) {
        float accT[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) accT[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pix = tile * 32 + s * 8 + psub;
            float v[8];
            if (pix < HW) {
                const long long off = ((long long)b * HW + pix) * N + col;
                if constexpr (BF) {
                    const u32x4 g = *(const u32x4 *)((const unsigned short *)dh + off);
                    const u32x4 y = *(const u32x4 *)((const unsigned short *)hid + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = lo16(y[e]) > 0.f ? lo16(g[e]) : 0.f;
                        v[2 * e + 1] = hi16(y[e]) > 0.f ? hi16(g[e]) : 0.f;
                    }
                } else {
                    const f32x4 g0 = *(const f32x4 *)((const float *)dh + off), g1 = *(const f32x4 *)((const float *)dh + off + 4);
                    const f32x4 y0 = *(const f32x4 *)((const float *)hid + off), y1 = *(const f32x4 *)((const float *)hid + off + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = y0[e] > 0.f ? g0[e] : 0.f;
                        v[4 + e] = y1[e] > 0.f ? g1[e] : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                accP[s][e] += v[e];
                accT[e] += v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = accT[e];
            t += __shfl_xor(t, 8, 64);
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            accT[e] = t;
        }
        if (psub == 0) {
            float *dst = dT_part + ((long long)tile * B + b) * N + col;
            *(f32x4 *)dst = f32x4{accT[0], accT[1], accT[2], accT[3]};
            *(f32x4 *)(dst + 4) = f32x4{accT[4], accT[5], accT[6], accT[7]};
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pix = tile * 32 + s * 8 + psub;
        if (pix < HW) {
            float *dst = dP_part + ((long long)z * HW + pix) * N + col;
            *(f32x4 *)dst = f32x4{accP[s][0], accP[s][1], accP[s][2], accP[s][3]};
            *(f32x4 *)(dst + 4) = f32x4{accP[s][4], accP[s][5], accP[s][6], accP[s][7]};
        }
    }
}

}  // namespace

extern "C" int ldm_film_hidden(const float *P, const float *T, void *out, int out_bf16, int B, int HW, int N, void *stream)
{
    LDM_REQUIRE(P && T && out && B > 0 && HW > 0 && N > 0 && N % 8 == 0, "ldm_film_hidden: bad arguments (N %% 8 == 0)");
    LDM_REQUIRE(ldm_aligned16(P) && ldm_aligned16(T) && ldm_aligned16(out), "ldm_film_hidden: unaligned pointer");
    const long long total = (long long)B * HW * (N / 8);
    if (out_bf16)
        hipLaunchKernelGGL(film_hidden_kernel<true>, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)P, (const f32x4 *)T, out, B, HW, N / 8);
    else
        hipLaunchKernelGGL(film_hidden_kernel<false>, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)P, (const f32x4 *)T, out, B, HW, N / 8);
    LDM_CHECK_LAUNCH("ldm_film_hidden");
    return LDM_OK;
}

extern "C" int ldm_film_hidden_bwd_chunks(int B, int HW, int N)
{
    // sample chunks (grid.y): enough waves to fill the chip (>= ~2048) without more partial planes than needed
    const long long waves = (long long)((HW + 31) / 32) * (N / 64);
    int z = 1;
    while (waves * z < 2048 && z * 2 <= B && z < 32) z *= 2;
    return z;
}

extern "C" int ldm_film_hidden_bwd(const void *dh, const void *hid, int is_bf16, float *dP_part, float *dT_part, int B, int HW, int N, int zchunks,
                                   void *stream)
{
    LDM_REQUIRE(dh && hid && dP_part && dT_part && B > 0 && HW > 0 && N >= 64 && N % 64 == 0 && zchunks >= 1 && zchunks <= 65535,
                "ldm_film_hidden_bwd: bad arguments (N %% 64 == 0)");
    LDM_REQUIRE(ldm_aligned16(dh) && ldm_aligned16(hid) && ldm_aligned16(dP_part) && ldm_aligned16(dT_part), "ldm_film_hidden_bwd: unaligned pointer");
    const int ptiles = (HW + 31) / 32;
    const long long waves = (long long)ptiles * (N / 64);
    const int bchunk = (B + zchunks - 1) / zchunks;
    dim3 grid(blocks_for(waves, 4), zchunks);
    if (is_bf16)
        hipLaunchKernelGGL(film_hidden_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dh, hid, dP_part, dT_part, B, HW, N, ptiles, bchunk);
    else
        hipLaunchKernelGGL(film_hidden_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dh, hid, dP_part, dT_part, B, HW, N, ptiles, bchunk);
    LDM_CHECK_LAUNCH("ldm_film_hidden_bwd");
    return LDM_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// All bf16 weight copies of a training step in ONE launch (master weights are fp32 and change every optimizer step): job j is
// a row-major fp32 matrix [rows, cols]; the kernel writes bf16(W) [rows, cols] and bf16(W^T) [cols, rows] (either may be
// absent).  One workgroup = one 64 x 64 tile of one job, found by binary search in the tile prefix sums of the device-side
// job table.  ~300 launches of 8-10 us (mostly launch-bound, 5 ms per step) become one bandwidth-bound pass (< 1 ms).
// ------------------------------------------------------------------------------------------------------------------------
namespace {

struct CastJobDev {
    const float *src;
    unsigned short *dst, *dst_t;
    int rows, cols, tiles_c;
    int pad;
    long long tile0;          // first tile id of this job
};

__global__ __launch_bounds__(256) void multi_cast_kernel(const CastJobDev *__restrict__ jobs, int njobs)
{
    __shared__ float tile[64][65];
    const long long tid = blockIdx.x;
    int lo = 0, hi = njobs - 1;                          // last job with tile0 <= tid
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= tid) lo = mid; else hi = mid - 1;
    }
    const CastJobDev j = jobs[lo];
    const int t = (int)(tid - j.tile0);
    const int r0 = (t / j.tiles_c) * 64, c0 = (t % j.tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < j.rows && c < j.cols) ? j.src[(long long)r * j.cols + c] : 0.f;
        tile[i][tx] = v;
        if (j.dst && r < j.rows && c < j.cols) j.dst[(long long)r * j.cols + c] = (unsigned short)(pack2(v, 0.f) & 0xFFFFu);
    }
    if (!j.dst_t) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < j.cols && r < j.rows) j.dst_t[(long long)c * j.rows + r] = (unsigned short)(pack2(tile[tx][i], 0.f) & 0xFFFFu);
    }
}

}  // namespace

extern "C" size_t ldm_multi_cast_table_bytes(int njobs) { return (size_t)njobs * sizeof(CastJobDev); }

/* items: HOST array of njobs (src, dst, dst_t, rows, cols); table_dev: DEVICE scratch of ldm_multi_cast_table_bytes(njobs).
 * rebuild != 0 uploads the job table first (needed once, and again only when a pointer or shape changed); returns the tile
 * count through *tiles_io (pass the value back on later calls with rebuild == 0). */
extern "C" int ldm_multi_cast_bf16(const ldm_cast_job *items, int njobs, void *table_dev, int rebuild, long long *tiles_io, void *stream)
{
    LDM_REQUIRE(items && njobs > 0 && table_dev && tiles_io, "ldm_multi_cast_bf16: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (rebuild) {
        std::vector<CastJobDev> tab((size_t)njobs);
        long long tiles = 0;
        for (int i = 0; i < njobs; ++i) {
            LDM_REQUIRE(items[i].src && items[i].rows > 0 && items[i].cols > 0 && (items[i].dst || items[i].dst_t), "ldm_multi_cast_bf16: bad job %d", i);
            tab[i].src = items[i].src; tab[i].dst = (unsigned short *)items[i].dst; tab[i].dst_t = (unsigned short *)items[i].dst_t;
            tab[i].rows = (int)items[i].rows; tab[i].cols = items[i].cols; tab[i].tiles_c = (items[i].cols + 63) / 64; tab[i].pad = 0;
            tab[i].tile0 = tiles;
            tiles += (long long)((items[i].rows + 63) / 64) * tab[i].tiles_c;
        }
        LDM_REQUIRE(tiles > 0 && tiles <= 0x7fffffffLL, "ldm_multi_cast_bf16: too many tiles");
        // pageable host memory: hipMemcpyAsync returns once the staging copy is done, so `tab` may go out of scope
        if (hipMemcpyAsync(table_dev, tab.data(), tab.size() * sizeof(CastJobDev), hipMemcpyHostToDevice, st) != hipSuccess) {
            ldm_set_error("ldm_multi_cast_bf16: table upload failed");
            return LDM_ELAUNCH;
        }
        *tiles_io = tiles;
    }
    LDM_REQUIRE(*tiles_io > 0, "ldm_multi_cast_bf16: table was never built");
    hipLaunchKernelGGL(multi_cast_kernel, dim3((unsigned)*tiles_io), dim3(256), 0, st, (const CastJobDev *)table_dev, njobs);
    LDM_CHECK_LAUNCH("ldm_multi_cast_bf16");
    return LDM_OK;
}
