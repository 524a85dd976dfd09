// HBM-bound kernels of the UNet / VAE / DDIM path: coalesced 16-B accesses,
// wave-shuffle reductions, no LDS except for the two layout transposes.
#include "common.h"
#include <cmath>
#include <cstring>

static thread_local char g_err[512] = "";

void ldm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ldm_last_error(void) { return g_err; }
extern "C" int ldm_version(void) { return 100; }

extern "C" int ldm_device_ok(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

namespace {

constexpr int kMaxV = 8;   // float4 per lane per row -> C <= 2048

__device__ __forceinline__ float group_sum(float v, int lpr)
{
    for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// ChannelNorm + FiLM: one row (pixel) per group of `lpr` lanes; NV float4 per lane known at compile time and R row groups per
// wave: the R * NV loads of a lane are issued before the first reduction, so a wave keeps R times the bytes in flight (with one
// row group per wave -- ONE 16-byte load per lane between two shuffle trees at C = 128 -- the kernel ran at 3.9 TB/s at the
// B = 256 stage-0 shape).  Arithmetic and its order do not depend on NV / R.
// ---------------------------------------------------------------------------
template <int NV, int R>
__global__ __launch_bounds__(256) void channelnorm_film_rows_kernel(const float *__restrict__ x, const float *__restrict__ film,
                                                                    const int *__restrict__ slot, float *__restrict__ out,
                                                                    long long rows, int HW, int C, float eps, int lpr, int normalize)
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
        for (int i = 0; i < NV; ++i) {
            const int c4 = sub + i * lpr;
            if (c4 < c4n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[r][i][e] - mean;
                    ss += d * d;
                }
            }
        }
        const float var = group_sum(ss, lpr) / (float)(C - 1);
        const float den = sqrtf(var + eps);
        if (!live[r]) continue;
        const float mean_ = normalize ? mean : 0.f;
        const int b = (int)(row[r] / HW), pix = (int)(row[r] - (long long)b * HW);
        const int sl = slot ? slot[b] : 0;
        const f32x4 *fr = (const f32x4 *)(film + ((long long)sl * HW + pix) * 2 * C);
        f32x4 *orow = (f32x4 *)(out + row[r] * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = sub + i * lpr;
            if (c4 < c4n) {
                const f32x4 mu = fr[c4], bi = fr[c4n + c4];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xn = normalize ? (v[r][i][e] - mean_) / den : v[r][i][e];
                    o[e] = __fadd_rn(__fmul_rn(xn, mu[e]), bi[e]);
                }
                orow[c4] = o;
            }
        }
    }
}

static void launch_channelnorm_film(const float *x, const float *film, const int *slot, float *out, long long rows, int HW, int C, float eps, int lpr,
                                    int normalize, hipStream_t st)
{
    const int nv = (C / 4 + lpr - 1) / lpr;
    const long long per_wave = 64 / lpr;
    const long long waves1 = (rows + per_wave - 1) / per_wave;
    // several row groups per wave only while that leaves >= 16 k waves (measured: +29 % at stage 0, -10 % on the 16 k-row stages)
    const int r = (nv <= 2 && waves1 >= 4 * 16384) ? 4 : 1;
    const dim3 grid((unsigned)(((waves1 + r - 1) / r + 3) / 4));
#define LDM_CNF_LAUNCH(NV_, R_) \
    hipLaunchKernelGGL((channelnorm_film_rows_kernel<NV_, R_>), grid, dim3(256), 0, st, x, film, slot, out, rows, HW, C, eps, lpr, normalize)
    if (nv == 1 && r == 4) LDM_CNF_LAUNCH(1, 4);
    else if (nv == 2 && r == 4) LDM_CNF_LAUNCH(2, 4);
    else if (nv == 1) LDM_CNF_LAUNCH(1, 1);
    else if (nv == 2) LDM_CNF_LAUNCH(2, 1);
    else if (nv <= 4) LDM_CNF_LAUNCH(4, 1);
    else LDM_CNF_LAUNCH(kMaxV, 1);
#undef LDM_CNF_LAUNCH
}

// ---------------------------------------------------------------------------
// sin/cos position + time codes, emb[nT, HW, 2C]
// ---------------------------------------------------------------------------
__global__ void sincos_embed_kernel(const long long *__restrict__ t, int nT, int H, int W, int C,
                                    const float *__restrict__ pf, const float *__restrict__ tf, float *__restrict__ emb)
{
    const long long total = (long long)nT * H * W * 2 * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % (2 * C));
    const long long rowi = idx / (2 * C);
    const int pix = (int)(rowi % (H * W));
    const int ti = (int)(rowi / (H * W));
    const float pi = 3.14159265358979323846f;
    float val;
    if (c < C) {
        const int q = C >> 2;
        const int k = c % q, kind = c / q;            // 0 sin(row) 1 cos(row) 2 sin(col) 3 cos(col)
        const int y = pix / W, xx = pix - y * W;
        const float pos = (kind < 2) ? (float)y / (float)H : (float)xx / (float)W;   // sinusoidal.py:13-14
        const float arg = __fmul_rn(__fmul_rn(pos, pi), pf[k]);                       // (ev * pi) * f
        val = (kind & 1) ? cosf(arg) : sinf(arg);
    } else {
        const int cc = c - C, half = C >> 1;
        const int k = cc % half;
        const float arg = __fmul_rn(__fmul_rn((float)t[ti], pi), tf[k]);              // sinusoidal.py:36-37
        val = (cc >= half) ? cosf(arg) : sinf(arg);
    }
    emb[idx] = val;
}

// ---------------------------------------------------------------------------
__global__ void avgpool2_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ out, int B, int H, int W, int c4n)
{
    const int OH = H >> 1, OW = W >> 1;
    const long long total = (long long)B * OH * OW * c4n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const long long b = r / OH;
    const f32x4 *p = x + ((b * H + 2 * oy) * W + 2 * ox) * c4n + c4;
    const f32x4 a = p[0], bq = p[c4n], c = p[(long long)W * c4n], d = p[(long long)W * c4n + c4n];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (((a[e] + bq[e]) + c[e]) + d[e]) * 0.25f;
    out[idx] = o;
}

// stem: NCHW -> NHWC, 1x1 conv with tiny K.  W^T lives in LDS ([Cin][C0], read as conflict-free
// float4 rows); a thread owns 4 consecutive output channels and walks pixels; x values are
// wave-broadcast loads; stores are full 16-B, row-contiguous.
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ out, long long M,
                                                   int Cin, int HW, int C0, int pix_per_block)
{
    extern __shared__ __attribute__((aligned(16))) float wt[];      // [Cin][C0]
    const int t = threadIdx.x;
    for (int i = t; i < Cin * C0; i += 256) {
        const int n = i / Cin, ci = i - n * Cin;
        wt[ci * C0 + n] = w[i];
    }
    __syncthreads();
    const int n4n = C0 >> 2;
    const long long m0 = (long long)blockIdx.x * pix_per_block;
    for (int i = t; i < pix_per_block * n4n; i += 256) {
        const int pl = i / n4n, n4 = i - pl * n4n;
        const long long m = m0 + pl;
        if (m >= M) break;
        const long long b = m / HW;
        const int pix = (int)(m - b * HW);
        f32x4 acc = bias ? *(const f32x4 *)(bias + 4 * n4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float *xp = x + b * Cin * HW + pix;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xp[(long long)ci * HW];
            const f32x4 wv = *(const f32x4 *)(wt + ci * C0 + 4 * n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv, wv[e], acc[e]);
        }
        *(f32x4 *)(out + m * C0 + 4 * n4) = acc;
    }
}

// head: NHWC -> NCHW, ConvTranspose 1x1 (w [C0, Cin]); 64 pixels per block, C0 walked in chunks of 128.
// The chunk's weights are staged in LDS once per block and read as wave-wide broadcasts (wave cg owns outputs
// cg, cg + 4, cg + 8, cg + 12); the pixel tile is read conflict-free (row stride 129).
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ out, long long M,
                                                   int C0, int HW, int Cin)
{
    __shared__ float tile[64 * 129];
    __shared__ float wl[128 * 16];
    const int t = threadIdx.x, p = t & 63, cg = t >> 6;
    const long long m0 = (long long)blockIdx.x * 64;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};          // outputs cg, cg+4, cg+8, cg+12  (Cin <= 16)
    for (int c0 = 0; c0 < C0; c0 += 128) {
        const int cw = min(128, C0 - c0);
        if (cw == 128) {                          // 32 float4 per pixel row: coalesced 16-byte loads, no division
            for (int i = t; i < 64 * 32; i += 256) {
                const int rr = i >> 5, c4 = (i & 31) * 4;
                const long long m = m0 + rr;
                f32x4 v{0.f, 0.f, 0.f, 0.f};
                if (m < M) v = *(const f32x4 *)(x + m * C0 + c0 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[rr * 129 + c4 + e] = v[e];
            }
        } else {
            for (int i = t; i < 64 * cw; i += 256) {
                const int rr = i / cw, cc = i - rr * cw;
                const long long m = m0 + rr;
                tile[rr * 129 + cc] = m < M ? x[m * C0 + c0 + cc] : 0.f;
            }
        }
        for (int i = t; i < cw * Cin; i += 256) wl[i] = w[(long long)c0 * Cin + i];
        __syncthreads();
        for (int c = 0; c < cw; ++c) {
            const float xv = tile[p * 129 + c];
            const float *wr = wl + c * Cin;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = cg + 4 * k;
                if (co < Cin) acc[k] = fmaf(xv, wr[co], acc[k]);
            }
        }
        __syncthreads();
    }
    const long long m = m0 + p;
    if (m >= M) return;
    const long long b = m / HW;
    const int pix = (int)(m - b * HW);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = cg + 4 * k;
        if (co < Cin) out[(b * Cin + co) * HW + pix] = acc[k] + (bias ? bias[co] : 0.f);
    }
}

// head, wide-row form (C0 = 4 * LPR channels, LPR a power of two <= 64; CIN outputs): a group of LPR lanes owns one pixel row (16 bytes
// per lane, coalesced), keeps the weights of its 4 channels in registers, and the CIN dot products are finished by xor-shuffles.
// A wave walks 64 CONSECUTIVE pixels and parks their results in LDS, so the NCHW planes are written as 256-byte runs (the kernel
// above wrote them 4 bytes at a time from 64-pixel blocks: 1.2 TB/s at the B = 256 shape).
template <int CIN>
__global__ __launch_bounds__(256) void head_rows_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                        float *__restrict__ out, long long M, int C0, int HW, int lpr)
{
    __shared__ float res[4][CIN][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr;                       // rows per wave-iteration
    const int sub = lane % lpr, rsel = lane / lpr;
    const long long p0 = ((long long)blockIdx.x * 4 + wave) * 64;       // first pixel of this wave
    float wr[4][CIN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int co = 0; co < CIN; ++co) wr[c][co] = w[(long long)(4 * sub + c) * CIN + co];
    // eight row loads in flight per lane (a wave that waits for each 1-KiB load before the next keeps 16 KiB in flight per CU: latency-bound)
    constexpr int UN = 8;
    const int iters = 64 / rpw;                     // 2 ... 64, a power of two
    for (int it0 = 0; it0 < iters; it0 += UN) {
        f32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long m = p0 + (it0 + u) * rpw + rsel;
            v[u] = (it0 + u < iters && m < M) ? *(const f32x4 *)(x + m * C0 + 4 * sub) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (it0 + u >= iters) break;
            const int pl = (it0 + u) * rpw + rsel;
            float d[CIN];
#pragma unroll
            for (int co = 0; co < CIN; ++co) d[co] = fmaf(v[u][3], wr[3][co], fmaf(v[u][2], wr[2][co], fmaf(v[u][1], wr[1][co], v[u][0] * wr[0][co])));
#pragma unroll
            for (int co = 0; co < CIN; ++co) d[co] = group_sum(d[co], lpr);
            if (sub == 0) {
#pragma unroll
                for (int co = 0; co < CIN; ++co) res[wave][co][pl] = d[co];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    const long long m = p0 + lane;
    if (m >= M) return;
    const long long b = m / HW;
    const int pix = (int)(m - b * HW);
#pragma unroll
    for (int co = 0; co < CIN; ++co) out[(b * CIN + co) * HW + pix] = res[wave][co][lane] + (bias ? bias[co] : 0.f);
}

__global__ void ddim_update_kernel(float *__restrict__ x, const float *__restrict__ e, const float *__restrict__ noise,
                                   long long n, float s1, float s2, float s3, float s4, float sigma, int last)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ev = e[i];
    const float x0 = __fsub_rn(x[i], __fmul_rn(s1, ev)) / s2;                         // ddpm.py:82
    float r = x0;
    if (!last) {
        r = __fadd_rn(__fmul_rn(s3, x0), __fmul_rn(s4, ev));                          // ddpm.py:83-84,91
        r = __fadd_rn(r, __fmul_rn(sigma, noise ? noise[i] : 0.f));                   // ddpm.py:85
    }
    x[i] = r;
}

__global__ void qsample_kernel(const float *__restrict__ x, const float *__restrict__ e, const float *__restrict__ sa,
                               const float *__restrict__ sb, float *__restrict__ out, long long n, long long per)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long b = i / per;
    out[i] = __fadd_rn(__fmul_rn(sa[b], x[i]), __fmul_rn(sb[b], e[i]));               // ddpm.py:46
}

// to_rgb (C -> OC channels; 3 in every script of the reference, vae.py:100-103 allows any) + bilinear x2 accumulation of the previous
// stage's planes (NCHW)
template <int OC>
__global__ __launch_bounds__(256) void rgb_head_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias, const float *__restrict__ prev,
                                                       float *__restrict__ out, int B, int H, int W, int C, int lpr)
{
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const long long rows = (long long)B * H * W;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long row = wave * rpw + lane / lpr;
    const int sub = lane % lpr;
    const int c4n = C >> 2;
    const bool live = row < rows;
    const f32x4 *xr = (const f32x4 *)(x + (live ? row : 0) * C);
    float d[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) d[j] = 0.f;
    for (int c4 = sub; c4 < c4n; c4 += lpr) {
        const f32x4 v = live ? xr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 wj[OC];
#pragma unroll
        for (int j = 0; j < OC; ++j) wj[j] = ((const f32x4 *)(w + j * C))[c4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < OC; ++j) d[j] = fmaf(v[e], wj[j][e], d[j]);
    }
#pragma unroll
    for (int j = 0; j < OC; ++j) d[j] = group_sum(d[j], lpr);
    if (!live || sub != 0) return;
    const int HW = H * W;
    const long long b = row / HW;
    const int pix = (int)(row - b * HW);
    const int y = pix / W, xx = pix - y * W;
    float r[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) r[j] = d[j] + bias[j];
    if (prev) {
        const int PH = H >> 1, PW = W >> 1;
        // F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)  (vae.py:131)
        float sy = 0.5f * (float)y - 0.25f, sx = 0.5f * (float)xx - 0.25f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < PH - 1 ? 1 : 0), x1 = x0 + (x0 < PW - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int j = 0; j < OC; ++j) {
            const float *pp = prev + (b * OC + j) * PH * PW;
            const float top = hx * pp[y0 * PW + x0] + lx * pp[y0 * PW + x1];
            const float bot = hx * pp[y1 * PW + x0] + lx * pp[y1 * PW + x1];
            r[j] = (hy * top + ly * bot) + r[j];
        }
    }
#pragma unroll
    for (int j = 0; j < OC; ++j) out[(b * OC + j) * HW + pix] = r[j];
}

// [B, R, Cc] -> [B, Cc, R] for any R, Cc (32x32 LDS tiles); used for NCHW <-> NHWC
__global__ void transpose_kernel(const float *__restrict__ x, float *__restrict__ out, int R, int Cc)
{
    __shared__ float tile[32][33];
    const long long b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        const int rr = r0 + i, cc = c0 + tx;
        tile[i][tx] = (rr < R && cc < Cc) ? x[(b * R + rr) * Cc + cc] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int cc = c0 + i, rr = r0 + tx;
        if (rr < R && cc < Cc) out[(b * Cc + cc) * R + rr] = tile[tx][i];
    }
}

__global__ void to_uint8_kernel(const float *__restrict__ img, unsigned char *__restrict__ out, int B, int C, int HW)
{
    const long long total = (long long)B * C * HW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over the OUTPUT [B, HW, C]
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long r = idx / C;
    const int pix = (int)(r % HW);
    const long long b = r / HW;
    float v = img[(b * C + c) * HW + pix];
    v = fminf(fmaxf(v, -1.f), 1.f);                                              // sample_ldm.py:75
    v = __fadd_rn(__fmul_rn(v, 127.5f), 127.5f);                                 // sample_ldm.py:77
    out[idx] = (unsigned char)(int)v;                                            // astype(uint8): truncation
}

inline int pow2_lanes(int c4n)
{
    int lpr = 1;
    while (lpr < c4n && lpr < 64) lpr <<= 1;
    return lpr;
}

inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int ldm_channelnorm_film_f32(const float *x, const float *film, const int *slot, float *out, int B, int HW, int C,
                                        float eps, void *stream)
{
    LDM_REQUIRE(x && film && out, "ldm_channelnorm_film_f32: null pointer");
    LDM_REQUIRE(B > 0 && HW > 0 && C >= 8 && C % 4 == 0 && C <= 64 * 4 * kMaxV, "ldm_channelnorm_film_f32: bad shape B=%d HW=%d C=%d", B, HW, C);
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(film) && ldm_aligned16(out), "ldm_channelnorm_film_f32: unaligned pointer");
    const int lpr = pow2_lanes(C / 4);
    const long long rows = (long long)B * HW;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    (void)waves;
    launch_channelnorm_film(x, film, slot, out, rows, HW, C, eps, lpr, 1, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_channelnorm_film_f32");
    return LDM_OK;
}

extern "C" int ldm_film_f32(const float *x, const float *film, const int *slot, float *out, int B, int HW, int C, void *stream)
{
    LDM_REQUIRE(x && film && out, "ldm_film_f32: null pointer");
    LDM_REQUIRE(B > 0 && HW > 0 && C >= 8 && C % 4 == 0 && C <= 64 * 4 * kMaxV, "ldm_film_f32: bad shape B=%d HW=%d C=%d", B, HW, C);
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(film) && ldm_aligned16(out), "ldm_film_f32: unaligned pointer");
    const int lpr = pow2_lanes(C / 4);
    const long long rows = (long long)B * HW;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    (void)waves;
    launch_channelnorm_film(x, film, slot, out, rows, HW, C, 0.f, lpr, 0, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_film_f32");
    return LDM_OK;
}

extern "C" int ldm_sincos_embed_f32(const long long *t, int nT, int H, int W, int C, const float *pos_freq, const float *time_freq,
                                    float *emb, void *stream)
{
    LDM_REQUIRE(t && pos_freq && time_freq && emb, "ldm_sincos_embed_f32: null pointer");
    LDM_REQUIRE(nT > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ldm_sincos_embed_f32: bad shape nT=%d H=%d W=%d C=%d", nT, H, W, C);
    const long long total = (long long)nT * H * W * 2 * C;
    hipLaunchKernelGGL(sincos_embed_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, t, nT, H, W, C, pos_freq, time_freq, emb);
    LDM_CHECK_LAUNCH("ldm_sincos_embed_f32");
    return LDM_OK;
}

extern "C" int ldm_avgpool2_f32(const float *x, float *out, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(x && out, "ldm_avgpool2_f32: null pointer");
    LDM_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "ldm_avgpool2_f32: bad shape %d %d %d %d", B, H, W, C);
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(out), "ldm_avgpool2_f32: unaligned pointer");
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (f32x4 *)out, B, H, W, C / 4);
    LDM_CHECK_LAUNCH("ldm_avgpool2_f32");
    return LDM_OK;
}

extern "C" int ldm_stem_nchw_f32(const float *x, const float *w, const float *bias, float *out, int B, int Cin, int HW, int C0, void *stream)
{
    LDM_REQUIRE(x && w && out, "ldm_stem_nchw_f32: null pointer");
    LDM_REQUIRE(B > 0 && Cin > 0 && HW > 0 && C0 >= 4 && C0 % 4 == 0, "ldm_stem_nchw_f32: bad shape (C0 %% 4 == 0)");
    LDM_REQUIRE((size_t)Cin * C0 * sizeof(float) <= 64 * 1024, "ldm_stem_nchw_f32: Cin*C0 too large for the LDS weight tile");
    LDM_REQUIRE(ldm_aligned16(out) && (!bias || ldm_aligned16(bias)), "ldm_stem_nchw_f32: unaligned pointer");
    const long long M = (long long)B * HW;
    const int ppb = 64;
    hipLaunchKernelGGL(stem_kernel, dim3(blocks_for(M, ppb)), dim3(256), (size_t)Cin * C0 * sizeof(float), (hipStream_t)stream, x, w, bias, out, M, Cin, HW, C0, ppb);
    LDM_CHECK_LAUNCH("ldm_stem_nchw_f32");
    return LDM_OK;
}

extern "C" int ldm_head_nchw_f32(const float *x, const float *w, const float *bias, float *out, int B, int C0, int HW, int Cin, void *stream)
{
    LDM_REQUIRE(x && w && out, "ldm_head_nchw_f32: null pointer");
    LDM_REQUIRE(B > 0 && C0 > 0 && HW > 0 && Cin > 0 && Cin <= 16, "ldm_head_nchw_f32: bad shape (Cin=%d must be <= 16)", Cin);
    LDM_REQUIRE(C0 % 4 != 0 || ldm_aligned16(x), "ldm_head_nchw_f32: unaligned input");
    const long long M = (long long)B * HW;
    const int c4 = C0 / 4;
    if (Cin == 8 && C0 % 4 == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0 && ldm_aligned16(x))      // the UNet's head (128 -> 8): one pixel row per lane group
        hipLaunchKernelGGL(head_rows_kernel<8>, dim3(blocks_for(M, 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, M, C0, HW, c4);
    else
        hipLaunchKernelGGL(head_kernel, dim3(blocks_for(M, 64)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, M, C0, HW, Cin);
    LDM_CHECK_LAUNCH("ldm_head_nchw_f32");
    return LDM_OK;
}

extern "C" int ldm_ddim_update_f32(float *x, const float *e_theta, const float *noise, long long n, float s1, float s2, float s3,
                                   float s4, float sigma, int last, void *stream)
{
    LDM_REQUIRE(x && e_theta && n > 0, "ldm_ddim_update_f32: null pointer / empty");
    LDM_REQUIRE(noise || sigma == 0.f, "ldm_ddim_update_f32: sigma != 0 needs noise");
    hipLaunchKernelGGL(ddim_update_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, e_theta, noise, n, s1, s2, s3, s4, sigma, last);
    LDM_CHECK_LAUNCH("ldm_ddim_update_f32");
    return LDM_OK;
}

extern "C" int ldm_qsample_f32(const float *x, const float *e, const float *sa, const float *sb, float *out, int B, long long per_sample, void *stream)
{
    LDM_REQUIRE(x && e && sa && sb && out && B > 0 && per_sample > 0, "ldm_qsample_f32: bad arguments");
    const long long n = (long long)B * per_sample;
    hipLaunchKernelGGL(qsample_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, e, sa, sb, out, n, per_sample);
    LDM_CHECK_LAUNCH("ldm_qsample_f32");
    return LDM_OK;
}

extern "C" int ldm_rgb_head_f32(const float *x, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, void *stream)
{
    return ldm_rgb_head_oc_f32(x, w, bias, prev, out, B, H, W, C, 3, stream);
}

extern "C" int ldm_rgb_head_oc_f32(const float *x, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, int OC,
                                   void *stream)
{
    LDM_REQUIRE(x && w && bias && out, "ldm_rgb_head_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && OC >= 1 && OC <= 4, "ldm_rgb_head_f32: bad shape (1 <= output channels <= 4)");
    LDM_REQUIRE(!prev || (H % 2 == 0 && W % 2 == 0), "ldm_rgb_head_f32: prev needs even H, W");
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(w), "ldm_rgb_head_f32: unaligned pointer");
    const int lpr = pow2_lanes(C / 4) > 16 ? 16 : pow2_lanes(C / 4);
    const long long rows = (long long)B * H * W;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    const dim3 grid(blocks_for(waves, 4));
    if (OC == 3) hipLaunchKernelGGL(rgb_head_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, prev, out, B, H, W, C, lpr);
    else if (OC == 1) hipLaunchKernelGGL(rgb_head_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, prev, out, B, H, W, C, lpr);
    else if (OC == 2) hipLaunchKernelGGL(rgb_head_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, prev, out, B, H, W, C, lpr);
    else hipLaunchKernelGGL(rgb_head_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, prev, out, B, H, W, C, lpr);
    LDM_CHECK_LAUNCH("ldm_rgb_head_f32");
    return LDM_OK;
}

static int transpose_launch(const float *x, float *out, int B, int R, int Cc, void *stream, const char *who)
{
    LDM_REQUIRE(x && out && B > 0 && R > 0 && Cc > 0, "%s: bad arguments", who);
    LDM_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.z", who, B);
    dim3 grid((Cc + 31) / 32, (R + 31) / 32, B);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, R, Cc);
    LDM_CHECK_LAUNCH(who);
    return LDM_OK;
}

extern "C" int ldm_nchw_to_nhwc_f32(const float *x, float *out, int B, int C, int HW, void *stream)
{
    return transpose_launch(x, out, B, C, HW, stream, "ldm_nchw_to_nhwc_f32");
}

extern "C" int ldm_nhwc_to_nchw_f32(const float *x, float *out, int B, int C, int HW, void *stream)
{
    return transpose_launch(x, out, B, HW, C, stream, "ldm_nhwc_to_nchw_f32");
}

extern "C" int ldm_to_uint8_hwc(const float *img, unsigned char *out, int B, int C, int HW, void *stream)
{
    LDM_REQUIRE(img && out && B > 0 && C > 0 && HW > 0, "ldm_to_uint8_hwc: bad arguments");
    const long long total = (long long)B * C * HW;
    hipLaunchKernelGGL(to_uint8_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, C, HW);
    LDM_CHECK_LAUNCH("ldm_to_uint8_hwc");
    return LDM_OK;
}
