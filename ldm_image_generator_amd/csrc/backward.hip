// Backward-pass kernels of the UNet training step (ddpm.py:39-48 + autograd of unet.py / modules.py /
// attention.py).  All GEMM-shaped gradients reuse ldm_gemm_f32 (data grads: transposed weights; weight
// grads: transposed activations + split-K over grid groups); this file holds what is not a GEMM.
#include "common.h"
#include <cmath>

namespace {

constexpr int kMaxV = 8;

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

// ---- elementwise ------------------------------------------------------------------------------
__global__ void gate_fwd_kernel(const f32x4 *a, const f32x4 *b, f32x4 *out, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 av = a[i], bv = b[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = av[e] * fmaxf(bv[e], 0.f);
    out[i] = o;
}

__global__ void gate_bwd_kernel(const f32x4 *dh, const f32x4 *a, const f32x4 *b, f32x4 *da, f32x4 *db, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 g = dh[i], av = a[i], bv = b[i];
    f32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        oa[e] = g[e] * fmaxf(bv[e], 0.f);                 // d/da  a*relu(b)
        ob[e] = bv[e] > 0.f ? g[e] * av[e] : 0.f;         // d/db
    }
    da[i] = oa;
    db[i] = ob;
}

__global__ void relu_bwd_kernel(const f32x4 *dy, const f32x4 *y, f32x4 *dx, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 g = dy[i], yv = y[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = yv[e] > 0.f ? g[e] : 0.f;
    dx[i] = o;
}

__global__ void add_kernel(f32x4 *y, const f32x4 *x, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 a = y[i], b = x[i];
    y[i] = f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
}

// column sums of a row-major [M, N] matrix (bias gradients): slab blockIdx.y writes ITS sums to plane blockIdx.y of `out` ([slabs][N]);
// the planes are added in slab order by reduce_planes (no float atomics: bit-reproducible)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, float *__restrict__ out, long long M, int N, int slab)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long long m0 = (long long)blockIdx.y * slab;
    const long long m1 = m0 + slab < M ? m0 + slab : M;
    float s = 0.f;
    for (long long m = m0; m < m1; ++m) s += x[m * N + n];
    out[(long long)blockIdx.y * N + n] = s;
}

__global__ void reduce_partials_kernel(const float *__restrict__ parts, float *__restrict__ out, int S, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += parts[(long long)k * n + i];
    out[i] = s;
}

// Same sum for n % 4 == 0, laid out for memory-level parallelism: a block owns 64 float4 columns, its four waves each sum a
// contiguous quarter of the S planes (four independent 16-byte loads in flight per lane) and wave 0 adds the four quarter sums
// in plane order.  Fixed association ((q0 + q1) + q2) + q3, no atomics: the result does not depend on scheduling.  The scalar
// kernel above had one dependent 4-byte load chain per element: 15 us per launch on average over the ~330 weight-gradient
// reductions of a training step (S = 4 ... 128 planes of 50 k ... 3 M elements), most of it latency.
// row4 / seg4 (optional, float4 units): the summed [rows, row4] matrix is stored as [row4 / seg4][rows][seg4] -- the column blocks of
// a weight gradient that belongs to several parameters (the c-weights of the three ReGLUs of a block) land as contiguous tensors
__device__ __forceinline__ void reduce_partials_v4_body(const f32x4 *__restrict__ parts, f32x4 *__restrict__ out, int S, long long n4, long long block,
                                                        long long row4 = 0, long long seg4 = 0)
{
    __shared__ f32x4 part[3][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = block * 64 + lane;
    const bool live = i < n4;
    const int s0 = (int)((long long)S * q / 4), s1 = (int)((long long)S * (q + 1) / 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const f32x4 *src = parts + (long long)s0 * n4 + i;
        int k = s0;
        for (; k + 4 <= s1; k += 4) {
            const f32x4 v0 = src[0], v1 = src[n4], v2 = src[2 * n4], v3 = src[3 * n4];
            acc += v0;
            acc += v1;
            acc += v2;
            acc += v3;
            src += 4 * n4;
        }
        for (; k < s1; ++k) {
            acc += src[0];
            src += n4;
        }
    }
    if (q) part[q - 1][lane] = acc;
    __syncthreads();
    if (q == 0 && live) {
        acc += part[0][lane];
        acc += part[1][lane];
        acc += part[2][lane];
        long long dst = i;
        if (seg4) {
            const long long row = i / row4, k4 = i - row * row4, e = k4 / seg4;
            dst = e * (n4 / row4) * seg4 + row * seg4 + (k4 - e * seg4);
        }
        out[dst] = acc;
    }
}

__global__ __launch_bounds__(256) void reduce_partials_v4_kernel(const f32x4 *__restrict__ parts, f32x4 *__restrict__ out, int S, long long n4)
{
    reduce_partials_v4_body(parts, out, S, n4, blockIdx.x);
}

// two sums with the same S in one launch (a weight gradient's planes and its bias gradient's): blocks [0, blocks_a) take job a
__global__ __launch_bounds__(256) void reduce_partials_pair_kernel(const f32x4 *__restrict__ pa, f32x4 *__restrict__ oa, long long na4, unsigned blocks_a,
                                                                   const f32x4 *__restrict__ pb, f32x4 *__restrict__ ob, long long nb4, int S,
                                                                   long long row4, long long seg4)
{
    if (blockIdx.x < blocks_a) reduce_partials_v4_body(pa, oa, S, na4, blockIdx.x, row4, seg4);
    else reduce_partials_v4_body(pb, ob, S, nb4, blockIdx.x - blocks_a);
}

// ---- ChannelNorm + FiLM backward ----------------------------------------------------------------
// xf = xn * mul + bias, xn = (x - mean) / sqrt(var_unbiased + eps)
//   dfilm[slot, pix, c] += dxf * xn ; dfilm[slot, pix, C + c] += dxf            (atomic: samples sharing a slot)
//   dx = dres + (dxn - mean(dxn) - xn * sum(dxn * xn) / (C - 1)) / den ,  dxn = dxf * mul
__global__ __launch_bounds__(256) void channelnorm_film_bwd_kernel(const float *__restrict__ x, const float *__restrict__ film,
                                                                   const int *__restrict__ slot, const float *__restrict__ dxf,
                                                                   const float *__restrict__ dres, float *__restrict__ dx,
                                                                   float *__restrict__ dfilm, long long rows, int HW, int C,
                                                                   float eps, int lpr, int unique)
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
    const f32x4 *fr = (const f32x4 *)(film + frow);
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
            const f32x4 mu = live ? fr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && unique) {                 // every (slot, pixel) row belongs to exactly one sample: plain 16-byte stores
                f32x4 gm;
#pragma unroll
                for (int e = 0; e < 4; ++e) gm[e] = g[i][e] * ((v[i][e] - mean) / den);
                *(f32x4 *)(dfilm + frow + 4 * c4) = gm;
                *(f32x4 *)(dfilm + frow + C + 4 * c4) = g[i];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xn = (v[i][e] - mean) / den;
                const float gx = g[i][e];
                if (live && !unique) {
                    atomicAdd(dfilm + frow + 4 * c4 + e, gx * xn);
                    atomicAdd(dfilm + frow + C + 4 * c4 + e, gx);
                }
                const float dxn = gx * mu[e];
                v[i][e] = xn;
                g[i][e] = dxn;
                s1 += dxn;
                s2 += dxn * xn;
            }
        }
    }
    s1 = group_sum(s1, lpr) / (float)C;
    s2 = group_sum(s2, lpr) / (float)(C - 1);
    if (!live) return;
    const f32x4 *rres = dres ? (const f32x4 *)(dres + row * C) : nullptr;
    f32x4 *orow = (f32x4 *)(dx + row * C);
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
        const int c4 = sub + i * lpr;
        if (c4 < c4n) {
            f32x4 o = rres ? rres[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += (g[i][e] - s1 - v[i][e] * s2) / den;
            orow[c4] = o;
        }
    }
}

// ---- pooling ------------------------------------------------------------------------------------
// backward of AvgPool2d(2): dx[b, 2y+dy, 2x+dx, :] (+)= 0.25 * dlo[b, y, x, :]
__global__ void avgpool2_bwd_kernel(const f32x4 *__restrict__ dlo, f32x4 *__restrict__ dx, int B, int OH, int OW, int c4n, int accumulate)
{
    const long long total = (long long)B * OH * OW * c4n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const long long b = r / OH;
    f32x4 g = dlo[idx];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= 0.25f;
    const int W = 2 * OW;
    f32x4 *p = dx + ((b * 2 * OH + 2 * oy) * W + 2 * ox) * c4n + c4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        f32x4 *q = p + ((d >> 1) * (long long)W + (d & 1)) * c4n;
        if (accumulate) {
            const f32x4 o = *q;
            *q = f32x4{o[0] + g[0], o[1] + g[1], o[2] + g[2], o[3] + g[3]};
        } else {
            *q = g;
        }
    }
}

// backward of nearest Upsample(x2): dlo[b, y, x, :] = sum of the 4 fine gradients
__global__ void sumpool2_kernel(const f32x4 *__restrict__ dhi, f32x4 *__restrict__ dlo, int B, int OH, int OW, int c4n)
{
    const long long total = (long long)B * OH * OW * c4n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const long long b = r / OH;
    const int W = 2 * OW;
    const f32x4 *p = dhi + ((b * 2 * OH + 2 * oy) * W + 2 * ox) * c4n + c4;
    const f32x4 a = p[0], bq = p[c4n], c = p[(long long)W * c4n], d = p[(long long)W * c4n + c4n];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((a[e] + bq[e]) + c[e]) + d[e];
    dlo[idx] = o;
}

// ---- stem / head ---------------------------------------------------------------------------------
// stem: y[m, n] = sum_ci x[b, ci, p] w[n, ci] + bias[n]   ->  dw[n, ci] += sum_m dy[m, n] x[b, ci, p]
__global__ __launch_bounds__(256) void stem_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dw,
                                                       long long M, int Cin, int HW, int C0, int slab)
{
    const long long m0 = (long long)blockIdx.x * slab;
    const long long m1 = m0 + slab < M ? m0 + slab : M;
    for (int idx = threadIdx.x; idx < C0 * Cin; idx += 256) {
        const int n = idx / Cin, ci = idx - n * Cin;
        float s = 0.f;
        for (long long m = m0; m < m1; ++m) {
            const long long b = m / HW;
            const int pix = (int)(m - b * HW);
            s = fmaf(dy[m * C0 + n], x[(b * Cin + ci) * HW + pix], s);
        }
        dw[(long long)blockIdx.x * C0 * Cin + idx] = s;                  // plane blockIdx.x of the partial sums
    }
}

// Same gradient with a thread per OUTPUT channel (C0 <= 256, Cin <= 16: every net of the reference): dy rows are read coalesced, the
// chunk's x pixels sit in LDS as [row][CP] (CP = Cin padded to 4 / 8 / 16, read back as 16-byte broadcasts), CP accumulators per
// thread.  ONE workgroup of 1024 threads per CU: same-address float atomics retire at ~0.2 us each on this chip (measured: 247 us
// with 1 k blocks, 429 us with 2 k), so the block count IS the run time once the loads are coalesced; the kernel above spends
// 715 us at the cfg-5 shape (a 64-bit division per multiply, 2 M atomics).
constexpr int kRowsNT = 1024;
template <int CP>
__global__ __launch_bounds__(kRowsNT) void stem_bwd_rows_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dw,
                                                                long long M, int Cin, int HW, int C0, int slab)
{
    __shared__ __attribute__((aligned(16))) float xs[kRowsNT * CP];
    const int t = threadIdx.x;
    const int nrg = kRowsNT / C0;                               // row groups: threads [rg * C0, (rg + 1) * C0) take rows rg, rg + nrg, ...
    const int n = t % C0, rg = t / C0;
    const bool active = rg < nrg;
    const long long m0 = (long long)blockIdx.x * slab;
    const long long m1 = m0 + slab < M ? m0 + slab : M;
    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;
    for (long long c0 = m0; c0 < m1; c0 += kRowsNT) {
        __syncthreads();
        const long long m = c0 + t;
        if (m < m1) {
            const long long b = m / HW;
            const long long pix = m - b * HW;
#pragma unroll
            for (int c = 0; c < CP; ++c) xs[t * CP + c] = c < Cin ? x[(b * Cin + c) * HW + pix] : 0.f;
        }
        __syncthreads();
        const int rows = (int)(m1 - c0 < kRowsNT ? m1 - c0 : kRowsNT);
        if (active) {
            const float *dyp = dy + c0 * C0 + n;
#pragma unroll 8
            for (int r = rg; r < rows; r += nrg) {
                const float d = dyp[(long long)r * C0];
#pragma unroll
                for (int q = 0; q < CP / 4; ++q) {
                    const f32x4 v = *(const f32x4 *)(xs + r * CP + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * q + e] = fmaf(d, v[e], acc[4 * q + e]);
                }
            }
        }
    }
    // sum the row groups through LDS (reusing xs), then one atomic per weight and block
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CP; ++c) xs[t * CP + c] = active ? acc[c] : 0.f;
    __syncthreads();
    for (int i = t; i < C0 * Cin; i += kRowsNT) {
        const int nn = i / Cin, c = i - nn * Cin;
        float sum = 0.f;
        for (int g = 0; g < nrg; ++g) sum += xs[(g * C0 + nn) * CP + c];
        dw[(long long)blockIdx.x * C0 * Cin + i] = sum;                  // plane blockIdx.x of the partial sums
    }
}

// head: out[b, co, p] = sum_c x[m, c] w[c, co] + bias[co]
//   dx[m, c] = sum_co dout[b, co, p] w[c, co];  dw[c, co] += sum_m x[m, c] dout[b, co, p];  db[co] += sum dout
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dout,
                                                       float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db, long long M,
                                                       int C0, int HW, int Cin)
{
    __shared__ float gt[64 * 17];          // dout tile [64 pixels][Cin <= 16]
    const int t = threadIdx.x;
    const long long m0 = (long long)blockIdx.x * 64;
    for (int i = t; i < 64 * Cin; i += 256) {
        const int co = i / 64, pl = i - co * 64;
        const long long m = m0 + pl;
        float v = 0.f;
        if (m < M) {
            const long long b = m / HW;
            v = dout[(b * Cin + co) * HW + (m - b * HW)];
        }
        gt[pl * 17 + co] = v;
    }
    __syncthreads();
    // data gradient
    for (int i = t; i < 64 * C0; i += 256) {
        const int pl = i / C0, c = i - pl * C0;
        const long long m = m0 + pl;
        if (m >= M) continue;
        float s = 0.f;
        for (int co = 0; co < Cin; ++co) s = fmaf(gt[pl * 17 + co], w[c * Cin + co], s);
        dx[m * C0 + c] = s;
    }
    // weight / bias gradient partials of this 64-row slab
    for (int i = t; i < C0 * Cin; i += 256) {
        const int c = i / Cin, co = i - c * Cin;
        float s = 0.f;
        for (int pl = 0; pl < 64; ++pl) {
            const long long m = m0 + pl;
            if (m < M) s = fmaf(x[m * C0 + c], gt[pl * 17 + co], s);
        }
        dw[(long long)blockIdx.x * (C0 * Cin + Cin) + i] = s;            // plane blockIdx.x: [C0 * Cin weight sums | Cin bias sums]
    }
    if (t < Cin) {
        float s = 0.f;
        for (int pl = 0; pl < 64; ++pl) s += gt[pl * 17 + t];
        dw[(long long)blockIdx.x * (C0 * Cin + Cin) + C0 * Cin + t] = s;
    }
}

// The same three gradients with a thread per INPUT channel c (C0 <= 256, Cin <= 16): one pass over the rows reads x[m][c] and writes
// dx[m][c] coalesced, the tile's dout values come from LDS as 16-byte broadcasts, w[c][:] and the dw[c][:] partial sums stay in
// registers across the 256-row tiles a block walks.  One workgroup of 1024 threads per CU and one atomic per weight and block (see
// stem_bwd_rows_kernel: same-address atomics are what the kernel above -- 8 k blocks -- spends its 640 us on at the cfg-5 shape).
// dx is bit-identical to the kernel above (same fmaf order over co).
template <int CP>
__global__ __launch_bounds__(kRowsNT) void head_bwd_rows_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dout,
                                                                float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db, long long M,
                                                                int C0, int HW, int Cin, long long ntiles)
{
    constexpr int TR = 256;                                             // rows per tile
    __shared__ __attribute__((aligned(16))) float sm[kRowsNT * CP];     // dout tile [TR][CP]; at the end the row groups' partial sums
    const int t = threadIdx.x;
    const int nrg = kRowsNT / C0;
    const int c = t % C0, rg = t / C0;
    const bool active = rg < nrg;
    float wreg[CP], acc[CP], accb = 0.f;
#pragma unroll
    for (int co = 0; co < CP; ++co) {
        wreg[co] = (active && co < Cin) ? w[c * Cin + co] : 0.f;
        acc[co] = 0.f;
    }
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * TR;
        __syncthreads();
        for (int i = t; i < TR * CP; i += kRowsNT) {
            const int co = i / TR, pl = i - co * TR;
            const long long m = m0 + pl;
            float v = 0.f;
            if (m < M && co < Cin) {
                const long long b = m / HW;
                v = dout[(b * Cin + co) * HW + (m - b * HW)];
            }
            sm[pl * CP + co] = v;
        }
        __syncthreads();
        const int rows = (int)(M - m0 < TR ? M - m0 : TR);
        if (active) {
#pragma unroll 8
            for (int pl = rg; pl < rows; pl += nrg) {
                const long long m = m0 + pl;
                const float xv = x[m * C0 + c];
                float g[CP];
#pragma unroll
                for (int q = 0; q < CP / 4; ++q) {
                    const f32x4 v = *(const f32x4 *)(sm + pl * CP + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[4 * q + e] = v[e];
                }
                float s_ = 0.f;
#pragma unroll
                for (int co = 0; co < CP; ++co) {
                    if (co < Cin) s_ = fmaf(g[co], wreg[co], s_);
                    acc[co] = fmaf(xv, g[co], acc[co]);
                }
                dx[m * C0 + c] = s_;
            }
        }
        if (t < Cin) {
            float s_ = 0.f;
            for (int pl = 0; pl < rows; ++pl) s_ += sm[pl * CP + t];
            accb += s_;
        }
    }
    __syncthreads();
#pragma unroll
    for (int co = 0; co < CP; ++co) sm[t * CP + co] = active ? acc[co] : 0.f;
    __syncthreads();
    for (int i = t; i < C0 * Cin; i += kRowsNT) {
        const int cc = i / Cin, co = i - cc * Cin;
        float sum = 0.f;
        for (int g2 = 0; g2 < nrg; ++g2) sum += sm[(g2 * C0 + cc) * CP + co];
        dw[(long long)blockIdx.x * (C0 * Cin + Cin) + i] = sum;          // plane blockIdx.x: [C0 * Cin weight sums | Cin bias sums]
    }
    if (t < Cin) dw[(long long)blockIdx.x * (C0 * Cin + Cin) + C0 * Cin + t] = accb;
}

// ---- L1 loss (ddpm.py:47 with nn.L1Loss) ----------------------------------------------------------
// per-block sums to parts[blockIdx.x]; l1_loss_finish_kernel adds them in a fixed order (no float atomics: the loss is bit-reproducible)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float *__restrict__ p, const float *__restrict__ q, long long n, float *__restrict__ parts)
{
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += fabsf(p[i] - q[i]);
    s = group_sum(s, 64);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(256) void l1_loss_finish_kernel(const float *__restrict__ parts, int nparts, float inv_n, float *__restrict__ loss)
{
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += parts[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sh[0] * inv_n;
}

__global__ void l1_loss_bwd_kernel(const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ gscale, float inv_n,
                                   float *__restrict__ grad, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = p[i] - q[i];
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    grad[i] = sgn * gscale[0] * inv_n;
}

// [R, Cc] -> [Cc, R] and, in the same pass, csum[c] += sum_r x[r][c] (the bias gradient that always accompanies a
// weight-gradient GEMM over the transposed activation): 32x32 LDS tiles, one atomic per column per tile
__global__ __launch_bounds__(256) void transpose_colsum_kernel(const float *__restrict__ x, float *__restrict__ out, float *__restrict__ csum,
                                                               long long R, int Cc)
{
    __shared__ float tile[32][33];
    __shared__ float part[8][32];
    const long long r0 = (long long)blockIdx.y * 32;
    const int c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float s = 0.f;
    for (int i = ty; i < 32; i += 8) {
        const long long rr = r0 + i;
        const int cc = c0 + tx;
        const float v = (rr < R && cc < Cc) ? x[rr * Cc + cc] : 0.f;
        tile[i][tx] = v;
        s += v;
    }
    part[ty][tx] = s;
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int cc = c0 + i;
        const long long rr = r0 + tx;
        if (rr < R && cc < Cc) out[(long long)cc * R + rr] = tile[tx][i];
    }
    if (ty == 0 && c0 + tx < Cc) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][tx];
        csum[(long long)blockIdx.y * Cc + c0 + tx] = t;                   // plane blockIdx.y of the partial column sums
    }
}

// ---- grouped-conv weight gradient helper: transposed im2col -----------------------------------------
// out[g][tap*32 + ci][m] = x[(pixel m shifted by tap)][g*32 + ci]  (0 outside the image);  x is [B,H,W,C]
__global__ __launch_bounds__(256) void im2col3x3_t_kernel(const float *__restrict__ x, float *__restrict__ out, int B, int H, int W, int C)
{
    __shared__ float tile[32][33];
    const int g = blockIdx.z, tap = blockIdx.y;
    const long long M = (long long)B * H * W;
    const long long m0 = (long long)blockIdx.x * 32;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {           // row i of the tile = pixel m0 + i, column tx = channel
        const long long m = m0 + i;
        float v = 0.f;
        if (m < M) {
            const int xx = (int)(m % W), yy = (int)((m / W) % H);
            if ((unsigned)(yy + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W)
                v = x[(m + (long long)dy * W + dx) * C + g * 32 + tx];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    float *o = out + ((long long)g * 288 + tap * 32) * M;
    for (int i = ty; i < 32; i += 8) {           // write row ci = i, contiguous over m
        const long long m = m0 + tx;
        if (m < M) o[(long long)i * M + m] = tile[tx][i];
    }
}

// ---- window attention backward -----------------------------------------------------------------------
struct AttnB {
    const float *qkv, *bias, *xf, *dctx;
    float *dqkv, *dbias_pad;
    // bf16 training step: q, k, v / dO arrive as bf16 rows and dq, dk, dv leave as bf16 rows (then the fp32 pointers above are NULL); the
    // float "mask" of shifted windows comes from the bf16 normalised input.  The arithmetic in between is unchanged (fp32 MFMA).
    const unsigned short *qkv16, *dctx16, *xf16;
    unsigned short *dqkv16;
    float *pad_parts;            // [total_waves][64]: every wave's (dk | dv) sums over ITS zero-padded tokens (zeros if it has none); added per
                                 // head in wave order by attn_pad_finish_kernel -- no float atomics, the bias gradient is bit-reproducible
    int B, H, W, C, ws, shift;
    int Hp, Wp, nwh, nww, heads, L, global;
    long long total_waves;
};

__device__ __forceinline__ bool tok_src(const AttnB &p, int wr, int wc, int j, int &sy, int &sx, int &py, int &px)
{
    if (p.global) {
        sy = py = j / p.W;
        sx = px = j - sy * p.W;
        return true;
    }
    const int wy = j / p.ws, wx = j - wy * p.ws;
    py = wr * p.ws + wy;
    px = wc * p.ws + wx;
    sy = py - p.shift;
    sy += sy < 0 ? p.Hp : 0;
    sx = px - p.shift;
    sx += sx < 0 ? p.Wp : 0;
    return sy < p.H && sx < p.W;
}

// One wave per (sample, window, head), lane i = token i (query AND key role).  LDS per wave:
// region A [2][L][32]: K, V during phase 1, then overwritten by Qs (scaled q), dO for phase 2;
// region B [2][L][L+1]: P, dS.  20 KB per wave -> 8 waves per CU.
template <int LMAX>
__global__ __launch_bounds__(128) void window_attention_bwd_kernel(const AttnB p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RS = 36;                         // row stride of the [L][32] images: 16-B aligned, conflict-free b128 row writes
    constexpr int ROW = LMAX * RS, MAT = LMAX * (LMAX + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *Ks = smem + wave * (2 * ROW + 2 * MAT + LMAX + 4);
    float *Vs = Ks + ROW, *Qs = Ks, *dOs = Vs, *Ps = Vs + ROW, *dSs = Ps + MAT, *Kb = dSs + MAT;
    const int L = p.L, C = p.C;
    const long long gw = (long long)blockIdx.x * 2 + wave;
    const bool active = gw < p.total_waves;
    const int head = (int)(gw % p.heads);
    const long long t1 = gw / p.heads;
    const int nwin = p.global ? 1 : p.nwh * p.nww;
    const int win = (int)(t1 % nwin);
    const long long b = t1 / nwin;
    const int wr = win / p.nww, wc = win - wr * p.nww;
    const long long img = b * p.H * p.W;
    const float scale = 0.17677669529663687f;

    int sy = 0, sx = 0, py = 0, px = 0;
    bool ok = false;
    f32x4 qr[8], gr[8];                            // this lane's scaled query and output gradient
#pragma unroll
    for (int c = 0; c < 8; ++c) qr[c] = gr[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active && lane < L) {
        ok = tok_src(p, wr, wc, lane, sy, sx, py, px);
        const long long tokrow = img + (long long)sy * p.W + sx;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 qv, kv, vv;
            if (ok) {
                const float *row = p.qkv + tokrow * 3 * C + head * 32 + 4 * c;
                qv = *(const f32x4 *)row;
                kv = *(const f32x4 *)(row + C);
                vv = *(const f32x4 *)(row + 2 * C);
                gr[c] = *(const f32x4 *)(p.dctx + tokrow * C + head * 32 + 4 * c);
            } else {                              // zero-padded token (projection of 0 = bias); its output is cropped: dO = 0
                qv = *(const f32x4 *)(p.bias + head * 32 + 4 * c);
                kv = *(const f32x4 *)(p.bias + C + head * 32 + 4 * c);
                vv = *(const f32x4 *)(p.bias + 2 * C + head * 32 + 4 * c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) qr[c][e] = qv[e] * scale;
            *(f32x4 *)(Ks + lane * RS + 4 * c) = kv;
            *(f32x4 *)(Vs + lane * RS + 4 * c) = vv;
        }
        float kb = 0.f;
        if (!p.global) {
            if (p.shift == 0) {
                kb = ok ? 0.f : -INFINITY;
            } else {
                int my = (py - 2 * p.shift) % p.Hp, mx = (px - 2 * p.shift) % p.Wp;
                my += my < 0 ? p.Hp : 0;
                mx += mx < 0 ? p.Wp : 0;
                kb = (my < p.H && mx < p.W) ? p.xf[(img + (long long)my * p.W + mx) * C] : 0.f;
            }
        }
        Kb[lane] = kb;
    }
    __syncthreads();
    // ---- phase 1 (lane = query i): P[i][:], dS[i][:], dq_i ----------------------------------------
    if (active && lane < L) {
        float s[LMAX], dp[LMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) {
                float a = 0.f, g = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 kv = *(const f32x4 *)(Ks + j * RS + 4 * c), vv = *(const f32x4 *)(Vs + j * RS + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a = fmaf(qr[c][e], kv[e], a);
                        g = fmaf(gr[c][e], vv[e], g);
                    }
                }
                a += Kb[j];
                s[j] = a;
                dp[j] = g;
                mx = fmaxf(mx, a);
            }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) {
                s[j] = expf(s[j] - mx);
                sum += s[j];
            }
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) {
                s[j] = s[j] / sum;
                dot = fmaf(s[j], dp[j], dot);
            }
        f32x4 dq[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) dq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < LMAX; ++j)
            if (j < L) {
                const float ds = s[j] * (dp[j] - dot);
                Ps[lane * (LMAX + 1) + j] = s[j];
                dSs[lane * (LMAX + 1) + j] = ds;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 kv = *(const f32x4 *)(Ks + j * RS + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[c][e] = fmaf(ds, kv[e], dq[c][e]);
                }
            }
        if (ok) {
            float *o = p.dqkv + (img + (long long)sy * p.W + sx) * 3 * C + head * 32;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                *(f32x4 *)(o + 4 * c) = f32x4{dq[c][0] * scale, dq[c][1] * scale, dq[c][2] * scale, dq[c][3] * scale};
        }
    }
    __syncthreads();
    if (active && lane < L) {                     // K, V are no longer needed: region A now holds Qs, dO
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            *(f32x4 *)(Qs + lane * RS + 4 * c) = qr[c];
            *(f32x4 *)(dOs + lane * RS + 4 * c) = gr[c];
        }
    }
    __syncthreads();
    // ---- phase 2 (lane = key j): dk_j = sum_i dS[i][j] qs_i ; dv_j = sum_i P[i][j] dO_i ----------------
    f32x4 dk[8], dv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dk[c] = dv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool padded = active && lane < L && !ok;
    if (active && lane < L) {
        for (int i = 0; i < L; ++i) {
            const float ds = dSs[i * (LMAX + 1) + lane], pp = Ps[i * (LMAX + 1) + lane];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 qv = *(const f32x4 *)(Qs + i * RS + 4 * c), gv = *(const f32x4 *)(dOs + i * RS + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dk[c][e] = fmaf(ds, qv[e], dk[c][e]);
                    dv[c][e] = fmaf(pp, gv[e], dv[c][e]);
                }
            }
        }
        if (ok) {
            float *o = p.dqkv + (img + (long long)sy * p.W + sx) * 3 * C + head * 32;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                *(f32x4 *)(o + C + 4 * c) = dk[c];
                *(f32x4 *)(o + 2 * C + 4 * c) = dv[c];
            }
        }
    }
    // zero-padded tokens: k, v are the in-proj bias, so their dk / dv belong to its gradient.  Sum them over the wave's
    // padded tokens through region A (free now) so that each wave issues 64 atomics, not 64 per padded token.
    const bool any_pad = __any(padded) != 0;      // wave-uniform
    __syncthreads();                              // every lane of the block is past its reads of region A
    if (any_pad) {
        if (active && lane < L) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 z{0.f, 0.f, 0.f, 0.f};
                *(f32x4 *)(Qs + lane * RS + 4 * c) = padded ? dk[c] : z;
                *(f32x4 *)(dOs + lane * RS + 4 * c) = padded ? dv[c] : z;
            }
        }
    }
    __syncthreads();
    if (any_pad) {
        const float *src = (lane < 32 ? Qs : dOs) + (lane & 31);
        float acc = 0.f;
        for (int r = 0; r < L; ++r) acc += src[r * RS];
        if (active) p.pad_parts[gw * 64 + lane] = acc;
    } else if (active) {
        p.pad_parts[gw * 64 + lane] = 0.f;
    }
}

// eight consecutive values of a bf16 row, widened exactly
__device__ __forceinline__ void widen8(const unsigned short *src, f32x4 &lo, f32x4 &hi)
{
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v w = *(const u32x4v *)src;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        lo[2 * e] = __uint_as_float(w[e] << 16);
        lo[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
        hi[2 * e] = __uint_as_float(w[2 + e] << 16);
        hi[2 * e + 1] = __uint_as_float(w[2 + e] & 0xFFFF0000u);
    }
}
// four fp32 values -> four bf16 (RNE), one 8-byte store
__device__ __forceinline__ void store4_bf16(unsigned short *dst, const f32x4 &v)
{
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const f32x2v a = {v[0], v[1]}, b = {v[2], v[3]};
    *(u32x2v *)dst = u32x2v{__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2v)), __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2v))};
}

// ---- MFMA version (v_mfma_f32_16x16x4_f32, exact fp32; every window of the reference: L <= 48) ----------------------------------
// One wave per (sample, window, head) as above, but the five products run on the matrix pipe (the scalar kernel is VALU-bound:
// 1.6 ms at the cfg-5 stage-0 shape against an HBM floor of 0.4 ms).  Operand conventions of the forward kernel
// (attention.hip): a lane (c = lane & 15, g = lane >> 4) holds row 16 t + c of a token tile, dims [8 g, 8 g + 8); the C/D map gives
// a lane ONE column (c) and rows {4 g + e}.  A C/D-layout matrix can be the B operand of the next product directly (contraction
// index = its rows 16 t + 4 g + e); it cannot be the A operand.  Hence BOTH orientations of the score matrix are computed from the
// same fragments (operands swapped) instead of transposing through LDS:
//   S^T = K Qs^T, dP^T = V dO^T  ->  softmax down the columns  ->  dS^T  ->  dQ^T = K^T dS^T   (K^T from an LDS image of K)
//   S   = Qs K^T, dP   = dO V^T  ->  softmax along the rows     ->  P, dS ->  dK^T = Qs^T dS, dV^T = dO^T P   (Qs, dO images)
// Results leave as 16-byte stores (token c, dims [16 dt + 4 g, + 4)).  Padded tokens' dk / dv go to the bias gradient (one shuffle
// tree + 8 atomics per lane group and matrix).
// DPP lane permutation inside rows of 16 lanes: 0xB1 / 0x4E quad_perm [1,0,3,2] / [2,3,0,1], 0x141 row_half_mirror, 0x140 row_mirror
template <int CTRL>
__device__ __forceinline__ float attn_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int NT>
__global__ __launch_bounds__(128, 2) void window_attention_bwd_mfma_kernel(const AttnB p)
{
    constexpr int RS = 36, LT = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int L = p.L, C = p.C;
    const int IMG = p.L * RS;                      // images hold the L real rows; reads past them are clamped (their partner is an exact 0)
    float *Ks = smem + wave * (3 * IMG + LT);
    float *Ql = Ks + IMG, *Gl = Ql + IMG, *Kb = Gl + IMG;
    const long long gw = (long long)blockIdx.x * 2 + wave;
    const bool active = gw < p.total_waves;
    const int head = (int)(gw % p.heads);
    const long long t1 = gw / p.heads;
    const int nwin = p.global ? 1 : p.nwh * p.nww;
    const int win = (int)(t1 % nwin);
    const long long b = t1 / nwin;
    const int wr = win / p.nww, wc = win - wr * p.nww;
    const long long img = b * p.H * p.W;
    const float scale = 0.17677669529663687f;

    f32x4 kf[NT][2], qf[NT][2], vf[NT][2], gf[NT][2];
    long long trow[NT];
    bool tok[NT], tpad[NT];                       // real token / zero-padded token (inside L, outside the image)
    if (active) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + c;
            int sy = 0, sx = 0, py = 0, px = 0;
            const bool ok = j < L && tok_src(p, wr, wc, j, sy, sx, py, px);
            tok[t] = ok;
            tpad[t] = j < L && !ok;
            trow[t] = img + (long long)sy * p.W + sx;
            f32x4 qv2[2];
            if (p.qkv16) {
                if (ok) {
                    const unsigned short *row16 = p.qkv16 + trow[t] * 3 * C + head * 32 + 8 * g;
                    widen8(row16, qv2[0], qv2[1]);
                    widen8(row16 + C, kf[t][0], kf[t][1]);
                    widen8(row16 + 2 * C, vf[t][0], vf[t][1]);
                    widen8(p.dctx16 + trow[t] * C + head * 32 + 8 * g, gf[t][0], gf[t][1]);
                } else {                                   // zero-padded token: q, k, v = the bias as the bf16 projection stored it; dO = 0
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float *bq = p.bias + head * 32 + 8 * g + 4 * u;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            qv2[u][e] = (float)(__bf16)bq[e];
                            kf[t][u][e] = (float)(__bf16)bq[C + e];
                            vf[t][u][e] = (float)(__bf16)bq[2 * C + e];
                            gf[t][u][e] = 0.f;
                        }
                    }
                }
            } else {
                const float *row = ok ? p.qkv + trow[t] * 3 * C + head * 32 : p.bias + head * 32;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 z{0.f, 0.f, 0.f, 0.f};
                    qv2[u] = *(const f32x4 *)(row + 8 * g + 4 * u);
                    kf[t][u] = *(const f32x4 *)(row + C + 8 * g + 4 * u);
                    vf[t][u] = *(const f32x4 *)(row + 2 * C + 8 * g + 4 * u);
                    gf[t][u] = ok ? *(const f32x4 *)(p.dctx + trow[t] * C + head * 32 + 8 * g + 4 * u) : z;   // cropped outputs: dO = 0
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[t][u][e] = qv2[u][e] * scale;
                if (j < L) {
                    *(f32x4 *)(Ks + j * RS + 8 * g + 4 * u) = kf[t][u];
                    *(f32x4 *)(Ql + j * RS + 8 * g + 4 * u) = qf[t][u];
                    *(f32x4 *)(Gl + j * RS + 8 * g + 4 * u) = gf[t][u];
                }
            }
        }
        if (lane < LT) {
            float kb = -INFINITY;
            if (lane < L) {
                int sy, sx, py, px;
                const bool ok = tok_src(p, wr, wc, lane, sy, sx, py, px);
                kb = 0.f;
                if (!p.global) {
                    if (p.shift == 0) {
                        kb = ok ? 0.f : -INFINITY;
                    } else {
                        int my = (py - 2 * p.shift) % p.Hp, mx = (px - 2 * p.shift) % p.Wp;
                        my += my < 0 ? p.Hp : 0;
                        mx += mx < 0 ? p.Wp : 0;
                        if (my < p.H && mx < p.W) {
                            const long long mi = (img + (long long)my * p.W + mx) * C;
                            kb = p.xf16 ? __uint_as_float((unsigned)p.xf16[mi] << 16) : p.xf[mi];
                        } else {
                            kb = 0.f;
                        }
                    }
                }
            }
            Kb[lane] = kb;
        }
    }
    __syncthreads();
    if (!active) return;

    auto zero9 = [](f32x4 (&m)[NT][NT]) {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int bq = 0; bq < NT; ++bq) m[a][bq] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // ---- column orientation: S^T[k][q], dP^T[k][q]  ->  dS^T  ->  dQ ------------------------------------------------
    {
        f32x4 st[NT][NT], dpt[NT][NT];
        zero9(st);
        zero9(dpt);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) {
                    st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][s8 >> 2][s8 & 3], qf[qt][s8 >> 2][s8 & 3], st[kt][qt], 0, 0, 0);
                    dpt[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][s8 >> 2][s8 & 3], gf[qt][s8 >> 2][s8 & 3], dpt[kt][qt], 0, 0, 0);
                }
        float kbv[NT][4];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) kbv[kt][e] = Kb[16 * kt + 4 * g + e];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kt][qt][e] += kbv[kt][e];
                    mx = fmaxf(mx, st[kt][qt][e]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kt][qt][e] = __expf(st[kt][qt][e] - mx);
                    sum += st[kt][qt][e];
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            float dot = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kt][qt][e] *= inv;
                    dot = fmaf(st[kt][qt][e], dpt[kt][qt][e], dot);
                }
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[kt][qt][e] *= dpt[kt][qt][e] - dot;             // dS^T
        }
        f32x4 dq[NT][2];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) dq[qt][0] = dq[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float a = Ks[min(16 * kt + 4 * g + e, L - 1) * RS + 16 * dt + c];      // keys >= L: dS^T is exactly 0
#pragma unroll
                    for (int qt = 0; qt < NT; ++qt) dq[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, st[kt][qt][e], dq[qt][dt], 0, 0, 0);
                }
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
            if (tok[qt]) {
                const long long doff = trow[qt] * 3 * C + head * 32 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x4 v = f32x4{dq[qt][dt][0] * scale, dq[qt][dt][1] * scale, dq[qt][dt][2] * scale, dq[qt][dt][3] * scale};
                    if (p.dqkv16) store4_bf16(p.dqkv16 + doff + 16 * dt, v);
                    else *(f32x4 *)(p.dqkv + doff + 16 * dt) = v;
                }
            }
    }
    // ---- row orientation: S[q][k], dP[q][k]  ->  P, dS  ->  dK, dV -----------------------------------------------------
    f32x4 sm[NT][NT], dpm[NT][NT];
    zero9(sm);
    zero9(dpm);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[qt][s8 >> 2][s8 & 3], kf[kt][s8 >> 2][s8 & 3], sm[qt][kt], 0, 0, 0);
                dpm[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[qt][s8 >> 2][s8 & 3], vf[kt][s8 >> 2][s8 & 3], dpm[qt][kt], 0, 0, 0);
            }
    float kbc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) kbc[kt] = Kb[16 * kt + c];
    // reductions over the 16 lanes of a DPP row (= the 16 key columns of a tile): quad swaps, then half-row and row mirrors --
    // every lane pairs with a lane holding the complementary partial result, so all 16 end with the same value; no LDS traffic
    auto row_max = [](float v) {
        v = fmaxf(v, attn_dpp<0xB1>(v));
        v = fmaxf(v, attn_dpp<0x4E>(v));
        v = fmaxf(v, attn_dpp<0x141>(v));
        return fmaxf(v, attn_dpp<0x140>(v));
    };
    auto row_sum = [](float v) {
        v += attn_dpp<0xB1>(v);
        v += attn_dpp<0x4E>(v);
        v += attn_dpp<0x141>(v);
        return v + attn_dpp<0x140>(v);
    };
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                  // row q = 16 qt + 4 g + e, this lane's columns k = 16 kt + c
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt][e] += kbc[kt];
                mx = fmaxf(mx, sm[qt][kt][e]);
            }
            mx = row_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt][e] = __expf(sm[qt][kt][e] - mx);
                sum += sm[qt][kt][e];
            }
            const float rsum = row_sum(sum);
            const float inv = (16 * qt + 4 * g + e < L) ? 1.0f / rsum : 0.f;                // rows past L: P = 0 (their dO image row is clamped)
            float dot = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt][e] *= inv;                                   // P
                dot = fmaf(sm[qt][kt][e], dpm[qt][kt][e], dot);
            }
            dot = row_sum(dot);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) dpm[qt][kt][e] = sm[qt][kt][e] * (dpm[qt][kt][e] - dot);     // dS
        }
    f32x4 dk[NT][2], dv[NT][2];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) dk[kt][0] = dk[kt][1] = dv[kt][0] = dv[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int qrow = min(16 * qt + 4 * g + e, L - 1);          // queries >= L: dS and P are exactly 0
                const float aq = Ql[qrow * RS + 16 * dt + c];
                const float ag = Gl[qrow * RS + 16 * dt + c];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, dpm[qt][kt][e], dk[kt][dt], 0, 0, 0);
                    dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag, sm[qt][kt][e], dv[kt][dt], 0, 0, 0);
                }
            }
    bool padded = false;
    f32x4 pk[2], pv[2];
    pk[0] = pk[1] = pv[0] = pv[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (tok[kt]) {
            const long long doff = trow[kt] * 3 * C + head * 32 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (p.dqkv16) {
                    store4_bf16(p.dqkv16 + doff + C + 16 * dt, dk[kt][dt]);
                    store4_bf16(p.dqkv16 + doff + 2 * C + 16 * dt, dv[kt][dt]);
                } else {
                    *(f32x4 *)(p.dqkv + doff + C + 16 * dt) = dk[kt][dt];
                    *(f32x4 *)(p.dqkv + doff + 2 * C + 16 * dt) = dv[kt][dt];
                }
            }
        } else if (tpad[kt]) {                    // k, v of a zero-padded token are the in-proj bias: their gradients belong to it
            padded = true;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                pk[dt] += dk[kt][dt];
                pv[dt] += dv[kt][dt];
            }
        }
    }
    if (__any(padded)) {                           // wave-uniform
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sk = row_sum(pk[dt][e]), sv = row_sum(pv[dt][e]);
                if (c == 0) {
                    p.pad_parts[gw * 64 + 16 * dt + 4 * g + e] = sk;
                    p.pad_parts[gw * 64 + 32 + 16 * dt + 4 * g + e] = sv;
                }
            }
    } else {
        p.pad_parts[gw * 64 + lane] = 0.f;
    }
}

// ---- bf16 matrix-core version (bf16 rows in and out: the bf16 training step) ------------------------------------------------------------
// The same five products in both orientations, on the bf16 matrix cores: the four d-contractions (S^T = K Q^T, dP^T = V dO^T, S = Q K^T,
// dP = dO V^T) are ONE v_mfma_f32_16x16x32_bf16 per tile -- a lane's operand is one 16-byte load of the token's bf16 row -- and the three
// token-contractions (dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P) are v_mfma_f32_16x16x16_bf16 whose B operand is the score tile's own
// registers (P, dS rounded once to bf16) and whose A operand comes from LDS images of K, Q, dO ([token][32 dims] bf16, 80-byte rows, rows
// past L zero) through the transposing read ds_read_b64_tr_b16.  The 1/sqrt(32) scale is applied to the fp32 scores and to dq, dk (the
// images hold the unscaled q).  Scores, softmax, dS in fp32 as before.  90 matrix instructions of 8-16 cycles instead of 360 of 32.
typedef short bwfrag8 __attribute__((ext_vector_type(8)));
typedef short bwfrag4 __attribute__((ext_vector_type(4)));
typedef unsigned bwu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bwu32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) bwfrag4 *lds_bwfrag4_ptr;

__device__ __forceinline__ unsigned bw_pack2(float lo, float hi)
{
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ bwu32x4 bw_bias8(const float *b)
{
    const f32x4 b0 = *(const f32x4 *)b, b1 = *(const f32x4 *)(b + 4);
    return bwu32x4{bw_pack2(b0[0], b0[1]), bw_pack2(b0[2], b0[3]), bw_pack2(b1[0], b1[1]), bw_pack2(b1[2], b1[3])};
}
__device__ __forceinline__ bwfrag4 bw_pack4(const f32x4 &v)
{
    const bwu32x2 w = {bw_pack2(v[0], v[1]), bw_pack2(v[2], v[3])};
    return __builtin_bit_cast(bwfrag4, w);
}

template <int NT>
__global__ __launch_bounds__(128, 2) void window_attention_bwd_bf16_kernel(const AttnB p)
{
    constexpr int LT = 16 * NT;
    constexpr int VS = 80;                                    // bytes per image row (64 used)
    constexpr int IMG = LT * VS;
    constexpr int WB = 3 * IMG + LT * 4;                      // K, Q, dO images, then the key bias
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int L = p.L, C = p.C;
    unsigned char *Ki = smem16 + wave * WB, *Qi = Ki + IMG, *Gi = Qi + IMG;
    float *Kb = (float *)(Gi + IMG);
    const long long gw = (long long)blockIdx.x * 2 + wave;
    const bool active = gw < p.total_waves;
    const int head = (int)(gw % p.heads);
    const long long t1 = gw / p.heads;
    const int nwin = p.global ? 1 : p.nwh * p.nww;
    const int win = (int)(t1 % nwin);
    const long long b = t1 / nwin;
    const int wr = win / p.nww, wc = win - wr * p.nww;
    const long long img = b * p.H * p.W;
    const float scale = 0.17677669529663687f;

    bwu32x4 kf[NT], qf[NT], vf[NT], gf[NT];
    long long trow[NT];
    bool tok[NT], tpad[NT];
    if (active) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + c;
            int sy = 0, sx = 0, py = 0, px = 0;
            const bool ok = j < L && tok_src(p, wr, wc, j, sy, sx, py, px);
            tok[t] = ok;
            tpad[t] = j < L && !ok;
            trow[t] = img + (long long)sy * p.W + sx;
            const bwu32x4 z{0u, 0u, 0u, 0u};
            if (ok) {
                const unsigned short *row16 = p.qkv16 + trow[t] * 3 * C + head * 32 + 8 * g;
                qf[t] = *(const bwu32x4 *)row16;
                kf[t] = *(const bwu32x4 *)(row16 + C);
                vf[t] = *(const bwu32x4 *)(row16 + 2 * C);
                gf[t] = *(const bwu32x4 *)(p.dctx16 + trow[t] * C + head * 32 + 8 * g);
            } else if (j < L) {                               // zero-padded token: q, k, v = the bias as the bf16 projection stored it; dO = 0
                qf[t] = bw_bias8(p.bias + head * 32 + 8 * g);
                kf[t] = bw_bias8(p.bias + C + head * 32 + 8 * g);
                vf[t] = bw_bias8(p.bias + 2 * C + head * 32 + 8 * g);
                gf[t] = z;
            } else {                                          // rows past L: zeros everywhere (their key bias is -inf, their P row is forced to 0)
                qf[t] = kf[t] = vf[t] = gf[t] = z;
            }
            *(bwu32x4 *)(Ki + j * VS + 16 * g) = kf[t];
            *(bwu32x4 *)(Qi + j * VS + 16 * g) = qf[t];
            *(bwu32x4 *)(Gi + j * VS + 16 * g) = gf[t];
        }
        if (lane < LT) {
            float kb = -INFINITY;
            if (lane < L) {
                int sy, sx, py, px;
                const bool ok = tok_src(p, wr, wc, lane, sy, sx, py, px);
                kb = 0.f;
                if (!p.global) {
                    if (p.shift == 0) {
                        kb = ok ? 0.f : -INFINITY;
                    } else {
                        int my = (py - 2 * p.shift) % p.Hp, mx = (px - 2 * p.shift) % p.Wp;
                        my += my < 0 ? p.Hp : 0;
                        mx += mx < 0 ? p.Wp : 0;
                        kb = (my < p.H && mx < p.W) ? __uint_as_float((unsigned)p.xf16[(img + (long long)my * p.W + mx) * C] << 16) : 0.f;
                    }
                }
            }
            Kb[lane] = kb;
        }
    }
    __syncthreads();
    if (!active) return;                                      // whole waves only: the transposing reads need a full EXEC mask

    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    const int tq = c >> 2, tp = c & 3;                         // this lane SUPPLIES row tq, columns 4 tp .. + 3 of its group's 4 x 16 block
    auto tr_frag = [&](const unsigned char *image, int t, int dt) -> bwfrag4 {      // image rows 16 t + 4 g .. + 3, dims 16 dt .. + 15, transposed
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bwfrag4_ptr)(image + (16 * t + 4 * g + tq) * VS + (16 * dt + 4 * tp) * 2));
    };
    auto mm32 = [&](const bwu32x4 &a, const bwu32x4 &bq) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bwfrag8, a), __builtin_bit_cast(bwfrag8, bq), zero4, 0, 0, 0);
    };
    // ---- column orientation: S^T[k][q], dP^T[k][q]  ->  dS^T  ->  dQ ------------------------------------------------
    {
        f32x4 st[NT][NT], dpt[NT][NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                st[kt][qt] = mm32(kf[kt], qf[qt]);
                dpt[kt][qt] = mm32(vf[kt], gf[qt]);
            }
        float kbv[NT][4];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) kbv[kt][e] = Kb[16 * kt + 4 * g + e];
        bwfrag4 dsb[NT][NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kt][qt][e] = st[kt][qt][e] * scale + kbv[kt][e];
                    mx = fmaxf(mx, st[kt][qt][e]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kt][qt][e] = __expf(st[kt][qt][e] - mx);
                    sum += st[kt][qt][e];
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            float dot = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kt][qt][e] *= inv;
                    dot = fmaf(st[kt][qt][e], dpt[kt][qt][e], dot);
                }
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) st[kt][qt][e] *= dpt[kt][qt][e] - dot;             // dS^T
                dsb[kt][qt] = bw_pack4(st[kt][qt]);
            }
        }
        f32x4 dq[NT][2];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) dq[qt][0] = dq[qt][1] = zero4;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bwfrag4 a = tr_frag(Ki, kt, dt);                                         // K^T: dims x keys
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) dq[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, dsb[kt][qt], dq[qt][dt], 0, 0, 0);
            }
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
            if (tok[qt]) {
                const long long doff = trow[qt] * 3 * C + head * 32 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    store4_bf16(p.dqkv16 + doff + 16 * dt, f32x4{dq[qt][dt][0] * scale, dq[qt][dt][1] * scale, dq[qt][dt][2] * scale, dq[qt][dt][3] * scale});
            }
    }
    // ---- row orientation: S[q][k], dP[q][k]  ->  P, dS  ->  dK, dV -----------------------------------------------------
    f32x4 sm[NT][NT], dpm[NT][NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            sm[qt][kt] = mm32(qf[qt], kf[kt]);
            dpm[qt][kt] = mm32(gf[qt], vf[kt]);
        }
    float kbc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) kbc[kt] = Kb[16 * kt + c];
    auto row_max = [](float v) {
        v = fmaxf(v, attn_dpp<0xB1>(v));
        v = fmaxf(v, attn_dpp<0x4E>(v));
        v = fmaxf(v, attn_dpp<0x141>(v));
        return fmaxf(v, attn_dpp<0x140>(v));
    };
    auto row_sum = [](float v) {
        v += attn_dpp<0xB1>(v);
        v += attn_dpp<0x4E>(v);
        v += attn_dpp<0x141>(v);
        return v + attn_dpp<0x140>(v);
    };
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                  // row q = 16 qt + 4 g + e, this lane's columns k = 16 kt + c
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt][e] = sm[qt][kt][e] * scale + kbc[kt];
                mx = fmaxf(mx, sm[qt][kt][e]);
            }
            mx = row_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt][e] = __expf(sm[qt][kt][e] - mx);
                sum += sm[qt][kt][e];
            }
            const float rsum = row_sum(sum);
            const float inv = (16 * qt + 4 * g + e < L) ? 1.0f / rsum : 0.f;                // rows past L: P = 0
            float dot = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                sm[qt][kt][e] *= inv;                                   // P
                dot = fmaf(sm[qt][kt][e], dpm[qt][kt][e], dot);
            }
            dot = row_sum(dot);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) dpm[qt][kt][e] = sm[qt][kt][e] * (dpm[qt][kt][e] - dot);     // dS
        }
    f32x4 dk[NT][2], dv[NT][2];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) dk[kt][0] = dk[kt][1] = dv[kt][0] = dv[kt][1] = zero4;
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        bwfrag4 dsr[NT], pr[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            dsr[kt] = bw_pack4(dpm[qt][kt]);
            pr[kt] = bw_pack4(sm[qt][kt]);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const bwfrag4 aq = tr_frag(Qi, qt, dt), ag = tr_frag(Gi, qt, dt);                  // Q^T, dO^T: dims x queries
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aq, dsr[kt], dk[kt][dt], 0, 0, 0);
                dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ag, pr[kt], dv[kt][dt], 0, 0, 0);
            }
        }
    }
    bool padded = false;
    f32x4 pk[2], pv[2];
    pk[0] = pk[1] = pv[0] = pv[1] = zero4;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) dk[kt][dt][e] *= scale;
        if (tok[kt]) {
            const long long doff = trow[kt] * 3 * C + head * 32 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                store4_bf16(p.dqkv16 + doff + C + 16 * dt, dk[kt][dt]);
                store4_bf16(p.dqkv16 + doff + 2 * C + 16 * dt, dv[kt][dt]);
            }
        } else if (tpad[kt]) {                    // k, v of a zero-padded token are the in-proj bias: their gradients belong to it
            padded = true;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                pk[dt] += dk[kt][dt];
                pv[dt] += dv[kt][dt];
            }
        }
    }
    if (__any(padded)) {                           // wave-uniform
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sk = row_sum(pk[dt][e]), sv = row_sum(pv[dt][e]);
                if (c == 0) {
                    p.pad_parts[gw * 64 + 16 * dt + 4 * g + e] = sk;
                    p.pad_parts[gw * 64 + 32 + 16 * dt + 4 * g + e] = sv;
                }
            }
    } else {
        p.pad_parts[gw * 64 + lane] = 0.f;
    }
}

// dbias_pad [3C] in two fixed-order steps.  Step 1 (grid heads x kPadRanges): block (h, r) adds the waves' (dk | dv) sums of head h over
// its contiguous range r of (sample, window) pairs -- sixteen sub-ranges in pair order, then the sixteen sub-sums in order -- into
// mid[(h * kPadRanges + r) * 64 + j].  Step 2 (one block per head): the ranges in order; the q third of the bias gets zeros.
constexpr int kPadRanges = 64;
__global__ __launch_bounds__(1024) void attn_pad_partial_kernel(const float *__restrict__ parts, long long pairs, int heads, float *__restrict__ mid)
{
    __shared__ float sh[16][64];
    const int h = blockIdx.x, rg = blockIdx.y, j = threadIdx.x & 63, r = threadIdx.x >> 6;
    const long long per_blk = (pairs + kPadRanges - 1) / kPadRanges;
    const long long b0 = rg * per_blk, b1 = b0 + per_blk < pairs ? b0 + per_blk : pairs;
    const long long per = (per_blk + 15) / 16;
    const long long w0 = b0 + r * per, w1 = w0 + per < b1 ? w0 + per : b1;
    float s = 0.f;
    for (long long w = w0; w < w1; ++w) s += parts[(w * heads + h) * 64 + j];
    sh[r][j] = s;
    __syncthreads();
    if (r == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][j];
        mid[((long long)h * kPadRanges + rg) * 64 + j] = t;
    }
}

__global__ __launch_bounds__(64) void attn_pad_finish_kernel(const float *__restrict__ mid, int C, float *__restrict__ dbias_pad)
{
    const int h = blockIdx.x, j = threadIdx.x;
    float t = 0.f;
    for (int rg = 0; rg < kPadRanges; ++rg) t += mid[((long long)h * kPadRanges + rg) * 64 + j];
    dbias_pad[(j < 32 ? C : 2 * C) + h * 32 + (j & 31)] = t;
    if (j < 32) dbias_pad[h * 32 + j] = 0.f;
}

}  // namespace

#define EW4_ENTRY(name, kern, ...)                                                                        \
    LDM_REQUIRE(n > 0 && n % 4 == 0, name ": n must be a positive multiple of 4");                       \
    hipLaunchKernelGGL(kern, dim3(blocks_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    LDM_CHECK_LAUNCH(name);                                                                               \
    return LDM_OK;

extern "C" int ldm_gate_fwd_f32(const float *a, const float *b, float *out, long long n, void *stream)
{
    LDM_REQUIRE(a && b && out, "ldm_gate_fwd_f32: null pointer");
    EW4_ENTRY("ldm_gate_fwd_f32", gate_fwd_kernel, (const f32x4 *)a, (const f32x4 *)b, (f32x4 *)out, n / 4)
}

extern "C" int ldm_gate_bwd_f32(const float *dh, const float *a, const float *b, float *da, float *db, long long n, void *stream)
{
    LDM_REQUIRE(dh && a && b && da && db, "ldm_gate_bwd_f32: null pointer");
    EW4_ENTRY("ldm_gate_bwd_f32", gate_bwd_kernel, (const f32x4 *)dh, (const f32x4 *)a, (const f32x4 *)b, (f32x4 *)da, (f32x4 *)db, n / 4)
}

extern "C" int ldm_relu_bwd_f32(const float *dy, const float *y, float *dx, long long n, void *stream)
{
    LDM_REQUIRE(dy && y && dx, "ldm_relu_bwd_f32: null pointer");
    EW4_ENTRY("ldm_relu_bwd_f32", relu_bwd_kernel, (const f32x4 *)dy, (const f32x4 *)y, (f32x4 *)dx, n / 4)
}

extern "C" int ldm_add_f32(float *y, const float *x, long long n, void *stream)
{
    LDM_REQUIRE(y && x, "ldm_add_f32: null pointer");
    EW4_ENTRY("ldm_add_f32", add_kernel, (f32x4 *)y, (const f32x4 *)x, n / 4)
}

// out[i] = sum over planes s < S of parts[s * n + i] (+ out[i] when accumulate), in plane order: the fixed-order second step of every
// reduction that used to end in float atomics.  `parts` may hold one spare plane behind the S used ones (accumulate copies out there).
static int reduce_planes(float *parts, int S, long long n, float *out, int accumulate, hipStream_t st, const char *who)
{
    if (accumulate) {
        if (hipMemcpyAsync(parts + (long long)S * n, out, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            ldm_set_error("%s: copy of the running sum failed", who);
            return LDM_ELAUNCH;
        }
        ++S;
    }
    if (n % 4 == 0 && S >= 4 && ldm_aligned16(parts) && ldm_aligned16(out))
        hipLaunchKernelGGL(reduce_partials_v4_kernel, dim3(blocks_for(n / 4, 64)), dim3(256), 0, st, (const f32x4 *)parts, (f32x4 *)out, S, n / 4);
    else
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const float *)parts, out, S, n);
    return LDM_OK;
}

extern "C" int ldm_colsum_f32(const float *x, float *out, long long M, int N, int accumulate, void *stream)
{
    LDM_REQUIRE(x && out && M > 0 && N > 0, "ldm_colsum_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    // slabs of >= 64 rows, ~2048 of them for a long matrix: with N <= 256 columns a slab is ONE workgroup, so the slab count is the kernel's
    // whole parallelism (256 slabs of 2048 rows ran the stem's bias gradient -- 524 288 x 128 -- at 0.27 TB/s); the fixed-order sum of a
    // few thousand N-float planes is cheap
    long long slab = (M + 2047) / 2048;
    slab = slab < 64 ? 64 : slab;
    const unsigned slabs = blocks_for(M, (int)slab);
    if (slabs == 1 && !accumulate) {
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, 1), dim3(256), 0, st, x, out, M, N, (int)slab);
    } else {
        float *parts = (float *)ldm_scratch(st, (size_t)(slabs + 1) * N * sizeof(float));
        if (!parts) return LDM_ELAUNCH;
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, slabs), dim3(256), 0, st, x, parts, M, N, (int)slab);
        if (reduce_planes(parts, (int)slabs, N, out, accumulate, st, "ldm_colsum_f32") != LDM_OK) return LDM_ELAUNCH;
    }
    LDM_CHECK_LAUNCH("ldm_colsum_f32");
    return LDM_OK;
}

extern "C" int ldm_transpose_colsum_f32(const float *x, float *out, float *csum, long long R, int Cc, void *stream)
{
    LDM_REQUIRE(x && out && csum && R > 0 && Cc > 0, "ldm_transpose_colsum_f32: bad arguments");
    LDM_REQUIRE((R + 31) / 32 <= 0x7fffffffLL, "ldm_transpose_colsum_f32: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const unsigned rt = (unsigned)((R + 31) / 32);
    float *parts = (float *)ldm_scratch(st, (size_t)rt * Cc * sizeof(float));
    if (!parts) return LDM_ELAUNCH;
    hipLaunchKernelGGL(transpose_colsum_kernel, dim3((Cc + 31) / 32, rt), dim3(256), 0, st, x, out, parts, R, Cc);
    if (reduce_planes(parts, (int)rt, Cc, csum, 0, st, "ldm_transpose_colsum_f32") != LDM_OK) return LDM_ELAUNCH;
    LDM_CHECK_LAUNCH("ldm_transpose_colsum_f32");
    return LDM_OK;
}

extern "C" int ldm_reduce_partials_f32(const float *parts, float *out, int S, long long n, void *stream)
{
    LDM_REQUIRE(parts && out && S > 0 && n > 0, "ldm_reduce_partials_f32: bad arguments");
    if (n % 4 == 0 && ldm_aligned16(parts) && ldm_aligned16(out))
        hipLaunchKernelGGL(reduce_partials_v4_kernel, dim3(blocks_for(n / 4, 64)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)parts,
                           (f32x4 *)out, S, n / 4);
    else
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, parts, out, S, n);
    LDM_CHECK_LAUNCH("ldm_reduce_partials_f32");
    return LDM_OK;
}

extern "C" int ldm_reduce_partials_pair_f32(const float *parts_a, float *out_a, long long n_a, const float *parts_b, float *out_b, long long n_b,
                                           int S, long long row_len_a, long long seg_len_a, void *stream)
{
    LDM_REQUIRE(parts_a && out_a && parts_b && out_b && S > 0 && n_a > 0 && n_b > 0, "ldm_reduce_partials_pair_f32: bad arguments");
    LDM_REQUIRE(seg_len_a == 0 || (seg_len_a > 0 && seg_len_a % 4 == 0 && row_len_a > 0 && row_len_a % seg_len_a == 0 && n_a % row_len_a == 0 && parts_a != out_a),
                "ldm_reduce_partials_pair_f32: column segments must be multiples of 4 that divide the row, the row must divide n_a, no in-place");
    LDM_REQUIRE(n_a % 4 == 0 && n_b % 4 == 0, "ldm_reduce_partials_pair_f32: element counts must be multiples of 4");
    LDM_REQUIRE(ldm_aligned16(parts_a) && ldm_aligned16(out_a) && ldm_aligned16(parts_b) && ldm_aligned16(out_b), "ldm_reduce_partials_pair_f32: unaligned pointer");
    const unsigned ba = blocks_for(n_a / 4, 64), bb = blocks_for(n_b / 4, 64);
    hipLaunchKernelGGL(reduce_partials_pair_kernel, dim3(ba + bb), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)parts_a, (f32x4 *)out_a, n_a / 4, ba,
                       (const f32x4 *)parts_b, (f32x4 *)out_b, n_b / 4, S, seg_len_a ? row_len_a / 4 : 0, seg_len_a / 4);
    LDM_CHECK_LAUNCH("ldm_reduce_partials_pair_f32");
    return LDM_OK;
}

extern "C" int ldm_channelnorm_film_bwd_f32(const float *x, const float *film, const int *slot, const float *dxf, const float *dres,
                                            float *dx, float *dfilm, int B, int HW, int C, float eps, int unique_slots, void *stream)
{
    LDM_REQUIRE(x && film && dxf && dx && dfilm, "ldm_channelnorm_film_bwd_f32: null pointer");
    LDM_REQUIRE(B > 0 && HW > 0 && C >= 8 && C % 4 == 0 && C <= 64 * 4 * kMaxV, "ldm_channelnorm_film_bwd_f32: bad shape");
    const int lpr = pow2_lanes(C / 4);
    const long long rows = (long long)B * HW;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    hipLaunchKernelGGL(channelnorm_film_bwd_kernel, dim3(blocks_for(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, film, slot, dxf, dres,
                       dx, dfilm, rows, HW, C, eps, lpr, unique_slots);
    LDM_CHECK_LAUNCH("ldm_channelnorm_film_bwd_f32");
    return LDM_OK;
}

extern "C" int ldm_avgpool2_bwd_f32(const float *dlo, float *dx, int B, int H, int W, int C, int accumulate, void *stream)
{
    LDM_REQUIRE(dlo && dx && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "ldm_avgpool2_bwd_f32: bad arguments");
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)dlo, (f32x4 *)dx, B,
                       H / 2, W / 2, C / 4, accumulate);
    LDM_CHECK_LAUNCH("ldm_avgpool2_bwd_f32");
    return LDM_OK;
}

extern "C" int ldm_sumpool2_f32(const float *dhi, float *dlo, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(dhi && dlo && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "ldm_sumpool2_f32: bad arguments");
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(sumpool2_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)dhi, (f32x4 *)dlo, B, H / 2,
                       W / 2, C / 4);
    LDM_CHECK_LAUNCH("ldm_sumpool2_f32");
    return LDM_OK;
}

extern "C" int ldm_stem_bwd_f32(const float *x, const float *dy, float *dw, int B, int Cin, int HW, int C0, void *stream)
{
    LDM_REQUIRE(x && dy && dw && B > 0 && Cin > 0 && HW > 0 && C0 > 0, "ldm_stem_bwd_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)B * HW;
    const long long nw = (long long)C0 * Cin;
    // every block writes its partial sums to ITS plane of the scratch; the planes are added in block order (bit-reproducible)
    if (C0 <= kRowsNT && Cin <= 16) {           // (the Decoder's 512-wide input layer too: the generic kernel below took 1.4 ms for its 8 k rows)
        long long per = ((M + 255) / 256 + kRowsNT - 1) / kRowsNT * kRowsNT;     // rows per block: ~256 blocks, whole 1024-row chunks
        if (per > 0x40000000LL) per = 0x40000000LL;
        const int slab = (int)per;
        const dim3 grid(blocks_for(M, slab));
        float *parts = (float *)ldm_scratch(st, (size_t)grid.x * nw * sizeof(float));
        if (!parts) return LDM_ELAUNCH;
        if (Cin <= 4) hipLaunchKernelGGL(stem_bwd_rows_kernel<4>, grid, dim3(kRowsNT), 0, st, x, dy, parts, M, Cin, HW, C0, slab);
        else if (Cin <= 8) hipLaunchKernelGGL(stem_bwd_rows_kernel<8>, grid, dim3(kRowsNT), 0, st, x, dy, parts, M, Cin, HW, C0, slab);
        else hipLaunchKernelGGL(stem_bwd_rows_kernel<16>, grid, dim3(kRowsNT), 0, st, x, dy, parts, M, Cin, HW, C0, slab);
        if (reduce_planes(parts, (int)grid.x, nw, dw, 0, st, "ldm_stem_bwd_f32") != LDM_OK) return LDM_ELAUNCH;
    } else {
        const int slab = (int)((M + 511) / 512 < 256 ? 256 : (M + 511) / 512);      // at most ~512 planes
        const unsigned nb = blocks_for(M, slab);
        float *parts = (float *)ldm_scratch(st, (size_t)nb * nw * sizeof(float));
        if (!parts) return LDM_ELAUNCH;
        hipLaunchKernelGGL(stem_bwd_kernel, dim3(nb), dim3(256), 0, st, x, dy, parts, M, Cin, HW, C0, slab);
        if (reduce_planes(parts, (int)nb, nw, dw, 0, st, "ldm_stem_bwd_f32") != LDM_OK) return LDM_ELAUNCH;
    }
    LDM_CHECK_LAUNCH("ldm_stem_bwd_f32");
    return LDM_OK;
}

extern "C" int ldm_head_bwd_f32(const float *x, const float *w, const float *dout, float *dx, float *dw, float *db, int B, int C0, int HW,
                                int Cin, void *stream)
{
    LDM_REQUIRE(x && w && dout && dx && dw && db, "ldm_head_bwd_f32: null pointer");
    LDM_REQUIRE(B > 0 && C0 > 0 && HW > 0 && Cin > 0 && Cin <= 16, "ldm_head_bwd_f32: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)B * HW;
    const long long plane = (long long)C0 * Cin + Cin;                 // [weight sums | bias sums] per block, added in block order afterwards
    unsigned nb;
    float *parts;
    if (C0 <= kRowsNT) {
        const long long ntiles = (M + 255) / 256;
        nb = (unsigned)(ntiles < 256 ? ntiles : 256);
        parts = (float *)ldm_scratch(st, (size_t)(nb + 1) * plane * sizeof(float));
        if (!parts) return LDM_ELAUNCH;
        const dim3 grid(nb);
        if (Cin <= 4) hipLaunchKernelGGL(head_bwd_rows_kernel<4>, grid, dim3(kRowsNT), 0, st, x, w, dout, dx, parts, parts, M, C0, HW, Cin, ntiles);
        else if (Cin <= 8) hipLaunchKernelGGL(head_bwd_rows_kernel<8>, grid, dim3(kRowsNT), 0, st, x, w, dout, dx, parts, parts, M, C0, HW, Cin, ntiles);
        else hipLaunchKernelGGL(head_bwd_rows_kernel<16>, grid, dim3(kRowsNT), 0, st, x, w, dout, dx, parts, parts, M, C0, HW, Cin, ntiles);
    } else {
        nb = blocks_for(M, 64);
        parts = (float *)ldm_scratch(st, (size_t)(nb + 1) * plane * sizeof(float));
        if (!parts) return LDM_ELAUNCH;
        hipLaunchKernelGGL(head_bwd_kernel, dim3(nb), dim3(256), 0, st, x, w, dout, dx, parts, parts, M, C0, HW, Cin);
    }
    // one fixed-order sum over the planes into a contiguous [C0 * Cin + Cin] vector (the first spare cells behind the planes), then two copies
    float *sum = parts + (long long)nb * plane;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks_for(plane, 256)), dim3(256), 0, st, (const float *)parts, sum, (int)nb, plane);
    if (hipMemcpyAsync(dw, sum, (size_t)C0 * Cin * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(db, sum + (long long)C0 * Cin, (size_t)Cin * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        ldm_set_error("ldm_head_bwd_f32: copy of the summed gradients failed");
        return LDM_ELAUNCH;
    }
    LDM_CHECK_LAUNCH("ldm_head_bwd_f32");
    return LDM_OK;
}

extern "C" int ldm_l1_loss_f32(const float *pred, const float *target, long long n, float *loss, void *stream)
{
    LDM_REQUIRE(pred && target && loss && n > 0, "ldm_l1_loss_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    unsigned blocks = blocks_for(n, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    float *parts = (float *)ldm_scratch(st, blocks * sizeof(float));
    if (!parts) return LDM_ELAUNCH;
    hipLaunchKernelGGL(l1_loss_kernel, dim3(blocks), dim3(256), 0, st, pred, target, n, parts);
    hipLaunchKernelGGL(l1_loss_finish_kernel, dim3(1), dim3(256), 0, st, (const float *)parts, (int)blocks, 1.0f / (float)n, loss);
    LDM_CHECK_LAUNCH("ldm_l1_loss_f32");
    return LDM_OK;
}

extern "C" int ldm_l1_loss_bwd_f32(const float *pred, const float *target, const float *gscale, float *grad, long long n, void *stream)
{
    LDM_REQUIRE(pred && target && gscale && grad && n > 0, "ldm_l1_loss_bwd_f32: bad arguments");
    hipLaunchKernelGGL(l1_loss_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, gscale, 1.0f / (float)n, grad, n);
    LDM_CHECK_LAUNCH("ldm_l1_loss_bwd_f32");
    return LDM_OK;
}

extern "C" int ldm_im2col3x3_t_f32(const float *x, float *out, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0, "ldm_im2col3x3_t_f32: bad arguments");
    const long long M = (long long)B * H * W;
    LDM_REQUIRE(C / 32 <= 65535, "ldm_im2col3x3_t_f32: too many groups");
    hipLaunchKernelGGL(im2col3x3_t_kernel, dim3(blocks_for(M, 32), 9, C / 32), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C);
    LDM_CHECK_LAUNCH("ldm_im2col3x3_t_f32");
    return LDM_OK;
}

static int g_attn_bwd_mfma = 1;                       // 0: the scalar kernel (A/B tests)
extern "C" int ldm_window_attention_bwd_mfma(int v)
{
    const int old = g_attn_bwd_mfma;
    if (v == 0 || v == 1) g_attn_bwd_mfma = v;
    return old;
}

int g_attn_bwd_bf16_core = 1;        // ldm_window_attention_bwd_bf16: 1 = bf16 matrix cores (default), 0 = the fp32 16x16x4 core (A/B tests)

extern "C" int ldm_window_attention_bwd_bf16_core(int v)
{
    const int old = g_attn_bwd_bf16_core;
    if (v == 0 || v == 1) g_attn_bwd_bf16_core = v;
    return old;
}

static int attention_bwd_impl(const char *who, const void *qkv, const float *in_proj_bias, const void *xf, const void *dctx, void *dqkv, float *dbias_pad,
                              int B, int H, int W, int C, int ws, int shift, bool io16, void *stream)
{
    LDM_REQUIRE(qkv && in_proj_bias && dctx && dqkv && dbias_pad, "%s: null pointer", who);
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0 && ws >= 1 && ws <= 6 && shift >= 0 && shift < ws, "%s: bad shape (window_size <= 6)", who);
    AttnB p{};
    p.bias = in_proj_bias; p.dbias_pad = dbias_pad;
    if (io16) {
        p.qkv16 = (const unsigned short *)qkv; p.dctx16 = (const unsigned short *)dctx; p.xf16 = (const unsigned short *)xf; p.dqkv16 = (unsigned short *)dqkv;
        LDM_REQUIRE(ldm_aligned16(qkv) && ldm_aligned16(dctx) && (((size_t)dqkv) & 7) == 0, "%s: unaligned bf16 rows", who);
    } else {
        p.qkv = (const float *)qkv; p.dctx = (const float *)dctx; p.xf = (const float *)xf; p.dqkv = (float *)dqkv;
    }
    p.B = B; p.H = H; p.W = W; p.C = C; p.ws = ws; p.shift = shift; p.heads = C / 32;
    p.global = (H <= ws && W <= ws) ? 1 : 0;
    if (p.global) {
        p.Hp = H; p.Wp = W; p.nwh = p.nww = 1; p.L = H * W; p.shift = 0;
    } else {
        p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
        p.nwh = p.Hp / ws; p.nww = p.Wp / ws; p.L = ws * ws;
        LDM_REQUIRE(shift == 0 || xf != nullptr, "%s: shift != 0 needs xf", who);
    }
    p.total_waves = (long long)B * p.nwh * p.nww * p.heads;
    hipStream_t st = (hipStream_t)stream;
    p.pad_parts = (float *)ldm_scratch(st, ((size_t)p.total_waves + (size_t)p.heads * kPadRanges) * 64 * sizeof(float));
    if (!p.pad_parts) return LDM_ELAUNCH;
    float *pad_mid = p.pad_parts + (size_t)p.total_waves * 64;
    const dim3 grid((unsigned)((p.total_waves + 1) / 2));
    if (io16 && g_attn_bwd_bf16_core && p.L <= 48 && (p.global || p.shift == 0 || xf)) {      // bf16 rows in and out: the bf16 matrix cores
        const int nt = (p.L + 15) / 16;
        const size_t smem = 2ull * (3 * 16 * nt * 80 + 16 * nt * 4);
        if (nt == 1) hipLaunchKernelGGL(window_attention_bwd_bf16_kernel<1>, grid, dim3(128), smem, st, p);
        else if (nt == 2) hipLaunchKernelGGL(window_attention_bwd_bf16_kernel<2>, grid, dim3(128), smem, st, p);
        else hipLaunchKernelGGL(window_attention_bwd_bf16_kernel<3>, grid, dim3(128), smem, st, p);
    } else if (g_attn_bwd_mfma || io16) {               // every window the reference builds has L <= 36
        const int nt = (p.L + 15) / 16;
        const size_t smem = 2ull * (3 * p.L * 36 + 16 * nt) * sizeof(float);
        if (nt == 1) hipLaunchKernelGGL(window_attention_bwd_mfma_kernel<1>, grid, dim3(128), smem, st, p);
        else if (nt == 2) hipLaunchKernelGGL(window_attention_bwd_mfma_kernel<2>, grid, dim3(128), smem, st, p);
        else hipLaunchKernelGGL(window_attention_bwd_mfma_kernel<3>, grid, dim3(128), smem, st, p);
    } else {
        constexpr int LMAX = 36;
        const size_t smem = 2ull * (2 * LMAX * 36 + 2 * LMAX * (LMAX + 1) + LMAX + 4) * sizeof(float);
        hipLaunchKernelGGL(window_attention_bwd_kernel<LMAX>, grid, dim3(128), smem, st, p);
    }
    hipLaunchKernelGGL(attn_pad_partial_kernel, dim3(p.heads, kPadRanges), dim3(1024), 0, st, (const float *)p.pad_parts, p.total_waves / p.heads, p.heads, pad_mid);
    hipLaunchKernelGGL(attn_pad_finish_kernel, dim3(p.heads), dim3(64), 0, st, (const float *)pad_mid, C, dbias_pad);
    LDM_CHECK_LAUNCH(who);
    return LDM_OK;
}

extern "C" int ldm_window_attention_bwd_f32(const float *qkv, const float *in_proj_bias, const float *xf, const float *dctx, float *dqkv,
                                            float *dbias_pad, int B, int H, int W, int C, int ws, int shift, void *stream)
{
    return attention_bwd_impl("ldm_window_attention_bwd_f32", qkv, in_proj_bias, xf, dctx, dqkv, dbias_pad, B, H, W, C, ws, shift, false, stream);
}

extern "C" int ldm_window_attention_bwd_bf16(const void *qkv, const float *in_proj_bias, const void *xf, const void *dctx, void *dqkv, float *dbias_pad,
                                             int B, int H, int W, int C, int ws, int shift, void *stream)
{
    return attention_bwd_impl("ldm_window_attention_bwd_bf16", qkv, in_proj_bias, xf, dctx, dqkv, dbias_pad, B, H, W, C, ws, shift, true, stream);
}
