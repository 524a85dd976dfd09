// HBM-bound helpers of the bf16 ("autocast") sampling / decode mode: ddpm.py:52,75 -- on a GPU the reference samples under 16-bit
// autocast (sample_ldm.py:17,72); here that mode keeps the residual stream of the UNet in fp32 and feeds every GEMM bf16 operands
// (fp32 accumulate), and the VAE decoder keeps its activations as bf16 rows.  fp16 overflows on these weights, so bf16
// is the 16-bit type.
//
//   ldm_stem_nchw_bf16         NCHW fp32 -> channels-last bf16 rows through the 1x1 input layer (vae.py:112,123)
//   ldm_depth_to_space2_bf16   [M, (dy, dx, co)] -> fine rows [4 M, co]: the scatter of ConvTranspose2d(k = 2, s = 2) (vae.py:120)
//   ldm_rgb_head_bf16          to_rgb (C -> 3) of bf16 rows + bilinear x2 accumulation of the previous stage's RGB planes (vae.py:104,131)
//   ldm_up2_add_bf16           fine[m] = coarse[parent(m)] + skip[m] -- nearest x2 of the 1x1 conv's output + the UNet skip (unet.py:85,101)
//   ldm_avgpool2_bf16          2x2 average of fp32 rows, rounded once to bf16 (the 1x1 down conv's operand; unet.py:83)
#include "common.h"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));     // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}
__device__ __forceinline__ float lo16(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

inline unsigned blocks_of(long long items, int per_block) { return (unsigned)((items + per_block - 1) / per_block); }

// W^T in LDS ([Cin][C0] fp32); a thread owns 4 consecutive output channels of a pixel; x values are wave-broadcast loads
__global__ __launch_bounds__(256) void stem_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                        unsigned short *__restrict__ out, long long M, int Cin, int HW, int C0, int pix_per_block)
{
    extern __shared__ __attribute__((aligned(16))) float wt[];
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
        *(u32x2 *)(out + m * C0 + 4 * n4) = u32x2{pack2(acc[0], acc[1]), pack2(acc[2], acc[3])};
    }
}

// one 16-byte chunk (8 channels) per thread: coarse row m = (b, y, x), column q * C + c  ->  fine row (b, 2 y + (q >> 1), 2 x + (q & 1)), column c
__global__ __launch_bounds__(256) void depth_to_space2_bf16_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, long long M, int H, int W, int c8n)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = M * 4 * c8n;
    if (idx >= total) return;
    const int c8 = (int)(idx % c8n);
    long long r = idx / c8n;
    const int q = (int)(r & 3);
    const long long m = r >> 2;
    const int xx = (int)(m % W);
    const long long r2 = m / W;
    const int yy = (int)(r2 % H);
    const long long b = r2 / H;
    const long long fine = (b * 2 * H + 2 * yy + (q >> 1)) * (2 * W) + 2 * xx + (q & 1);
    out[fine * c8n + c8] = in[idx];
}

__device__ __forceinline__ float group_sum16(float v, int lpr)
{
    for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// a group of `lpr` lanes owns one pixel; a lane reads 8 channels (16 bytes) per step
template <int OC>
__global__ __launch_bounds__(256) void rgb_head_bf16_kernel(const unsigned short *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                            const float *__restrict__ prev, float *__restrict__ out, int B, int H, int W, int C, int lpr)
{
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;
    const long long rows = (long long)B * H * W;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long row = wave * rpw + lane / lpr;
    const int sub = lane % lpr;
    const int c8n = C >> 3;
    const bool live = row < rows;
    const u32x4 *xr = (const u32x4 *)(x + (live ? row : 0) * C);
    float d[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) d[j] = 0.f;
    for (int c8 = sub; c8 < c8n; c8 += lpr) {
        const u32x4 v = live ? xr[c8] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x4 wj[OC];
#pragma unroll
            for (int j = 0; j < OC; ++j) wj[j] = ((const f32x4 *)(w + j * C))[2 * c8 + hh];
            const float xv[4] = {lo16(v[2 * hh]), hi16(v[2 * hh]), lo16(v[2 * hh + 1]), hi16(v[2 * hh + 1])};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < OC; ++j) d[j] = fmaf(xv[e], wj[j][e], d[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < OC; ++j) d[j] = group_sum16(d[j], lpr);
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

// fine[(b, y, x), :] = coarse[(b, y / 2, x / 2), :] + skip[(b, y, x), :]; one float4 per thread
__global__ __launch_bounds__(256) void up2_add_kernel(const f32x4 *__restrict__ coarse, const f32x4 *__restrict__ skip, f32x4 *__restrict__ out, long long Mf, int H2,
                                                      int W2, int c4n)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mf * c4n) return;
    const int c4 = (int)(idx % c4n);
    const long long m = idx / c4n;
    const int xx = (int)(m % W2);
    const long long r2 = m / W2;
    const int yy = (int)(r2 % H2);
    const long long b = r2 / H2;
    const long long mc = (b * (H2 >> 1) + (yy >> 1)) * (W2 >> 1) + (xx >> 1);
    f32x4 v = coarse[mc * c4n + c4];
    if (skip) v += skip[idx];
    out[idx] = v;
}

// 2x2 average of fp32 rows -> bf16 rows; 4 channels per thread, same association as ldm_avgpool2_f32
__global__ __launch_bounds__(256) void avgpool2_bf16_kernel(const f32x4 *__restrict__ x, u32x2 *__restrict__ out, int B, int H, int W, int c4n)
{
    const int OH = H >> 1, OW = W >> 1;
    const long long total = (long long)B * OH * OW * c4n;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const long long b = r / OH;
    const f32x4 *p = x + ((b * H + 2 * oy) * W + 2 * ox) * c4n + c4;
    const f32x4 a = p[0], bq = p[c4n], c = p[(long long)W * c4n], d = p[(long long)W * c4n + c4n];
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (((a[e] + bq[e]) + c[e]) + d[e]) * 0.25f;
    out[idx] = u32x2{pack2(o[0], o[1]), pack2(o[2], o[3])};
}

int pow2_floor(int v)
{
    int p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}

}  // namespace

extern "C" int ldm_stem_nchw_bf16(const float *x, const float *w, const float *bias, void *out_bf16, int B, int Cin, int HW, int C0, void *stream)
{
    LDM_REQUIRE(x && w && out_bf16, "ldm_stem_nchw_bf16: null pointer");
    LDM_REQUIRE(B > 0 && Cin > 0 && HW > 0 && C0 >= 4 && C0 % 4 == 0, "ldm_stem_nchw_bf16: bad shape (C0 %% 4 == 0)");
    LDM_REQUIRE((size_t)Cin * C0 * sizeof(float) <= 64 * 1024, "ldm_stem_nchw_bf16: Cin*C0 too large for the LDS weight tile");
    LDM_REQUIRE((((size_t)out_bf16) & 7) == 0 && (!bias || ldm_aligned16(bias)), "ldm_stem_nchw_bf16: unaligned pointer");
    const long long M = (long long)B * HW;
    const int ppb = 64;
    hipLaunchKernelGGL(stem_bf16_kernel, dim3(blocks_of(M, ppb)), dim3(256), (size_t)Cin * C0 * sizeof(float), (hipStream_t)stream, x, w, bias,
                       (unsigned short *)out_bf16, M, Cin, HW, C0, ppb);
    LDM_CHECK_LAUNCH("ldm_stem_nchw_bf16");
    return LDM_OK;
}

extern "C" int ldm_depth_to_space2_bf16(const void *in_bf16, void *out_bf16, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(in_bf16 && out_bf16 && in_bf16 != out_bf16, "ldm_depth_to_space2_bf16: null / aliased pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "ldm_depth_to_space2_bf16: bad shape (C %% 8 == 0)");
    LDM_REQUIRE(ldm_aligned16(in_bf16) && ldm_aligned16(out_bf16), "ldm_depth_to_space2_bf16: unaligned pointer");
    const long long M = (long long)B * H * W;
    hipLaunchKernelGGL(depth_to_space2_bf16_kernel, dim3(blocks_of(M * 4 * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)in_bf16,
                       (u32x4 *)out_bf16, M, H, W, C / 8);
    LDM_CHECK_LAUNCH("ldm_depth_to_space2_bf16");
    return LDM_OK;
}

extern "C" int ldm_rgb_head_bf16(const void *x_bf16, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, void *stream)
{
    return ldm_rgb_head_oc_bf16(x_bf16, w, bias, prev, out, B, H, W, C, 3, stream);
}

extern "C" int ldm_rgb_head_oc_bf16(const void *x_bf16, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, int OC,
                                    void *stream)
{
    LDM_REQUIRE(x_bf16 && w && bias && out, "ldm_rgb_head_bf16: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && OC >= 1 && OC <= 4, "ldm_rgb_head_bf16: bad shape (C %% 8 == 0, 1 <= output channels <= 4)");
    LDM_REQUIRE(!prev || (H % 2 == 0 && W % 2 == 0), "ldm_rgb_head_bf16: prev needs even H, W");
    LDM_REQUIRE(ldm_aligned16(x_bf16) && ldm_aligned16(w) && (C % 4 == 0), "ldm_rgb_head_bf16: unaligned pointer");
    int lpr = pow2_floor(C / 8);
    lpr = lpr > 8 ? 8 : lpr;
    const long long rows = (long long)B * H * W;
    const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    const dim3 grid(blocks_of(waves, 4));
    const unsigned short *x16 = (const unsigned short *)x_bf16;
    if (OC == 3) hipLaunchKernelGGL(rgb_head_bf16_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x16, w, bias, prev, out, B, H, W, C, lpr);
    else if (OC == 1) hipLaunchKernelGGL(rgb_head_bf16_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x16, w, bias, prev, out, B, H, W, C, lpr);
    else if (OC == 2) hipLaunchKernelGGL(rgb_head_bf16_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x16, w, bias, prev, out, B, H, W, C, lpr);
    else hipLaunchKernelGGL(rgb_head_bf16_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x16, w, bias, prev, out, B, H, W, C, lpr);
    LDM_CHECK_LAUNCH("ldm_rgb_head_bf16");
    return LDM_OK;
}

extern "C" int ldm_up2_add_f32(const float *coarse, const float *skip, float *out, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(coarse && out, "ldm_up2_add_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "ldm_up2_add_f32: bad shape (coarse H, W; C %% 4 == 0)");
    LDM_REQUIRE(ldm_aligned16(coarse) && ldm_aligned16(out) && (!skip || ldm_aligned16(skip)), "ldm_up2_add_f32: unaligned pointer");
    const long long Mf = (long long)B * 4 * H * W;
    hipLaunchKernelGGL(up2_add_kernel, dim3(blocks_of(Mf * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)coarse, (const f32x4 *)skip, (f32x4 *)out,
                       Mf, 2 * H, 2 * W, C / 4);
    LDM_CHECK_LAUNCH("ldm_up2_add_f32");
    return LDM_OK;
}

extern "C" int ldm_avgpool2_bf16(const float *x, void *out_bf16, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(x && out_bf16, "ldm_avgpool2_bf16: null pointer");
    LDM_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C >= 4 && C % 4 == 0, "ldm_avgpool2_bf16: bad shape");
    LDM_REQUIRE(ldm_aligned16(x) && (((size_t)out_bf16) & 7) == 0, "ldm_avgpool2_bf16: unaligned pointer");
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_bf16_kernel, dim3(blocks_of(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (u32x2 *)out_bf16, B, H, W, C / 4);
    LDM_CHECK_LAUNCH("ldm_avgpool2_bf16");
    return LDM_OK;
}
