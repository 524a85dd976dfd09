// Backward pieces of the VAE Decoder (vae.py:54-66,99-132) that are not GEMMs: leaky-ReLU gradient, row-major im2col for the dense 3x3
// weight gradient, space-to-depth for the ConvTranspose 2x2 gradients, and the to_rgb + bilinear x2 head.  The GEMM-shaped
// gradients reuse ldm_gemm_f32 (data gradients: the implicit 3x3 GEMM on flipped / transposed filters) and ldm_gemm_tn_f32.
#include "common.h"

namespace {

inline unsigned vb_blocks(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// dx = dy * (y > 0 ? 1 : slope); y is the ACTIVATED output (slope > 0: its sign is the pre-activation's)
__global__ void lrelu_bwd_kernel(const f32x4 *__restrict__ dy, const f32x4 *__restrict__ y, f32x4 *__restrict__ dx, long long n4, float slope)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 g = dy[i], v = y[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? g[e] : g[e] * slope;
    dx[i] = o;
}

// out[p][tap * C + c] = x[p + tap][c] (zero outside the image): the A operand of dW = dY^T . im2col(X)
__global__ void im2col3x3_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ out, int B, int H, int W, int C4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece of one (pixel, tap)
    const long long total = (long long)B * H * W * 9 * C4;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const long long r = i / C4;
    const int tap = (int)(r % 9);
    const long long p = r / 9;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const bool ok = (unsigned)(yy + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W;
    out[i] = ok ? x[(p + (long long)dy * W + dx) * C4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
}

// out[(b, y, x)][(dy * 2 + dx) * C + c] = fine[(b, 2 y + dy, 2 x + dx)][c]   (H, W: the COARSE size)
__global__ void space_to_depth2_kernel(const f32x4 *__restrict__ fine, f32x4 *__restrict__ out, int B, int H, int W, int C4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * 4 * C4;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const long long r = i / C4;
    const int q = (int)(r & 3);
    const long long p = r >> 2;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    const long long fp = (b * 2 * H + 2 * yy + (q >> 1)) * 2 * W + 2 * xx + (q & 1);
    out[i] = fine[fp * C4 + c4];
}

// rgb head backward, per pixel: drows[p][c] (+)= sum_j drgb[b][j][pix] w[j][c]
template <int OC>
__global__ __launch_bounds__(256) void rgb_head_bwd_kernel(const float *__restrict__ drgb, const float *__restrict__ w, float *__restrict__ drows,
                                                           int B, int H, int W, int C, int accumulate)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (long long)B * H * W) return;
    const int HW = H * W;
    const long long b = row / HW;
    const int pix = (int)(row - b * HW);
    float gj[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) gj[j] = drgb[(b * OC + j) * HW + pix];
    f32x4 *dr = (f32x4 *)(drows + row * C);
    for (int c4 = lane; c4 < (C >> 2); c4 += 64) {
        f32x4 wj[OC];
#pragma unroll
        for (int j = 0; j < OC; ++j) wj[j] = ((const f32x4 *)(w + j * C))[c4];
        f32x4 o = accumulate ? dr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = gj[0] * wj[0][e];                               // ((g0 w0 + g1 w1) + g2 w2) + ...
#pragma unroll
            for (int j = 1; j < OC; ++j) t = t + gj[j] * wj[j][e];
            o[e] += t;
        }
        dr[c4] = o;
    }
}

// The adjoint of the bilinear x2 accumulation (F.interpolate(scale_factor=2, mode='bilinear', align_corners=False), vae.py:131) as a
// GATHER: a thread owns one coarse cell (b, j, py, px) and walks the fine pixels that can touch it (rows 2 py - 2 ... 2 py + 3, same
// for columns) in a fixed order, adding the corner weights of those whose source cell it is.  No atomics: bit-reproducible.
__global__ __launch_bounds__(256) void bilinear2x_adjoint_kernel(const float *__restrict__ drgb, float *__restrict__ dprev, int B, int H, int W, int OC)
{
    const int PH = H >> 1, PW = W >> 1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * OC * PH * PW) return;
    const int px = (int)(i % PW);
    const int py = (int)((i / PW) % PH);
    const long long bj = i / ((long long)PW * PH);
    const float *g = drgb + bj * H * W;
    float acc = 0.f;
    for (int y = 2 * py - 2; y <= 2 * py + 3; ++y) {
        if (y < 0 || y >= H) continue;
        float sy = 0.5f * (float)y - 0.25f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy, y1 = y0 + (y0 < PH - 1 ? 1 : 0);
        const float ly = sy - (float)y0;
        if (y0 != py && y1 != py) continue;
        for (int x = 2 * px - 2; x <= 2 * px + 3; ++x) {
            if (x < 0 || x >= W) continue;
            float sx = 0.5f * (float)x - 0.25f;
            sx = sx < 0.f ? 0.f : sx;
            const int x0 = (int)sx, x1 = x0 + (x0 < PW - 1 ? 1 : 0);
            const float lx = sx - (float)x0;
            if (x0 != px && x1 != px) continue;
            const float gv = g[y * W + x];
#pragma unroll
            for (int corner = 0; corner < 4; ++corner) {
                const int cy = (corner >> 1) ? y1 : y0, cx = (corner & 1) ? x1 : x0;
                if (cy != py || cx != px) continue;
                const float wy = (corner >> 1) ? ly : 1.f - ly, wx = (corner & 1) ? lx : 1.f - lx;
                acc += gv * wy * wx;
            }
        }
    }
    dprev[i] = acc;
}

// dw[j][c] = sum_p drgb[p][j] rows[p][c], db[j] = sum_p drgb[p][j]: a block sums its slab of pixels into ITS plane of `parts`
// ([blocks][OC C + OC]: weight sums, then the bias sums); sum_planes_kernel adds the planes in block order (no atomics)
template <int OC>
__global__ __launch_bounds__(256) void rgb_head_wgrad_kernel(const float *__restrict__ drgb, const float *__restrict__ rows, float *__restrict__ parts,
                                                             int B, int HW, int C, int slab)
{
    float *dw = parts + (long long)blockIdx.x * (OC * C + OC);
    float *db = dw + OC * C;
    const long long total = (long long)B * HW;
    const long long p0 = (long long)blockIdx.x * slab;
    const long long p1 = p0 + slab < total ? p0 + slab : total;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a[OC], sm[OC];
#pragma unroll
        for (int j = 0; j < OC; ++j) a[j] = sm[j] = 0.f;
        for (long long p = p0; p < p1; ++p) {
            const long long b = p / HW;
            const int pix = (int)(p - b * HW);
            const float v = rows[p * C + c];
#pragma unroll
            for (int j = 0; j < OC; ++j) {
                const float g = drgb[(b * OC + j) * HW + pix];
                a[j] = fmaf(g, v, a[j]);
                sm[j] += g;
            }
        }
#pragma unroll
        for (int j = 0; j < OC; ++j) dw[j * C + c] = a[j];
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < OC; ++j) db[j] = sm[j];
        }
    }
}

// dw[i] (i < 3 C) and db[i - 3 C] = sum over planes of parts[plane][i], in plane order
__global__ __launch_bounds__(256) void sum_planes_kernel(const float *__restrict__ parts, int nplanes, int C, float *__restrict__ dw, float *__restrict__ db, int OC)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = OC * C + OC;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nplanes; ++k) s += parts[(long long)k * n + i];
    if (i < OC * C) dw[i] = s;
    else db[i - OC * C] = s;
}

}  // namespace

extern "C" int ldm_lrelu_bwd_f32(const float *dy, const float *y, float *dx, long long n, float slope, void *stream)
{
    LDM_REQUIRE(dy && y && dx && n > 0 && n % 4 == 0, "ldm_lrelu_bwd_f32: bad arguments (n %% 4 == 0)");
    LDM_REQUIRE(ldm_aligned16(dy) && ldm_aligned16(y) && ldm_aligned16(dx), "ldm_lrelu_bwd_f32: unaligned pointer");
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(vb_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)dy, (const f32x4 *)y, (f32x4 *)dx,
                       n / 4, slope);
    LDM_CHECK_LAUNCH("ldm_lrelu_bwd_f32");
    return LDM_OK;
}

extern "C" int ldm_im2col3x3_f32(const float *x, float *out, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ldm_im2col3x3_f32: bad arguments (C %% 4 == 0)");
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(out), "ldm_im2col3x3_f32: unaligned pointer");
    const long long total = (long long)B * H * W * 9 * (C / 4);
    LDM_REQUIRE(total / 256 < 0x7fffffffLL, "ldm_im2col3x3_f32: problem too large");
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(vb_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (f32x4 *)out, B, H, W, C / 4);
    LDM_CHECK_LAUNCH("ldm_im2col3x3_f32");
    return LDM_OK;
}

extern "C" int ldm_space_to_depth2_f32(const float *fine, float *out, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(fine && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ldm_space_to_depth2_f32: bad arguments (C %% 4 == 0)");
    LDM_REQUIRE(ldm_aligned16(fine) && ldm_aligned16(out), "ldm_space_to_depth2_f32: unaligned pointer");
    const long long total = (long long)B * H * W * 4 * (C / 4);
    LDM_REQUIRE(total / 256 < 0x7fffffffLL, "ldm_space_to_depth2_f32: problem too large");
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(vb_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)fine, (f32x4 *)out, B, H, W,
                       C / 4);
    LDM_CHECK_LAUNCH("ldm_space_to_depth2_f32");
    return LDM_OK;
}

extern "C" int ldm_rgb_head_bwd_f32(const float *drgb, const float *w, const float *rows, float *drows, int accumulate, float *dprev, float *dw,
                                    float *db, int B, int H, int W, int C, void *stream)
{
    return ldm_rgb_head_bwd_oc_f32(drgb, w, rows, drows, accumulate, dprev, dw, db, B, H, W, C, 3, stream);
}

template <int OC>
static void rgb_head_bwd_launch(const float *drgb, const float *w, const float *rows, float *drows, int accumulate, float *parts, unsigned nb, int slab,
                                int B, int H, int W, int C, hipStream_t st)
{
    hipLaunchKernelGGL(rgb_head_bwd_kernel<OC>, dim3(vb_blocks((long long)B * H * W, 4)), dim3(256), 0, st, drgb, w, drows, B, H, W, C, accumulate);
    hipLaunchKernelGGL(rgb_head_wgrad_kernel<OC>, dim3(nb), dim3(256), 0, st, drgb, rows, parts, B, H * W, C, slab);
}

extern "C" int ldm_rgb_head_bwd_oc_f32(const float *drgb, const float *w, const float *rows, float *drows, int accumulate, float *dprev, float *dw,
                                       float *db, int B, int H, int W, int C, int OC, void *stream)
{
    LDM_REQUIRE(drgb && w && rows && drows && dw && db, "ldm_rgb_head_bwd_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && OC >= 1 && OC <= 4 && (!dprev || (H % 2 == 0 && W % 2 == 0)),
                "ldm_rgb_head_bwd_f32: bad shape (1 <= output channels <= 4)");
    LDM_REQUIRE(ldm_aligned16(w) && ldm_aligned16(drows), "ldm_rgb_head_bwd_f32: unaligned pointer");
    const long long rowsn = (long long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (dprev)
        hipLaunchKernelGGL(bilinear2x_adjoint_kernel, dim3(vb_blocks((long long)B * OC * (H / 2) * (W / 2), 256)), dim3(256), 0, st, drgb, dprev, B, H, W, OC);
    long long slab = (rowsn + 511) / 512;                                 // at most ~512 planes for the fixed-order sum
    slab = slab < 256 ? 256 : slab;
    const unsigned nb = vb_blocks(rowsn, (int)slab);
    float *parts = (float *)ldm_scratch(st, (size_t)nb * (OC * C + OC) * sizeof(float));
    if (!parts) return LDM_ELAUNCH;
    if (OC == 3) rgb_head_bwd_launch<3>(drgb, w, rows, drows, accumulate, parts, nb, (int)slab, B, H, W, C, st);
    else if (OC == 1) rgb_head_bwd_launch<1>(drgb, w, rows, drows, accumulate, parts, nb, (int)slab, B, H, W, C, st);
    else if (OC == 2) rgb_head_bwd_launch<2>(drgb, w, rows, drows, accumulate, parts, nb, (int)slab, B, H, W, C, st);
    else rgb_head_bwd_launch<4>(drgb, w, rows, drows, accumulate, parts, nb, (int)slab, B, H, W, C, st);
    hipLaunchKernelGGL(sum_planes_kernel, dim3((OC * C + OC + 255) / 256), dim3(256), 0, st, (const float *)parts, (int)nb, C, dw, db, OC);
    LDM_CHECK_LAUNCH("ldm_rgb_head_bwd_f32");
    return LDM_OK;
}
