// VectorQuantizer of the VAE training path (reference vae.py:7-26): nearest-codebook-row search (integer result, must equal the
// reference's indices), embedding gather, and the two-sided L1 loss with its gradients.
//
// quantize (vae.py:18-22) is  argmax(-torch.cdist(x, embeddings), dim=2).  For more than 25 rows torch.cdist takes the
// matrix-multiplication route (ATen native/Distance.cpp, _euclidean_dist), and index parity needs ITS rounding, not the
// textbook sqrt(sum (x - e)^2):
//     xn = sum_k x_k^2, en = sum_k e_k^2         (each square rounded, then a sequential fp32 sum over k)
//     s  = sum_{k < D+2} a_k b_k,  a = [-2 x, xn, 1],  b = [e, 1, en]     (one FMA per k, in k order: the sgemm inner loop)
//     dist = sqrt(max(s, 0));   index = FIRST position of the maximum of -dist
// Pinned in the build container against torch 2.10 / MKL: the chain above reproduces torch's matmul bit for bit; torch's CPU
// sqrt (MKL VML) is NOT correctly rounded (0.7 % of values are one ulp off the IEEE result used here), so an index can differ
// from the reference's only where the two best distances agree to within one ulp -- oracle/ldm_vq_oracle.py documents and counts it.
#include "common.h"
#include <cmath>

namespace {

constexpr int VQ_DMAX = 16;
constexpr int VQ_CHUNK = 1024;          // codebook rows staged in LDS per pass

template <int D>
__global__ __launch_bounds__(256) void vq_quantize_kernel(const float *__restrict__ x, const float *__restrict__ emb, long long *__restrict__ idx,
                                                          long long M, int N)
{
    __shared__ float cb[VQ_CHUNK][D + 1];          // [e_0 .. e_{D-1}, en]
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = m < M;
    float a[D], xn = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const float v = live ? x[m * D + k] : 0.f;
        const float sq = __fmul_rn(v, v);
        xn = k == 0 ? sq : __fadd_rn(xn, sq);
        a[k] = __fmul_rn(v, -2.0f);                // exact
    }
    float best = 0.f;
    int best_i = -1;
    for (int c0 = 0; c0 < N; c0 += VQ_CHUNK) {
        const int cn = N - c0 < VQ_CHUNK ? N - c0 : VQ_CHUNK;
        __syncthreads();
        for (int i = threadIdx.x; i < cn; i += 256) {
            float en = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float v = emb[(long long)(c0 + i) * D + k];
                cb[i][k] = v;
                const float sq = __fmul_rn(v, v);
                en = k == 0 ? sq : __fadd_rn(en, sq);
            }
            cb[i][D] = en;
        }
        __syncthreads();
        for (int i = 0; i < cn; ++i) {
            float s = __fmul_rn(a[0], cb[i][0]);                       // fma(a0, b0, 0)
#pragma unroll
            for (int k = 1; k < D; ++k) s = __fmaf_rn(a[k], cb[i][k], s);
            s = __fmaf_rn(xn, 1.0f, s);
            s = __fmaf_rn(1.0f, cb[i][D], s);
            const float prob = -__fsqrt_rn(fmaxf(s, 0.f));
            // torch.argmax: first maximum; NaN counts as the maximum
            const bool better = best_i < 0 || prob > best || (prob != prob && best == best);
            if (better) {
                best = prob;
                best_i = c0 + i;
            }
        }
    }
    if (live) idx[m] = best_i;
}

__global__ void vq_embed_kernel(const long long *__restrict__ idx, const float *__restrict__ emb, float *__restrict__ out, long long M, int D)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * D) return;
    const long long m = i / D;
    out[i] = emb[idx[m] * D + (i - m * D)];
}

// loss[0] = mean |x - e| + mean |e - x|  (vae.py:12-16: reg_loss + embedding_loss, each an F.l1_loss over M * D elements)
// Per-block sums go to `parts` (one float per block); vq_loss_finish_kernel adds them in block order -- no float atomics, so the loss is
// bit-reproducible from run to run.
__global__ __launch_bounds__(256) void vq_loss_kernel(const float *__restrict__ x, const float *__restrict__ e, long long n, float *__restrict__ parts)
{
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += fabsf(x[i] - e[i]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// one block: loss = 2 / n * (sum of the per-block sums: fixed strided partial sums, then a fixed tree)
__global__ __launch_bounds__(256) void vq_loss_finish_kernel(const float *__restrict__ parts, int nparts, float inv_n, float *__restrict__ loss)
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
    if (threadIdx.x == 0) loss[0] = 2.0f * inv_n * sh[0];
}

// dx = g * sign(x - e) / n (reg_loss, e detached);  demb[idx[m]] += g * sign(e - x) / n (embedding_loss, x detached).
// Every contribution to a codebook cell is +-(g / n) or 0, so the cell's sum is an INTEGER count times g / n: the counts are
// accumulated with integer atomics (associative: independent of the order of arrival) in the cells themselves and
// vq_demb_finish_kernel multiplies once -- bit-reproducible, unlike a float scatter-add.
__global__ void vq_loss_bwd_kernel(const float *__restrict__ x, const float *__restrict__ e, const long long *__restrict__ idx, const float *__restrict__ g,
                                   float inv_n, float *__restrict__ dx, int *__restrict__ counts, long long M, int D)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * D) return;
    const float d = x[i] - e[i];
    const int sg = d > 0.f ? 1 : (d < 0.f ? -1 : 0);
    dx[i] = (g[0] * inv_n) * (float)sg;
    const long long m = i / D;
    if (sg) atomicAdd(counts + idx[m] * D + (i - m * D), -sg);
}

__global__ void vq_demb_finish_kernel(float *__restrict__ demb, const float *__restrict__ g, float inv_n, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = ((const int *)demb)[i];                              // the counts sit in the gradient's own 4-byte cells
    demb[i] = (g[0] * inv_n) * (float)c;
}

}  // namespace

extern "C" int ldm_vq_quantize_f32(const float *x, const float *emb, long long *idx, long long M, int N, int D, void *stream)
{
    LDM_REQUIRE(x && emb && idx && M > 0 && N > 0, "ldm_vq_quantize_f32: bad arguments");
    LDM_REQUIRE(D == 8 || D == 4 || D == 16, "ldm_vq_quantize_f32: dim %d not built (4, 8, 16)", D);
    const unsigned blocks = (unsigned)((M + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (D == 8) hipLaunchKernelGGL(vq_quantize_kernel<8>, dim3(blocks), dim3(256), 0, st, x, emb, idx, M, N);
    else if (D == 4) hipLaunchKernelGGL(vq_quantize_kernel<4>, dim3(blocks), dim3(256), 0, st, x, emb, idx, M, N);
    else hipLaunchKernelGGL(vq_quantize_kernel<16>, dim3(blocks), dim3(256), 0, st, x, emb, idx, M, N);
    LDM_CHECK_LAUNCH("ldm_vq_quantize_f32");
    return LDM_OK;
}

extern "C" int ldm_vq_embed_f32(const long long *idx, const float *emb, float *out, long long M, int D, void *stream)
{
    LDM_REQUIRE(idx && emb && out && M > 0 && D > 0, "ldm_vq_embed_f32: bad arguments");
    hipLaunchKernelGGL(vq_embed_kernel, dim3((unsigned)((M * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, emb, out, M, D);
    LDM_CHECK_LAUNCH("ldm_vq_embed_f32");
    return LDM_OK;
}

extern "C" int ldm_vq_loss_f32(const float *x, const float *e, long long n, float *loss, void *stream)
{
    LDM_REQUIRE(x && e && loss && n > 0, "ldm_vq_loss_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    unsigned blocks = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > 1024) blocks = 1024;
    float *parts = (float *)ldm_scratch(st, blocks * sizeof(float));
    if (!parts) return LDM_ELAUNCH;
    hipLaunchKernelGGL(vq_loss_kernel, dim3(blocks), dim3(256), 0, st, x, e, n, parts);
    hipLaunchKernelGGL(vq_loss_finish_kernel, dim3(1), dim3(256), 0, st, (const float *)parts, (int)blocks, 1.0f / (float)n, loss);
    LDM_CHECK_LAUNCH("ldm_vq_loss_f32");
    return LDM_OK;
}

extern "C" int ldm_vq_loss_bwd_f32(const float *x, const float *e, const long long *idx, const float *gscale, float *dx, float *demb, long long M, int N,
                                   int D, void *stream)
{
    LDM_REQUIRE(x && e && idx && gscale && dx && demb && M > 0 && N > 0 && D > 0, "ldm_vq_loss_bwd_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(demb, 0, (size_t)N * D * sizeof(float), st) != hipSuccess) { ldm_set_error("ldm_vq_loss_bwd_f32: memset failed"); return LDM_ELAUNCH; }
    hipLaunchKernelGGL(vq_loss_bwd_kernel, dim3((unsigned)((M * D + 255) / 256)), dim3(256), 0, st, x, e, idx, gscale, 1.0f / (float)(M * D), dx, (int *)demb, M, D);
    hipLaunchKernelGGL(vq_demb_finish_kernel, dim3((unsigned)(((long long)N * D + 255) / 256)), dim3(256), 0, st, demb, gscale, 1.0f / (float)(M * D), (long long)N * D);
    LDM_CHECK_LAUNCH("ldm_vq_loss_bwd_f32");
    return LDM_OK;
}
