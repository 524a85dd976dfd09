// Feasibility probe (NOT part of the product): fp32-accurate GEMM inner loop on the bf16 matrix cores by
// 3-way operand splitting (a = hi + mid + lo, 6 products kept: hh, hm, mh, hl, lh, mm -> error ~2^-26 |a||b|).
// Measures the LDS-fed inner loop only: A fragments arrive as fp32 (split on the fly, VALU), B fragments are
// pre-split bf16 planes.  Reports fp32-EQUIVALENT TFLOP/s (2*M*N*K), to compare with v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// truncating split: hi/mid/lo are the top 16 bits of x, x-hi, x-hi-mid (exact residuals); 2 dwords packed by v_perm
__device__ __forceinline__ void split3_trunc(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 H, M, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h[2], m[2], l[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = x[2 * i + j];
            h[j] = __float_as_uint(v) & 0xFFFF0000u;
            const float r1 = v - __uint_as_float(h[j]);
            m[j] = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(m[j]);
            l[j] = __float_as_uint(r2);
        }
        H[i] = __builtin_amdgcn_perm(h[1], h[0], 0x07060302u);
        M[i] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
        L[i] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, H); mid = __builtin_bit_cast(bf16x8, M); lo = __builtin_bit_cast(bf16x8, L);
}

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned u = __float_as_uint(x[i]);
        const unsigned h = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;       // round-to-nearest-even bf16 (no NaN here)
        const float r1 = x[i] - __uint_as_float(h);
        const unsigned u1 = __float_as_uint(r1);
        const unsigned m = (u1 + 0x7FFFu + ((u1 >> 16) & 1u)) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(m);
        const unsigned u2 = __float_as_uint(r2);
        const unsigned l = (u2 + 0x7FFFu + ((u2 >> 16) & 1u)) & 0xFFFF0000u;
        hi[i] = (short)(h >> 16); mid[i] = (short)(m >> 16); lo[i] = (short)(l >> 16);
    }
}

template <int TM, int SPLIT_A, int TN>
__global__ __launch_bounds__(256) void probe(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * (i % 977) - 0.4f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[TM * TN];
    for (int i = 0; i < TM * TN; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        // B: three pre-split planes (bf16x8 each) -> 3 x ds_read_b128 per fragment
        bf16x8 bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *(const bf16x8 *)(lds + ((lane * 4 + it * 64 + j * 256) & 8188));
            bm[j] = *(const bf16x8 *)(lds + ((lane * 4 + it * 64 + 2048 + j * 256) & 8188));
            bl[j] = *(const bf16x8 *)(lds + ((lane * 4 + it * 64 + 4096 + j * 256) & 8188));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            bf16x8 ah, am, al;
            if (SPLIT_A) {
                float x[8];
                const f32x4 x0 = *(const f32x4 *)(lds + ((lane * 8 + it * 128 + i * 1024) & 8184));
                const f32x4 x1 = *(const f32x4 *)(lds + ((lane * 8 + it * 128 + i * 1024 + 4) & 8188));
                for (int e = 0; e < 4; ++e) { x[e] = x0[e]; x[4 + e] = x1[e]; }
                if (SPLIT_A == 2) split3_trunc(x, ah, am, al); else split3(x, ah, am, al);
            } else {
                ah = *(const bf16x8 *)(lds + ((lane * 4 + it * 96 + i * 512) & 8188));
                am = *(const bf16x8 *)(lds + ((lane * 4 + it * 96 + i * 512 + 1024) & 8188));
                al = *(const bf16x8 *)(lds + ((lane * 4 + it * 96 + i * 512 + 3072) & 8188));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 &c = acc[i * TN + j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], c, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < TM * TN; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TM, int SPLIT_A, int TN>
void run(float *out, int bpc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * bpc, iters = 40000 / (TM * TN);
    hipLaunchKernelGGL((probe<TM, SPLIT_A, TN>), dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<TM, SPLIT_A, TN>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * TM * TN * 2.0 * 32 * 32 * 16;      // fp32-equivalent
    printf("TM %d TN %d  split-A-on-the-fly %d  blocks/CU %d: %.2f ms, %.1f fp32-equivalent TFLOP/s\n", TM, TN, SPLIT_A, bpc, ms, flops / ms / 1e9);
}

int main()
{
    float *out;
    hipMalloc(&out, 4096 * 1024 * sizeof(float));
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<2, 0, 1>(out, bpc);
        run<2, 1, 1>(out, bpc);
        run<2, 2, 1>(out, bpc);
        run<2, 2, 2>(out, bpc);
        run<1, 2, 2>(out, bpc);
        run<2, 0, 2>(out, bpc);
    }
    return 0;
}
