// Which ingredient of the GEMM inner loop costs matrix-pipe time?  (no global memory anywhere)
//  mode 0: NACC independent accumulators, operands in registers
//  mode 1: + operands re-read from LDS with ds_read_b128 (one A + one B read per 8 MFMAs, as the 128x64 tile does)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int lane = threadIdx.x & 63;
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            a = *(const f32x4 *)(lds + ((lane * 4 + it * 64) & 8188));
            b = *(const f32x4 *)(lds + ((lane * 4 + it * 128 + 4096) & 8188));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int MODE>
void run(float *out, int bpc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * bpc, iters = 20000 / NACC;
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 4 * NACC * 4096.0;
    printf("NACC %d mode %d blocks/CU %d: %.2f ms, %.1f TFLOP/s\n", NACC, MODE, bpc, ms, flops / ms / 1e9);
}
int main()
{
    float *out;
    hipMalloc(&out, 4096 * 1024 * sizeof(float));
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<1, 0>(out, bpc); run<2, 0>(out, bpc); run<4, 0>(out, bpc);
        run<2, 1>(out, bpc); run<4, 1>(out, bpc);
    }
    return 0;
}
