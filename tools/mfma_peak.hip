// Pure-MFMA ceiling of v_mfma_f32_32x32x2_f32 on this box (no memory traffic): what "100 %" means under DVFS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x = a + threadIdx.x * 1e-3f, y = b + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out;
    hipMalloc(&out, 4096 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        for (int rep = 0; rep < 3; ++rep) {
            const int blocks = 256 * bpc, iters = 20000;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f, 0.5f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 * iters * 16 * 4096.0;
            printf("blocks/CU %d: %.2f ms, %.1f TFLOP/s\n", bpc, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
