// Which exact-fp32 MFMA shape sustains more FLOP/s on RANDOM operands (the chip lowers its clock under a dense matrix stream:
// MI355X_MICROARCH.md, DVFS give-back)?  v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, operands in registers, no memory
// traffic, same FLOPs; prints TFLOP/s and the in-kernel shader clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_shape_probe tools/mfma_shape_probe.hip && tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool RANDOM>
__global__ __launch_bounds__(256) void k(float *out, const float *vals, long long *clk, int iters)
{
    // 16 operand values per lane (random bits or one constant), cycled through the MFMAs
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = RANDOM ? vals[(threadIdx.x * 16 + i) & 4095] : 1.0f;
        y[i] = RANDOM ? vals[(threadIdx.x * 16 + 8 + i) & 4095] : 0.5f;
    }
    const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(u + i) & 7], y[(u + 2 * i) & 7], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[16];                                                   // same 64 accumulator registers
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)                                  // 64 x 2048 FLOP = 32 x 4096 FLOP per iteration
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[(u + i) & 7], y[(u + 2 * i) & 7], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, bool RANDOM>
void run(float *out, const float *vals, long long *clk, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = 8000;                                // two workgroups per CU = two waves per SIMD
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, vals, clk, 200);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, vals, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
        printf("%-28s %.2f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz\n", name, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0));
    }
}

int main()
{
    float *out, *vals;
    long long *clk;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&vals, 4096 * sizeof(float));
    hipMalloc(&clk, 2 * sizeof(long long));
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(vals, h, sizeof(h), hipMemcpyHostToDevice);
    run<32, false>(out, vals, clk, "32x32x2  constant operands");
    run<16, false>(out, vals, clk, "16x16x4  constant operands");
    run<32, true>(out, vals, clk, "32x32x2  random operands");
    run<16, true>(out, vals, clk, "16x16x4  random operands");
    return 0;
}
