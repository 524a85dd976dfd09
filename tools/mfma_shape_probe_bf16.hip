// Which bf16 MFMA shape sustains more FLOP/s on RANDOM operands (the chip lowers its clock under a dense matrix stream:
// MI355X_MICROARCH.md, DVFS give-back)?  v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16, operands in registers, no memory
// traffic, same FLOPs and the same 64 accumulator registers; prints TFLOP/s and the in-kernel shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_shape_probe_bf16 tools/mfma_shape_probe_bf16.hip && tools/mfma_shape_probe_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, bool RANDOM>
__global__ __launch_bounds__(256) void k(float *out, const bf16x8 *vals, long long *clk, int iters)
{
    bf16x8 x[4], y[4];
    for (int i = 0; i < 4; ++i) {
        if (RANDOM) {
            x[i] = vals[(threadIdx.x * 8 + i) & 2047];
            y[i] = vals[(threadIdx.x * 8 + 4 + i) & 2047];
        } else {
            for (int e = 0; e < 8; ++e) { x[i][e] = 0x3f80; y[i][e] = 0x3f00; }
        }
    }
    const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(u + i) & 3], y[(u + 2 * i) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)                                  // 32 x 16384 FLOP = 16 x 32768 FLOP per iteration
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[(u + i) & 3], y[(u + 2 * i) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, bool RANDOM>
void run(float *out, const bf16x8 *vals, long long *clk, const char *name, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 16000;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, vals, clk, 200);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, vals, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * 16 * 32768.0;
        printf("%-34s %d WG/CU  %.2f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz\n", name, blocks / 256, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0));
    }
}

int main()
{
    float *out;
    bf16x8 *vals;
    long long *clk;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&vals, 2048 * sizeof(bf16x8));
    hipMalloc(&clk, 2 * sizeof(long long));
    static unsigned short h[2048 * 8];
    srand(1);
    for (int i = 0; i < 2048 * 8; ++i) {
        const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        unsigned u; __builtin_memcpy(&u, &f, 4);
        h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(vals, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<32, false>(out, vals, clk, "32x32x16 bf16 constant operands", blocks);
        run<16, false>(out, vals, clk, "16x16x32 bf16 constant operands", blocks);
        run<32, true>(out, vals, clk, "32x32x16 bf16 random operands", blocks);
        run<16, true>(out, vals, clk, "16x16x32 bf16 random operands", blocks);
    }
    return 0;
}
