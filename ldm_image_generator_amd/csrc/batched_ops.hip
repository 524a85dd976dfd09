// Small launches of the bf16 training step folded together (round 2's profile: ~300 five-microsecond copies per step).
//
//   ldm_gconv_pack_bf16   conv.weight [C, 32, 3, 3] fp32 -> BOTH bf16 filter tables of the grouped conv in one launch: the forward's
//                         [C][tap][ci] and the data gradient's flipped, in/out-swapped [g*32 + ci][tap'][co] (was: permute copy, cast,
//                         flip, permute copy, cast -- five launches per block and step, the weights move every step)
//   ldm_replicate_f32     out[r][:] = src[:] for r < reps: the bias gradient that several biases of a block share (a column sum of dy)
//                         written as `reps` separate rows in one launch (was: one .clone() per bias)
//   ldm_pack3x3_f32       dense conv.weight [Cout, Cin, 3, 3] -> BOTH fp32 filter matrices of the implicit 3x3 GEMM in one launch: the forward's
//                         [Cout][tap][Cin] and the data gradient's mirrored, in/out-swapped [Cin][tap'][Cout] (train_vae.py's iteration did a
//                         permute copy in the forward and a flip + permute copy in the backward of each of its 80 convs)
#include "common.h"

namespace {

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short to_bf16(float v)
{
    const f32x2v t = {v, v};
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2v)) & 0xFFFFu);      // RNE, NaN stays NaN
}

__global__ __launch_bounds__(256) void gconv_pack_bf16_kernel(const float *__restrict__ w, unsigned short *__restrict__ fwd, unsigned short *__restrict__ rot, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * 288) return;
    const int row = i / 288, rem = i - row * 288;
    const int tap = rem >> 5, lane32 = rem & 31;
    const int ky = tap / 3, kx = tap - ky * 3;
    // forward table: row = output channel, (tap, ci)
    fwd[i] = to_bf16(w[((long long)row * 32 + lane32) * 9 + tap]);
    // data-gradient table: row = g * 32 + ci, (tap', co) with the taps mirrored
    const int g = row >> 5, ci = row & 31;
    rot[i] = to_bf16(w[((long long)(g * 32 + lane32) * 32 + ci) * 9 + (2 - ky) * 3 + (2 - kx)]);
}

// one thread per forward element (co, tap, ci): coalesced writes of the forward matrix; the data-gradient matrix gets the same value at
// [ci][8 - tap][co]
__global__ __launch_bounds__(256) void pack3x3_kernel(const float *__restrict__ w, float *__restrict__ fwd, float *__restrict__ dgrad, int Cout, int Cin)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)Cout * 9 * Cin;
    if (i >= n) return;
    const int ci = (int)(i % Cin);
    const long long r = i / Cin;
    const int tap = (int)(r % 9), co = (int)(r / 9);
    const float v = w[((long long)co * Cin + ci) * 9 + tap];
    if (fwd) fwd[i] = v;
    if (dgrad) dgrad[((long long)ci * 9 + (8 - tap)) * Cout + co] = v;
}

__global__ __launch_bounds__(256) void replicate_kernel(const float *__restrict__ src, float *__restrict__ out, int n, int reps)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = src[i];
    for (int r = 0; r < reps; ++r) out[(long long)r * n + i] = v;
}

}  // namespace

extern "C" int ldm_gconv_pack_bf16(const float *w, void *fwd_bf16, void *rot_bf16, int C, void *stream)
{
    LDM_REQUIRE(w && fwd_bf16 && rot_bf16 && C >= 32 && C % 32 == 0, "ldm_gconv_pack_bf16: bad arguments (C %% 32 == 0)");
    hipLaunchKernelGGL(gconv_pack_bf16_kernel, dim3((C * 288 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (unsigned short *)fwd_bf16,
                       (unsigned short *)rot_bf16, C);
    LDM_CHECK_LAUNCH("ldm_gconv_pack_bf16");
    return LDM_OK;
}

extern "C" int ldm_replicate_f32(const float *src, float *out, int n, int reps, void *stream)
{
    LDM_REQUIRE(src && out && n > 0 && reps > 0, "ldm_replicate_f32: bad arguments");
    hipLaunchKernelGGL(replicate_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, out, n, reps);
    LDM_CHECK_LAUNCH("ldm_replicate_f32");
    return LDM_OK;
}

extern "C" int ldm_pack3x3_f32(const float *w, float *fwd, float *dgrad, int Cout, int Cin, void *stream)
{
    LDM_REQUIRE(w && (fwd || dgrad) && Cout > 0 && Cin > 0, "ldm_pack3x3_f32: bad arguments");
    const long long n = (long long)Cout * 9 * Cin;
    LDM_REQUIRE((n + 255) / 256 <= 0x7fffffffLL, "ldm_pack3x3_f32: filter too large");
    hipLaunchKernelGGL(pack3x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, fwd, dgrad, Cout, Cin);
    LDM_CHECK_LAUNCH("ldm_pack3x3_f32");
    return LDM_OK;
}
