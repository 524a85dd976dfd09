// Weight gradient of the grouped 3x3 conv (32 in / 32 out per group; autograd of unet.py:30,44) from bf16 operands:
//     dW[g][co][tap][ci] = sum over pixels m of dy[m, g*32 + co] * x[m + tap, g*32 + ci]        (zero outside the image)
//
// The contraction runs over PIXELS, so both MFMA operands need "8 consecutive pixels of one channel" per lane: [pixel][32 ch]
// LDS images read with ds_read_b64_tr_b16 (64-byte rows: the 4 x 16 blocks of a half-wave cover all 64 banks, no swizzle needed).
// Border handling without per-element masks: the kernel works in the ZERO-PADDED index space of the images, p = (b, y + 1, x + 1) in a
// (H + 2) x (W + 2) frame -- a tap is then a constant offset dy * (W + 2) + dx everywhere, border slots hold zeros in BOTH images (a
// zero dy contributes nothing).  Cost: (H + 2)(W + 2) / HW more pixels (6 % at 64 x 64).
// One workgroup = one group x one range of padded pixels; its four waves take different 16-pixel slices of every 128-pixel tile and
// keep all nine 32 x 32 tap tiles in registers; the 4 x splits partial planes are summed by the caller (ldm_reduce_partials_f32).
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

constexpr int GT = 128;            // padded pixels per tile

struct GwP16 {
    const unsigned short *x, *dy;
    float *planes;
    int B, H, W, C, Wp, HpWp;
    long long Mp, per_split;
};

__device__ __forceinline__ s16x8 tr_frag(const unsigned short *img, int row0, int lane)
{
    // lane -> (group g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3); operand lane (r = lane & 31, hh = lane >> 5) receives rows
    // row0 + 8 hh + {0..7} of column r
    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned short *a = img + (row0 + 8 * (grp >> 1) + q) * 32 + 16 * (grp & 1) + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 4 * 32));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256) void gconv3x3_wgrad_bf16_kernel(const GwP16 p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    const int halo = p.Wp + 1;
    const int NP = GT + 2 * halo;
    unsigned short *dyS = lds16, *xS = lds16 + GT * 32;
    const int g = blockIdx.x, s = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long start = (long long)s * p.per_split;
    const long long end = start + p.per_split < p.Mp ? start + p.per_split : p.Mp;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // padded index -> address of the pixel's 32 channels of group g, or nullptr for a border / out-of-range slot
    auto src = [&](const unsigned short *base, long long pp_, long long hi_) -> const unsigned short * {
        if (pp_ < 0 || pp_ >= hi_) return nullptr;
        const int img = (int)(pp_ / p.HpWp);
        const int rem = (int)(pp_ - (long long)img * p.HpWp);
        const int yp = rem / p.Wp, xp = rem - yp * p.Wp;
        if (yp < 1 || yp > p.H || xp < 1 || xp > p.W) return nullptr;
        return base + (((long long)img * p.H + (yp - 1)) * p.W + (xp - 1)) * p.C + g * 32;
    };

#pragma unroll 1
    for (long long p0 = start; p0 < end; p0 += GT) {
        __syncthreads();                                   // every wave is done reading the previous tile
        for (int pix = tid; pix < GT + NP; pix += 256) {
            const bool is_dy = pix < GT;
            const unsigned short *sp = is_dy ? src(p.dy, p0 + pix, end) : src(p.x, p0 - halo + (pix - GT), p.Mp);
            unsigned short *dst = is_dy ? dyS + pix * 32 : xS + (pix - GT) * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 v = sp ? *(const u32x4 *)(sp + 8 * c) : u32x4{0u, 0u, 0u, 0u};
                *(u32x4 *)(dst + 8 * c) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int sl = 2 * wave + k;                   // this wave's 16-pixel slices of the tile
            const s16x8 a = tr_frag(dyS, 16 * sl, lane);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int off = (t / 3 - 1) * p.Wp + (t % 3 - 1);
                const s16x8 b = tr_frag(xS, halo + off + 16 * sl, lane);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D map: column = ci (lane & 31), row = co = (e & 3) + 8 (e >> 2) + 4 (lane >> 5); plane (s * 4 + wave) [C][288]
    const int r = lane & 31, h = lane >> 5;
    float *plane = p.planes + ((long long)(s * 4 + wave) * p.C + g * 32) * 288 + r;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
            plane[(long long)co * 288 + t * 32] = acc[t][e];
        }
}

}  // namespace

extern "C" int ldm_gconv3x3_wgrad_bf16_splits(int B, int H, int W, int C)
{
    // enough (group, split) workgroups to fill the chip twice, at least four 128-pixel tiles per workgroup
    const long long Mp = (long long)B * (H + 2) * (W + 2);
    const int G = C / 32;
    int s = 1;
    while ((long long)G * s < 512 && Mp / (2 * s) >= 4 * GT && s < 256) s *= 2;
    return s;
}

extern "C" int ldm_gconv3x3_wgrad_bf16(const void *x, const void *dy, float *out_planes, int B, int H, int W, int C, int splits, void *stream)
{
    LDM_REQUIRE(x && dy && out_planes, "ldm_gconv3x3_wgrad_bf16: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0 && splits >= 1 && splits <= 65535, "ldm_gconv3x3_wgrad_bf16: bad shape (C %% 32 == 0)");
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(dy), "ldm_gconv3x3_wgrad_bf16: unaligned pointer");
    GwP16 p{};
    p.x = (const unsigned short *)x; p.dy = (const unsigned short *)dy; p.planes = out_planes;
    p.B = B; p.H = H; p.W = W; p.C = C; p.Wp = W + 2; p.HpWp = (H + 2) * (W + 2);
    p.Mp = (long long)B * p.HpWp;
    p.per_split = ((p.Mp + splits - 1) / splits + GT - 1) / GT * GT;
    const size_t smem = ((size_t)GT * 32 + (size_t)(GT + 2 * (p.Wp + 1)) * 32) * sizeof(unsigned short);
    LDM_REQUIRE(smem <= 150 * 1024, "ldm_gconv3x3_wgrad_bf16: W=%d too wide for the LDS halo image", W);
    static LdmLdsOptIn opt_in;
    (void)opt_in((const void *)gconv3x3_wgrad_bf16_kernel, 150 * 1024);
    hipStream_t st = (hipStream_t)stream;
    void *rec = ldm_prof_begin(LDM_PROF_GCONV_BF16, 2.0 * (double)B * H * W * C * 288.0, st, 4.0 * (double)B * H * W * C + 4.0 * 4 * splits * C * 288.0);
    ldm_launch(gconv3x3_wgrad_bf16_kernel, dim3(C / 32, splits), dim3(256), smem, st, p);
    ldm_prof_end(rec, st);
    LDM_CHECK_LAUNCH("ldm_gconv3x3_wgrad_bf16");
    return LDM_OK;
}
