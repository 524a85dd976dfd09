// Grouped 3x3 convolution (32 channels in / 32 out per group, zero pad 1: unet.py:30,44) with bf16 operands, for the bf16
// training step: forward  y = conv(xf) + bias + x   and data gradient  dxf += conv(dy, flipped / swapped filter)  are the
// same kernel with different filter tables.
//
// With the bf16 matrix cores the arithmetic of this layer (2 * 288 FLOP per output element) is 16x cheaper than on the exact
// fp32 MFMA, so the kernel is HBM-bound (bf16 input once, fp32 addend once, fp32 output once) and needs no LDS staging:
//   * a wave owns ONE group for its whole life and keeps the group's 9 x 32 x 32 filter as 18 MFMA B-fragments in registers;
//   * per 32-pixel tile it loads 18 A-fragments straight from global memory -- lane (pixel r, half h) reads the 16 bytes
//     x[pixel + tap][g * 32 + 16 s + 8 h ...] that ARE its v_mfma_f32_32x32x16_bf16 operand; the 9x re-reads of a pixel hit L1 / L2;
//   * the four waves of a workgroup work on the same pixels of four consecutive groups, so rows leave as 512-byte runs.
#include "common.h"

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GcP16 {
    const unsigned short *x;       // [M, C] bf16
    const unsigned short *w;       // [C][9][32] bf16: (group g, output co) row-major, tap-major inside, input channel fastest
    const float *bias;             // [C] or NULL
    const float *addend;           // [M, C] fp32 or NULL (may alias out)
    float *out;                    // [M, C] fp32
    int M, H, W, C, G;
    int tiles_m;                   // ceil(M / 32)
};

__global__ __launch_bounds__(256) void gconv3x3_bf16_kernel(const GcP16 p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = (int)blockIdx.x * 4 + wave, total = (int)gridDim.x * 4;           // total is a multiple of G (host)
    const int g = gw % p.G;
    const int r = lane & 31, h = lane >> 5;
    // filter: B operand of tap t, k-slice s = w[(g*32 + r)][t][16 s + 8 h .. + 7]  (column = output channel r)
    s16x8 wf[18];
    const unsigned short *wrow = p.w + ((long long)(g * 32 + r) * 9) * 32 + 8 * h;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[2 * t + s] = *(const s16x8 *)(wrow + t * 32 + 16 * s);
    const float bias = p.bias ? p.bias[g * 32 + r] : 0.f;
    const int step = total / p.G;
#pragma unroll 1
    for (int tm = gw / p.G; tm < p.tiles_m; tm += step) {
        const int m0 = tm * 32;
        const int m = m0 + r;
        const int xx = m % p.W, yy = (m / p.W) % p.H;
        const bool live = m < p.M;
        const unsigned short *xrow = p.x + (long long)m * p.C + g * 32 + 8 * h;
        s16x8 af[18];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = live && (unsigned)(yy + dy) < (unsigned)p.H && (unsigned)(xx + dx) < (unsigned)p.W;
            const unsigned short *src = xrow + (long long)(dy * p.W + dx) * p.C;
#pragma unroll
            for (int s = 0; s < 2; ++s) af[2 * t + s] = ok ? *(const s16x8 *)(src + 16 * s) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 18; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k], wf[k], acc, 0, 0, 0);
        // C/D map: column = output channel r, row (pixel) = (e & 3) + 8 (e >> 2) + 4 h
        // all 16 addend loads are issued BEFORE the first store: out may alias addend (in-place accumulation), so the compiler
        // must not be left to interleave them (it serialised load -> wait -> store 16 times: 10x slower)
        const long long obase = (long long)(m0 + 4 * h) * p.C + g * 32 + r;
        const int lastrow = p.M - 1 - (m0 + 4 * h);              // rows past M are clamped for the loads, skipped for the stores
        float add[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2);
            const int rc = row <= lastrow ? row : lastrow;             // (negative lastrow still lands on row M - 1)
            add[e] = p.addend ? p.addend[obase + (long long)rc * p.C] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2);
            if (row <= lastrow) p.out[obase + (long long)row * p.C] = (acc[e] + bias) + add[e];
        }
    }
}

}  // namespace

extern "C" int ldm_gconv3x3_bf16(const void *x, const void *w, const float *bias, const float *addend, float *out, int B, int H, int W, int C, void *stream)
{
    LDM_REQUIRE(x && w && out, "ldm_gconv3x3_bf16: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0, "ldm_gconv3x3_bf16: bad shape (C %% 32 == 0)");
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(w), "ldm_gconv3x3_bf16: unaligned pointer");
    const long long M = (long long)B * H * W;
    LDM_REQUIRE(M < (1ll << 31) - 64, "ldm_gconv3x3_bf16: too many pixels");
    GcP16 p{};
    p.x = (const unsigned short *)x; p.w = (const unsigned short *)w; p.bias = bias; p.addend = addend; p.out = out;
    p.M = (int)M; p.H = H; p.W = W; p.C = C; p.G = C / 32; p.tiles_m = (int)((M + 31) / 32);
    const int cus = ldm_cu_count();
    // waves: a multiple of lcm(4, G) (a wave keeps its group; a workgroup = 4 consecutive groups of the same pixels), ~8 per CU
    const int quantum = p.G % 4 == 0 ? p.G : 4 * p.G;
    const long long items = (long long)p.tiles_m * p.G;
    long long waves = (long long)cus * 8;
    if (waves > items) waves = items;
    waves = (waves + quantum - 1) / quantum * quantum;
    hipStream_t st = (hipStream_t)stream;
    void *rec = ldm_prof_begin(LDM_PROF_GCONV_BF16, 2.0 * (double)M * C * 288.0, st, (double)M * C * (2.0 + 4.0 + (addend ? 4.0 : 0.0)));
    hipLaunchKernelGGL(gconv3x3_bf16_kernel, dim3((unsigned)(waves / 4)), dim3(256), 0, st, p);
    ldm_prof_end(rec, st);
    LDM_CHECK_LAUNCH("ldm_gconv3x3_bf16");
    return LDM_OK;
}
