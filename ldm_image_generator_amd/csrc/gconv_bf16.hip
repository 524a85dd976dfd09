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


// ---- tiled version (W a power of two in 8 .. 64, C % 64 == 0): the input block goes through LDS --------------------------------------
// The kernel above loads every A fragment straight from global memory: 18 wave-instructions per tile that each touch 32 cache lines
// (32 bytes of every 128-byte line -- the line's other quarters belong to the neighbouring groups / the other k-half), with nothing
// in flight while the 18 MFMAs run.  It measured 3.0-3.4 TB/s of algorithmic bytes (floor: ~5).  Here a workgroup owns TWO groups
// (64 channels: 128 contiguous bytes per pixel) and a tile of R whole image rows (R * W = 128 pixels, or the 8 x 8 image); the tile
// with its one-pixel halo is copied to LDS by all 256 threads as 16-byte chunks (8-9 loads in flight per thread, zero padding written
// as zeros), pixels 80 bytes apart so that the 32 lanes of a fragment read (consecutive pixels, 16 bytes each) fall into distinct
// 16-byte bank slots.  A wave then owns one group and every other 32-pixel strip: 18 ds_read_b128 + 18 MFMAs per strip, same k order
// as above -> bit-identical results.  Three workgroups share a CU (42 KiB of LDS each at W = 64), so one stages while others compute.
constexpr int kPixB = 80;                 // bytes per pixel and group in LDS: 64 of data + 16 of padding

__global__ __launch_bounds__(256, 2) void gconv3x3_bf16_tiled_kernel(const GcP16 p, int R, int wshift, int tiles_per_image, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gl[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int gi = wave & 1, half = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int W = p.W, H = p.H, C = p.C;
    const int WP = W + 2, NPIX = (R + 2) * WP;              // halo image of one group: (R + 2) x (W + 2) pixels
    const int npair = p.G >> 1;                              // group pairs
    const int P = R * W;                                     // pixels of a tile (a multiple of 32)
    for (long long item = blockIdx.x; item < (long long)total_tiles * npair; item += gridDim.x) {
        const int pair = (int)(item % npair);
        const int tile = (int)(item / npair);
        const int b = tile / tiles_per_image, y0 = (tile - b * tiles_per_image) * R;
        const int g0 = 2 * pair;
        // ---- stage: (R + 2) x (W + 2) pixels x 2 groups x 4 chunks of 16 bytes ----------------------------------------------
        __syncthreads();                                     // the previous tile's fragment reads are complete
        for (int i = t; i < NPIX * 8; i += 256) {
            const int q = i & 7, pix = i >> 3;
            const int py = pix / WP, px = pix - py * WP;
            const int yy = y0 + py - 1, xx = px - 1;
            u32x4 v{0u, 0u, 0u, 0u};
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                v = *(const u32x4 *)(p.x + ((long long)(b * H + yy) * W + xx) * C + (g0 + (q >> 2)) * 32 + (q & 3) * 8);
            *(u32x4 *)(gl + ((q >> 2) * NPIX + pix) * kPixB + (q & 3) * 16) = v;
        }
        __syncthreads();
        // ---- compute: this wave's group, every other 32-pixel strip ---------------------------------------------------------
        const int g = g0 + gi;
        s16x8 wf[18];
        const unsigned short *wrow = p.w + ((long long)(g * 32 + r) * 9) * 32 + 8 * h;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int s = 0; s < 2; ++s) wf[2 * tp + s] = *(const s16x8 *)(wrow + tp * 32 + 16 * s);
        const float bias = p.bias ? p.bias[g * 32 + r] : 0.f;
        const unsigned char *gimg = gl + gi * NPIX * kPixB;
        for (int mt = half; mt < (P >> 5); mt += 2) {
            const int pl = mt * 32 + r;
            const int ty = pl >> wshift, tx = pl & (W - 1);
            const unsigned char *a0 = gimg + ((ty + 1) * WP + tx + 1) * kPixB + 16 * h;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int dy = tp / 3 - 1, dx = tp % 3 - 1;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const s16x8 af = *(const s16x8 *)(a0 + (dy * WP + dx) * kPixB + 32 * s);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[2 * tp + s], acc, 0, 0, 0);
                }
            }
            // C/D map: column = output channel r, row (pixel of the strip) = (e & 3) + 8 (e >> 2) + 4 h; the strip's pixels are consecutive
            // rows of the [M, C] matrices (whole image rows)
            const long long m0 = ((long long)(b * H + y0) * W) + mt * 32 + 4 * h;
            const long long obase = m0 * C + g * 32 + r;
            float add[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) add[e] = p.addend ? p.addend[obase + (long long)((e & 3) + 8 * (e >> 2)) * C] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) p.out[obase + (long long)((e & 3) + 8 * (e >> 2)) * C] = (acc[e] + bias) + add[e];
        }
    }
}

int g_gconv16_tiled = 1;      // 1 (default): tiled kernel where the shape allows; 0: the direct kernel everywhere (A/B tests; bit-identical)

}  // namespace

extern "C" int ldm_gconv3x3_bf16_tiled(int v)
{
    const int old = g_gconv16_tiled;
    if (v == 0 || v == 1) g_gconv16_tiled = v;
    return old;
}

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
    // tiled kernel: W a power of two in 8 .. 64 (a tile = R whole rows = 128 pixels, or the 8 x 8 image), H a multiple of R, group pairs
    int R = 0, wshift = 0;
    if (g_gconv16_tiled && (W == 8 || W == 16 || W == 32 || W == 64) && p.G % 2 == 0 && ldm_aligned16(out) && (!addend || ldm_aligned16(addend))) {
        R = W == 8 ? (H >= 8 ? 8 : 4) : 128 / W;
        while ((1 << wshift) < W) ++wshift;
        if (H % R || (R * W) % 32) R = 0;
    }
    if (R) {
        const int tiles_per_image = H / R;
        const long long total_tiles = (long long)B * tiles_per_image;
        const long long items = total_tiles * (p.G / 2);
        const size_t smem = (size_t)2 * (R + 2) * (W + 2) * kPixB;
        static LdmLdsOptIn opt_in;
        if (total_tiles < 0x7fffffffLL && opt_in((const void *)gconv3x3_bf16_tiled_kernel, smem)) {
            long long grid = (long long)cus * 3;
            if (grid > items) grid = items;
            ldm_launch(gconv3x3_bf16_tiled_kernel, dim3((unsigned)grid), dim3(256), smem, st, p, R, wshift, tiles_per_image, (int)total_tiles);
            ldm_prof_end(rec, st);
            LDM_CHECK_LAUNCH("ldm_gconv3x3_bf16");
            return LDM_OK;
        }
    }
    ldm_launch(gconv3x3_bf16_kernel, dim3((unsigned)(waves / 4)), dim3(256), 0, st, p);
    ldm_prof_end(rec, st);
    LDM_CHECK_LAUNCH("ldm_gconv3x3_bf16");
    return LDM_OK;
}
