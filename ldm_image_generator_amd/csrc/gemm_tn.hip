// "TN" fp32-MFMA GEMM for weight gradients:  out[s][n][k] = sum_{m in split s} A[m][n] * B[m][k]
// (train_ldm.py:81-86 -> autograd of every 1x1 conv / Linear: dW = dY^T X, contraction over the PIXEL rows).
//
// Both operands are row-major with the contraction index on the ROWS, so the NT kernels would need explicit
// transposed copies of dY and X (8 % of a training step).  Here tiles [32 rows][128 columns] of both operands go
// global -> LDS by LDS-DMA as they lie in memory, and the MFMA operands are read along the columns: a lane reads TWO
// adjacent columns (ds_read_b64) of row 2*step + (lane >> 5) and uses them for two different accumulator tiles -- the
// rows (columns) of an output tile are then the even (odd) columns of the operand, which only the epilogue has to
// know.  128 x 128 output tile per workgroup, 4 waves of 64 x 64, fp32 exact (v_mfma_f32_32x32x2_f32).
// The reduction is split over grid groups (deterministic: partials + a fixed-order sum by the caller).
#include "common.h"

namespace {

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const float *src, float *lds_dst)
{
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

struct TnP {
    const float *a, *b;
    float *out, *colsum;      // colsum: optional [splits][N] column sums of A (the bias gradient that goes with dW)
    long long lda, ldb;
    int M, N, K, ms, splits;  // ms = rows per split
    int ntn, ntk;
    // CONV (implicit weight gradient of a dense 3x3 conv, zero pad 1): b is the conv INPUT as rows [M = B*H*W, Cin]; column k of the
    // virtual B matrix is (tap, ci) = (k / Cin, k % Cin) and its row m is x[pixel m shifted by the tap][ci] (0 outside the image, and
    // for the padding taps >= 9 that make K a multiple of 128); Nreal <= N: columns of A past Nreal read zeros
    int H, W, Cin, Nreal;
    float inv_w;
};

static __device__ __attribute__((aligned(64))) float tn_zero_block[16];

constexpr int BT = 128;       // output tile edge
constexpr int BR = 32;        // operand rows (contraction) per stage
constexpr int STAGE = 2 * BR * BT;

// LDS rows are 512 B (32 chunks of 16 B).  The two half-waves of a fragment read touch rows m and m + 1: chunk bit 4
// is XORed with (m & 1) so that they fall into different halves of the banks (applied on the DMA source side).
template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnP p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8 (round-robin dispatch).  All tiles of one split read the same rows of
    // A and B, so they are given ids b, b + 8, b + 16, ... -- same XCD, consecutive dispatch waves -- and share its L2
    const int T = p.ntn * p.ntk;
    int tile, split;
    if (p.splits % 8 == 0) {
        const int b = (int)blockIdx.x, blk = b / (8 * T), in = b - blk * 8 * T;
        split = blk * 8 + (in & 7);
        tile = in >> 3;
    } else {
        tile = (int)blockIdx.x % T;
        split = (int)blockIdx.x / T;
    }
    const int n0 = (tile / p.ntk) * BT, k0 = (tile % p.ntk) * BT;
    const long long row0 = (long long)split * p.ms;
    // the last split may be shorter (ms is rounded up so that tiles x splits fills ONE round of the 2-per-CU workgroup slots)
    const int nsteps = (int)((p.M - row0 < p.ms ? p.M - row0 : p.ms) / BR);

    // DMA: one instruction = 64 lanes x 16 B = two rows of one operand tile; a wave moves rows {2 (4 i + wave), +1}
    const int lrow = lane >> 5, lchunk = lane & 31;
    // CONV: this lane's two possible source chunks (row parity 0 / 1 swaps the halves of the 128-column tile) -> tap shift and channel
    int c_shift[2] = {0, 0}, c_dy[2] = {0, 0}, c_dx[2] = {0, 0}, c_ci[2] = {0, 0};
    bool c_tap_ok[2] = {true, true}, a_ok[2] = {true, true};
    if constexpr (CONV) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int csrc = (lchunk ^ (par << 4)) * 4;
            const int kcol = k0 + csrc;
            const int tap = kcol / p.Cin;
            c_ci[par] = kcol - tap * p.Cin;
            c_tap_ok[par] = tap < 9;
            c_dy[par] = tap / 3 - 1;
            c_dx[par] = tap - (tap / 3) * 3 - 1;
            c_shift[par] = c_dy[par] * p.W + c_dx[par];
            a_ok[par] = n0 + csrc < p.Nreal;
        }
    }
    // CONV: position inside its image of each of the four rows this lane moves per step, advanced by BR per step (a 64-bit m % W and
    // (m / W) % H per row and step was 40 % of this kernel's time); steps are issued in order 0, 1, 2, ...
    const int HW = CONV ? p.H * p.W : 1;
    int pix[4] = {0, 0, 0, 0};
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pix[i] = (int)((row0 + 2 * (4 * i + wave) + lrow) % HW);
    }
    auto issue = [&](int step) {
        float *As = lds + (step & 1) * STAGE, *Bs = As + BR * BT;
        const long long mbase = row0 + (long long)step * BR;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 2 * (4 * i + wave) + lrow;
            const int csrc = (lchunk ^ ((row & 1) << 4)) * 4;
            if constexpr (CONV) {
                const int par = row & 1;
                const long long m = mbase + row;
                int yy = (int)((float)pix[i] * p.inv_w), xx = pix[i] - yy * p.W;        // float estimate of pix / W, corrected to exact
                if (xx < 0) {
                    --yy;
                    xx += p.W;
                } else if (xx >= p.W) {
                    ++yy;
                    xx -= p.W;
                }
                pix[i] += BR;
                while (pix[i] >= HW) pix[i] -= HW;
                const bool ok = c_tap_ok[par] && (unsigned)(yy + c_dy[par]) < (unsigned)p.H && (unsigned)(xx + c_dx[par]) < (unsigned)p.W;
                glds16(a_ok[par] ? p.a + m * p.lda + n0 + csrc : tn_zero_block + (lchunk & 3) * 4, As + (4 * i + wave) * 256);
                glds16(ok ? p.b + (m + c_shift[par]) * p.ldb + c_ci[par] : tn_zero_block + (lchunk & 3) * 4, Bs + (4 * i + wave) * 256);
            } else {
                glds16(p.a + (mbase + row) * p.lda + n0 + csrc, As + (4 * i + wave) * 256);
                glds16(p.b + (mbase + row) * p.ldb + k0 + csrc, Bs + (4 * i + wave) * 256);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment address inside a stage: row m, columns (w * 64 + 2 r, + 1); physical chunk = logical ^ ((m & 1) << 4)
    auto frag = [&](const float *base, int m, int w) {
        const int col = w * 64 + 2 * r;
        const int chunk = (col >> 2) ^ ((m & 1) << 4);
        return *(const f32x2 *)(base + m * BT + chunk * 4 + (col & 3));
    };

    float cs[2] = {0.f, 0.f};                                  // column sums of this lane's two A columns, rows of parity h
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) issue(step + 1);
        const float *As = lds + (step & 1) * STAGE, *Bs = As + BR * BT;
        f32x2 a = frag(As, h, wm), b = frag(Bs, h, wn);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            f32x2 an = a, bn = b;
            if (s + 1 < 16) {
                an = frag(As, 2 * (s + 1) + h, wm);
                bn = frag(Bs, 2 * (s + 1) + h, wn);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            cs[0] += a[0];
            cs[1] += a[1];
            a = an;
            b = bn;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.colsum && k0 == 0 && wn == 0) {                       // one workgroup column and one wave column own each A column
        const float s0 = cs[0] + __shfl_xor(cs[0], 32), s1 = cs[1] + __shfl_xor(cs[1], 32);
        if (h == 0) *(f32x2 *)(p.colsum + (long long)split * p.N + n0 + wm * 64 + 2 * r) = f32x2{s0, s1};
    }
    // epilogue: tile (i, j) element (row q, column c) is out[n0 + wm*64 + 2 q + i][k0 + wn*64 + 2 c + j]
    float *obase = p.out + (long long)split * p.N * p.K + (long long)(n0 + wm * 64) * p.K + k0 + wn * 64 + 2 * r;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = (e & 3) + 8 * (e >> 2) + 4 * h;
            *(f32x2 *)(obase + (long long)(2 * q + i) * p.K) = f32x2{acc[i][0][e], acc[i][1][e]};
        }
}

// Narrow variant for Cout <= 64 (the C = 64 level of the VAE at 256 x 256 and the Discriminator's 32 / 48-channel stages, vae.py:57-58,
// 135-141): output tile BN x 128 with BN = 64 or 32 rows (A columns).  The 128-row tile computed 2 x / 4 x zero rows for these layers.
// All four waves share the BN A columns and take 32 B columns each: a lane reads ONE B column (ds_read_b32) and, for BN = 64, two
// adjacent A columns (even / odd output rows as in the wide kernel) or, for BN = 32, one.  Same DMA scheme (B: two 512-byte rows per
// instruction; A: 1024 / (4 BN) rows per instruction), same split / tile order, same fixed-order partial planes.
template <bool CONV, int BN>
__global__ __launch_bounds__(256, 2) void gemm_tn_narrow_kernel(const TnP p)
{
    static_assert(BN == 64 || BN == 32, "narrow tile: 64 or 32 A columns");
    constexpr int AI = BN / 32;                    // accumulator tiles per wave
    constexpr int NSTAGE = BR * BN + BR * BT;      // floats per stage: A tile, then B tile
    constexpr int ACH = BN / 4;                    // 16-byte chunks per A row
    constexpr int ARI = 64 / ACH;                  // A rows per DMA instruction (4 / 8)
    constexpr int AQ = BR / ARI / 4;               // A instructions per wave and step (2 / 1)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave;
    const int r = lane & 31, h = lane >> 5;
    const int T = p.ntn * p.ntk;
    int tile, split;
    if (p.splits % 8 == 0) {
        const int b = (int)blockIdx.x, blk = b / (8 * T), in = b - blk * 8 * T;
        split = blk * 8 + (in & 7);
        tile = in >> 3;
    } else {
        tile = (int)blockIdx.x % T;
        split = (int)blockIdx.x / T;
    }
    const int n0 = (tile / p.ntk) * BN, k0 = (tile % p.ntk) * BT;
    const long long row0 = (long long)split * p.ms;
    const int nsteps = (int)((p.M - row0 < p.ms ? p.M - row0 : p.ms) / BR);

    const int lrow = lane >> 5, lchunk = lane & 31;            // B: two rows per instruction
    const int arow = lane / ACH, achunk = lane % ACH;          // A: ARI rows per instruction
    int c_shift[2] = {0, 0}, c_dy[2] = {0, 0}, c_dx[2] = {0, 0}, c_ci[2] = {0, 0};
    bool c_tap_ok[2] = {true, true};
    bool a_ok = true;
    const int HW = CONV ? p.H * p.W : 1;
    int pix[4] = {0, 0, 0, 0};
    if constexpr (CONV) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int kcol = k0 + (lchunk ^ (par << 4)) * 4;
            const int tap = kcol / p.Cin;
            c_ci[par] = kcol - tap * p.Cin;
            c_tap_ok[par] = tap < 9;
            c_dy[par] = tap / 3 - 1;
            c_dx[par] = tap - (tap / 3) * 3 - 1;
            c_shift[par] = c_dy[par] * p.W + c_dx[par];
        }
        a_ok = n0 + achunk * 4 < p.Nreal;
#pragma unroll
        for (int i = 0; i < 4; ++i) pix[i] = (int)((row0 + 2 * (4 * i + wave) + lrow) % HW);
    }
    auto issue = [&](int step) {
        float *As = lds + (step & 1) * NSTAGE, *Bs = As + BR * BN;
        const long long mbase = row0 + (long long)step * BR;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int inst = 4 * q + wave;
            const long long m = mbase + ARI * inst + arow;
            glds16((!CONV || a_ok) ? p.a + m * p.lda + n0 + achunk * 4 : tn_zero_block + (lane & 3) * 4, As + inst * 256);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 2 * (4 * i + wave) + lrow;
            const int csrc = (lchunk ^ ((row & 1) << 4)) * 4;
            if constexpr (CONV) {
                const int par = row & 1;
                const long long m = mbase + row;
                int yy = (int)((float)pix[i] * p.inv_w), xx = pix[i] - yy * p.W;
                if (xx < 0) {
                    --yy;
                    xx += p.W;
                } else if (xx >= p.W) {
                    ++yy;
                    xx -= p.W;
                }
                pix[i] += BR;
                while (pix[i] >= HW) pix[i] -= HW;
                const bool ok = c_tap_ok[par] && (unsigned)(yy + c_dy[par]) < (unsigned)p.H && (unsigned)(xx + c_dx[par]) < (unsigned)p.W;
                glds16(ok ? p.b + (m + c_shift[par]) * p.ldb + c_ci[par] : tn_zero_block + (lchunk & 3) * 4, Bs + (4 * i + wave) * 256);
            } else {
                glds16(p.b + (mbase + row) * p.ldb + k0 + csrc, Bs + (4 * i + wave) * 256);
            }
        }
    };

    f32x16 acc[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    auto frag_a = [&](const float *base, int m, float (&a)[AI]) {
        if constexpr (AI == 2) {
            const f32x2 v = *(const f32x2 *)(base + m * BN + 2 * r);
            a[0] = v[0];
            a[1] = v[1];
        } else {
            a[0] = base[m * BN + r];
        }
    };
    auto frag_b = [&](const float *base, int m) {
        const int col = wn * 32 + r;
        const int chunk = (col >> 2) ^ ((m & 1) << 4);
        return base[m * BT + chunk * 4 + (col & 3)];
    };

    float cs[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) cs[i] = 0.f;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) issue(step + 1);
        const float *As = lds + (step & 1) * NSTAGE, *Bs = As + BR * BN;
        float a[AI], b;
        frag_a(As, h, a);
        b = frag_b(Bs, h);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float an[AI], bn = b;
#pragma unroll
            for (int i = 0; i < AI; ++i) an[i] = a[i];
            if (s + 1 < 16) {
                frag_a(As, 2 * (s + 1) + h, an);
                bn = frag_b(Bs, 2 * (s + 1) + h);
            }
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, acc[i], 0, 0, 0);
                cs[i] += a[i];
                a[i] = an[i];
            }
            b = bn;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.colsum && k0 == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const float sum = cs[i] + __shfl_xor(cs[i], 32);
            if (h == 0) p.colsum[(long long)split * p.N + n0 + AI * r + i] = sum;
        }
    }
    // tile i element (row q, column c) is out[n0 + AI q + i][k0 + wn * 32 + c]
    float *obase = p.out + (long long)split * p.N * p.K + (long long)n0 * p.K + k0 + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = (e & 3) + 8 * (e >> 2) + 4 * h;
            obase[(long long)(AI * q + i) * p.K] = acc[i][e];
        }
}

}  // namespace

extern "C" int ldm_gemm_tn_f32(const float *a, long long lda, const float *b, long long ldb, float *out, float *colsum_a, int M, int N,
                               int K, int splits, void *stream)
{
    LDM_REQUIRE(a && b && out, "ldm_gemm_tn_f32: null pointer");
    LDM_REQUIRE(M > 0 && N > 0 && K > 0 && N % BT == 0 && K % BT == 0, "ldm_gemm_tn_f32: N=%d and K=%d must be multiples of 128", N, K);
    // rows per split: M / splits rounded up to a multiple of 32; the last split takes what is left (at least 32 rows)
    const int ms_rows = (int)((((long long)M + splits - 1) / (splits > 0 ? splits : 1) + BR - 1) / BR * BR);
    LDM_REQUIRE(splits >= 1 && M % BR == 0 && (long long)(splits - 1) * ms_rows < M, "ldm_gemm_tn_f32: M=%d must be a multiple of 32 and give each of the %d splits at least 32 rows",
                M, splits);
    LDM_REQUIRE(lda >= N && ldb >= K && lda % 4 == 0 && ldb % 4 == 0 && ldm_aligned16(a) && ldm_aligned16(b) && (((size_t)out) & 7) == 0 &&
                    (((size_t)colsum_a) & 7) == 0,
                "ldm_gemm_tn_f32: operands must be 16-byte addressable (lda=%lld ldb=%lld)", lda, ldb);
    TnP p{};
    p.a = a; p.b = b; p.out = out; p.colsum = colsum_a; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.ms = ms_rows; p.splits = splits;
    p.ntn = N / BT; p.ntk = K / BT;
    const long long blocks = (long long)p.ntn * p.ntk * splits;
    LDM_REQUIRE(blocks <= 0x7fffffffLL, "ldm_gemm_tn_f32: grid too large");
    constexpr size_t smem = 2ull * STAGE * sizeof(float);
    static LdmLdsOptIn opt_in;
    (void)opt_in((const void *)gemm_tn_kernel<false>, smem);
    void *rec = ldm_prof_begin(LDM_PROF_GEMM_TN, 2.0 * M * (double)N * K, (hipStream_t)stream,
                               4.0 * M * ((double)N + K) + 4.0 * N * (double)K * splits);
    ldm_launch(gemm_tn_kernel<false>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, p);
    ldm_prof_end(rec, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_gemm_tn_f32");
    return LDM_OK;
}

// Weight gradient of a dense 3x3 conv (zero pad 1: vae.py:57-58 and its autograd) WITHOUT the im2col matrix:
//   out[s][n][tap * Cin + ci] = sum over the pixels m of split s of dy[m][n] * x[pixel m shifted by tap][ci]
// dy [M = B*H*W, Cout] (row stride lda), x [M, Cin] rows; out is [splits][Npad][Kpad] with Npad = ldm_conv3x3_wgrad_npad(Cout) and Kpad = 9 * Cin
// rounded up to 128 (rows >= Cout and columns >= 9 * Cin come out as zeros); the caller sums the split planes (ldm_reduce_partials_f32) and
// takes the [Cout][9 * Cin] corner.  colsum_dy: optional [splits][Npad] column sums of dy (the bias gradient).  Cin % 4 == 0.
// rows of the output planes of ldm_conv3x3_wgrad_f32: Cout rounded up to the tile height the kernel picks (32, 64, else multiples of 128)
extern "C" int ldm_conv3x3_wgrad_npad(int Cout)
{
    const int bn = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : BT);
    return (Cout + bn - 1) / bn * bn;
}

extern "C" int ldm_conv3x3_wgrad_f32(const float *dy, long long lda, const float *x, float *out, float *colsum_dy, int B, int H, int W, int Cin, int Cout,
                                     int splits, void *stream)
{
    LDM_REQUIRE(dy && x && out, "ldm_conv3x3_wgrad_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0 && Cout > 0 && Cout % 4 == 0 && lda >= Cout && lda % 4 == 0, "ldm_conv3x3_wgrad_f32: bad shape");
    const long long M = (long long)B * H * W;
    LDM_REQUIRE((long long)H * W <= (1ll << 24), "ldm_conv3x3_wgrad_f32: at most 2^24 pixels per image");
    const int ms_rows = (int)(((M + splits - 1) / (splits > 0 ? splits : 1) + BR - 1) / BR * BR);
    LDM_REQUIRE(M < (1ll << 31) && splits >= 1 && M % BR == 0 && (long long)(splits - 1) * ms_rows < M,
                "ldm_conv3x3_wgrad_f32: B*H*W=%lld must be a multiple of 32 and give each of the %d splits at least 32 pixels", M, splits);
    LDM_REQUIRE(ldm_aligned16(dy) && ldm_aligned16(x) && (((size_t)out) & 7) == 0 && (((size_t)colsum_dy) & 7) == 0, "ldm_conv3x3_wgrad_f32: unaligned pointer");
    TnP p{};
    const int bn = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : BT);                     // output rows per tile (ldm_conv3x3_wgrad_npad)
    const int Np = (Cout + bn - 1) / bn * bn, Kp = (9 * Cin + BT - 1) / BT * BT;
    p.a = dy; p.b = x; p.out = out; p.colsum = colsum_dy; p.lda = lda; p.ldb = Cin; p.M = (int)M; p.N = Np; p.K = Kp; p.ms = ms_rows; p.splits = splits;
    p.ntn = Np / bn; p.ntk = Kp / BT;
    p.H = H; p.W = W; p.Cin = Cin; p.Nreal = Cout; p.inv_w = 1.0f / (float)W;
    const long long blocks = (long long)p.ntn * p.ntk * splits;
    LDM_REQUIRE(blocks <= 0x7fffffffLL, "ldm_conv3x3_wgrad_f32: grid too large");
    constexpr size_t smem = 2ull * STAGE * sizeof(float);
    static LdmLdsOptIn opt_in;
    (void)opt_in((const void *)gemm_tn_kernel<true>, smem);
    void *rec = ldm_prof_begin(LDM_PROF_GEMM_TN, 2.0 * M * (double)Cout * 9.0 * Cin, (hipStream_t)stream,
                               4.0 * M * ((double)Cout + Cin) + 4.0 * Np * (double)Kp * splits);
    if (bn == 64) ldm_launch(gemm_tn_narrow_kernel<true, 64>, dim3((unsigned)blocks), dim3(256), 2ull * (BR * 64 + BR * BT) * sizeof(float), (hipStream_t)stream, p);
    else if (bn == 32) ldm_launch(gemm_tn_narrow_kernel<true, 32>, dim3((unsigned)blocks), dim3(256), 2ull * (BR * 32 + BR * BT) * sizeof(float), (hipStream_t)stream, p);
    else ldm_launch(gemm_tn_kernel<true>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, p);
    ldm_prof_end(rec, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_conv3x3_wgrad_f32");
    return LDM_OK;
}
