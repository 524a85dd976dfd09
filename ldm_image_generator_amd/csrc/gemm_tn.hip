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
    const int nsteps = p.ms / BR;

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
                const int xx = (int)(m % p.W), yy = (int)((m / p.W) % p.H);
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

}  // namespace

extern "C" int ldm_gemm_tn_f32(const float *a, long long lda, const float *b, long long ldb, float *out, float *colsum_a, int M, int N,
                               int K, int splits, void *stream)
{
    LDM_REQUIRE(a && b && out, "ldm_gemm_tn_f32: null pointer");
    LDM_REQUIRE(M > 0 && N > 0 && K > 0 && N % BT == 0 && K % BT == 0, "ldm_gemm_tn_f32: N=%d and K=%d must be multiples of 128", N, K);
    LDM_REQUIRE(splits >= 1 && M % splits == 0 && (M / splits) % BR == 0, "ldm_gemm_tn_f32: M=%d must split into %d runs of a multiple of 32 rows",
                M, splits);
    LDM_REQUIRE(lda >= N && ldb >= K && lda % 4 == 0 && ldb % 4 == 0 && ldm_aligned16(a) && ldm_aligned16(b) && (((size_t)out) & 7) == 0 &&
                    (((size_t)colsum_a) & 7) == 0,
                "ldm_gemm_tn_f32: operands must be 16-byte addressable (lda=%lld ldb=%lld)", lda, ldb);
    TnP p{};
    p.a = a; p.b = b; p.out = out; p.colsum = colsum_a; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.ms = M / splits; p.splits = splits;
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
// dy [M = B*H*W, Cout] (row stride lda), x [M, Cin] rows; out is [splits][Npad][Kpad] with Npad = Cout rounded up to 128 and Kpad = 9 * Cin
// rounded up to 128 (rows >= Cout and columns >= 9 * Cin come out as zeros); the caller sums the split planes (ldm_reduce_partials_f32) and
// takes the [Cout][9 * Cin] corner.  colsum_dy: optional [splits][Npad] column sums of dy (the bias gradient).  Cin % 4 == 0.
extern "C" int ldm_conv3x3_wgrad_f32(const float *dy, long long lda, const float *x, float *out, float *colsum_dy, int B, int H, int W, int Cin, int Cout,
                                     int splits, void *stream)
{
    LDM_REQUIRE(dy && x && out, "ldm_conv3x3_wgrad_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0 && Cout > 0 && Cout % 4 == 0 && lda >= Cout && lda % 4 == 0, "ldm_conv3x3_wgrad_f32: bad shape");
    const long long M = (long long)B * H * W;
    LDM_REQUIRE(M < (1ll << 31) && splits >= 1 && M % splits == 0 && (M / splits) % BR == 0,
                "ldm_conv3x3_wgrad_f32: B*H*W=%lld must split into %d runs of a multiple of 32 pixels", M, splits);
    LDM_REQUIRE(ldm_aligned16(dy) && ldm_aligned16(x) && (((size_t)out) & 7) == 0 && (((size_t)colsum_dy) & 7) == 0, "ldm_conv3x3_wgrad_f32: unaligned pointer");
    TnP p{};
    const int Np = (Cout + BT - 1) / BT * BT, Kp = (9 * Cin + BT - 1) / BT * BT;
    p.a = dy; p.b = x; p.out = out; p.colsum = colsum_dy; p.lda = lda; p.ldb = Cin; p.M = (int)M; p.N = Np; p.K = Kp; p.ms = (int)(M / splits); p.splits = splits;
    p.ntn = Np / BT; p.ntk = Kp / BT;
    p.H = H; p.W = W; p.Cin = Cin; p.Nreal = Cout;
    const long long blocks = (long long)p.ntn * p.ntk * splits;
    LDM_REQUIRE(blocks <= 0x7fffffffLL, "ldm_conv3x3_wgrad_f32: grid too large");
    constexpr size_t smem = 2ull * STAGE * sizeof(float);
    static LdmLdsOptIn opt_in;
    (void)opt_in((const void *)gemm_tn_kernel<true>, smem);
    void *rec = ldm_prof_begin(LDM_PROF_GEMM_TN, 2.0 * M * (double)Cout * 9.0 * Cin, (hipStream_t)stream,
                               4.0 * M * ((double)Cout + Cin) + 4.0 * Np * (double)Kp * splits);
    ldm_launch(gemm_tn_kernel<true>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, p);
    ldm_prof_end(rec, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_conv3x3_wgrad_f32");
    return LDM_OK;
}
