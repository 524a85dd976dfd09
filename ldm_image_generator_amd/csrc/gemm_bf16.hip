// bf16-operand GEMMs of the training step (BASELINE cfg 5: "bf16 ... fwd+bwd HIP kernels"; the reference itself trains under
// reduced-precision autocast, train_ldm.py:68,80).  Operands are rounded ONCE to bf16 by their producers, products are
// accumulated in fp32 on v_mfma_f32_32x32x16_bf16, master weights / gradients / AdamW stay fp32.
//
//   ldm_gemm_bf16     out = act(A . W^T + bias) (+ addend)     -- the persistent LDS-DMA stream kernel of gemm_stream.hip with ET = 1
//   ldm_gemm_tn_bf16  out[s] = A_s^T . B_s  (weight gradients)  -- this file: operands as they lie in memory, MFMA fragments by
//                                                                  ds_read_b64_tr_b16 (the LDS transposing read of gfx950)
#include "gemm_common.h"

using namespace ldmgemm;

int ldm_gemm_stream_dispatch_bf16(const GemmP &p, int groups, bool out_bf16, hipStream_t st, bool gate, int amode);
int ldm_gemm_ring_dispatch_bf16(const GemmP &p, int groups, bool out_bf16, hipStream_t st);      // gemm_ring.hip: 256-row tiles, one workgroup per CU, four-stage ring
int ldm_gemm_ring_dispatch_bf16_gate(const GemmP &p, int groups, hipStream_t st);

namespace {

// mode 0: plain;  1: gate forward (d->w2 / bias2 = the "b" branch; x1 = a_pre out, x2 = b_pre out, both optional together);
// 2: gate backward fused behind the GEMM that produces dh (x1 = a_pre in, x2 = b_pre in, x3 = db out; d->out = da)
int gemm_bf16_impl(const char *who, const ldm_gemm_desc *d, int out_bf16, int mode, void *x1, void *x2, void *x3, void *stream)
{
    LDM_REQUIRE(d != nullptr && d->a && d->out, "%s: null descriptor / operand", who);
    LDM_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->N % 64 == 0 && d->K % 64 == 0, "%s: M=%d, N=%d and K=%d must be positive multiples of (1, 64, 64)", who,
                d->M, d->N, d->K);
    const bool conv = d->a_mode == LDM_A_CONV3X3;        // dense 3x3 (VAE decode under autocast): implicit im2col of bf16 rows [M, Cin]
    LDM_REQUIRE((d->a_mode == LDM_A_ROWS || (conv && mode == 0)) && d->o_mode == LDM_O_ROWS && !d->w_table, "%s: rows (or 3x3 taps) in / rows out, no pointer table", who);
    if (conv) {
        LDM_REQUIRE(d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % 64 == 0 && d->K == 9 * d->Cin && d->M % (d->H * d->W) == 0 && out_bf16,
                    "%s: conv3x3 needs K == 9*Cin, Cin %% 64 == 0, M %% (H*W) == 0 and the bf16 output", who);
    }
    LDM_REQUIRE((d->act == LDM_ACT_GATE) == (mode == 1), "%s: act=%d does not fit this entry point", who, d->act);
    LDM_REQUIRE(d->nseg >= 1 && d->nseg <= LDM_MAX_SEG && (d->seg_mode == LDM_SEG_N || d->seg_mode == LDM_SEG_K), "%s: nseg=%d seg_mode=%d", who, d->nseg,
                d->seg_mode);
    LDM_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0 && ldm_aligned16(d->a), "%s: rows of A and W must be 16-byte addressable (lda, ldw multiples of 8)", who);
    const int groups = d->groups > 0 ? d->groups : 1;
    LDM_REQUIRE(d->a_gstride % 8 == 0 && d->w_gstride % 8 == 0, "%s: group strides must be multiples of 8 elements", who);
    const int seg_total = d->seg_mode == LDM_SEG_N ? d->N : d->K;
    const int seg_len = d->nseg == 1 ? seg_total : d->seg_len;
    LDM_REQUIRE(seg_len > 0 && seg_len % 64 == 0 && (long long)seg_len * d->nseg == seg_total, "%s: segments (%d x %d) do not cover %d", who, d->nseg, seg_len,
                seg_total);
    LDM_REQUIRE(!(mode == 1 && d->seg_mode == LDM_SEG_K && d->nseg > 1), "%s: GATE with K-segments is not supported", who);
    for (int s = 0; s < d->nseg; ++s) {
        LDM_REQUIRE(d->w[s] && ldm_aligned16(d->w[s]), "%s: weight segment %d null/unaligned", who, s);
        if (mode == 1) LDM_REQUIRE(d->w2[s] && ldm_aligned16(d->w2[s]), "%s: gate weight segment %d null/unaligned", who, s);
    }
    // an addend has the output's element type: fp32 [M, ldadd] with the fp32 output, bf16 [M, ldadd] with the bf16 output
    LDM_REQUIRE(!(out_bf16 && d->addend && mode), "%s: the fused gate forms take no addend", who);
    if (out_bf16 && d->addend) LDM_REQUIRE((((size_t)d->addend) & 7) == 0 && d->ldadd % 4 == 0, "%s: bf16 addend must be 8-byte addressable", who);
    if (out_bf16) LDM_REQUIRE((((size_t)d->out) & 7) == 0 && d->ldo % 4 == 0 && d->o_gstride % 4 == 0, "%s: bf16 output must be 8-byte addressable", who);
    if (mode) {
        LDM_REQUIRE(out_bf16 && groups == 1, "%s: the fused gate forms write bf16 and take one group", who);
        LDM_REQUIRE((x1 == nullptr) == (x2 == nullptr) && (mode == 1 || (x1 && x3)), "%s: missing pre-activation / output buffer", who);
        LDM_REQUIRE((((size_t)x1 | (size_t)x2 | (size_t)x3) & 7) == 0, "%s: bf16 side buffers must be 8-byte addressable", who);
    }
    GemmP p{};
    p.a = d->a; p.lda = d->lda / 2; p.M = d->M; p.N = d->N; p.K = d->K / 2;          // K axis in 4-byte units from here on
    p.cpt = 1;
    if (conv) { p.H = d->H; p.W = d->W; p.Cin = d->Cin / 2; p.cpt = d->Cin / 64; }      // a K-step is 64 bf16 channels of one tap
    p.nseg = d->nseg; p.seg_mode = d->seg_mode;
    p.seg_len = d->seg_mode == LDM_SEG_K ? seg_len / 2 : seg_len;
    for (int s = 0; s < LDM_MAX_SEG; ++s) {
        p.w[s] = s < d->nseg ? d->w[s] : nullptr;
        p.bias[s] = s < d->nseg ? d->bias[s] : nullptr;
        p.w2[s] = (mode == 1 && s < d->nseg) ? d->w2[s] : nullptr;
        p.bias2[s] = (mode == 1 && s < d->nseg) ? d->bias2[s] : nullptr;
    }
    p.ldw = d->ldw / 2; p.act = d->act; p.slope = d->slope;
    p.addend = out_bf16 ? nullptr : d->addend; p.addend16 = out_bf16 ? (const void *)d->addend : nullptr;
    p.ldadd = d->ldadd; p.out = d->out; p.ldo = d->ldo;
    p.o_mode = LDM_O_ROWS;
    p.a_gstride = d->a_gstride / 2; p.w_gstride = d->w_gstride / 2; p.o_gstride = d->o_gstride; p.b_gstride = d->b_gstride;
    p.wide_ok = out_bf16 ? 1
                         : (ldm_aligned16(d->out) && d->ldo % 4 == 0 && d->o_gstride % 4 == 0 && (!d->addend || (ldm_aligned16(d->addend) && d->ldadd % 4 == 0)));
    if (mode == 1) { p.out2 = x1; p.out3 = x2; }
    if (mode == 2) { p.in2 = x1; p.in3 = x2; p.out2 = x3; }
    hipStream_t st = (hipStream_t)stream;
    const double mn = (double)d->M * d->N * groups;
    const double out_planes = mode == 1 ? (x1 ? 3.0 : 1.0) : (mode == 2 ? 2.0 : 1.0);
    const double a_cols = conv ? d->Cin : d->K;                 // algorithmic bytes: every operand element once (im2col re-reads are not algorithmic)
    void *rec = ldm_prof_begin(LDM_PROF_GEMM_BF16, 2.0 * mn * d->K * (mode == 1 ? 2.0 : 1.0), st,
                               2.0 * ((double)d->M * a_cols * (d->a_gstride || groups == 1 ? groups : 1) + (double)d->N * d->K * groups * (mode == 1 ? 2.0 : 1.0)) +
                                   mn * (out_bf16 ? 2.0 : 4.0) * out_planes + (mode == 2 ? mn * 4.0 : 0.0) + (d->addend ? mn * (out_bf16 ? 2.0 : 4.0) : 0.0));
    const bool ring_ok = !conv && !p.addend16;
    const int ok = (ring_ok && mode == 0 && ldm_gemm_ring_dispatch_bf16(p, groups, out_bf16 != 0, st)) || (ring_ok && mode == 1 && ldm_gemm_ring_dispatch_bf16_gate(p, groups, st)) ||
                   ldm_gemm_stream_dispatch_bf16(p, groups, out_bf16 != 0, st, mode == 1, d->a_mode);
    ldm_prof_end(rec, st);
    LDM_REQUIRE(ok, "%s: no kernel instance for this shape (N=%d, seg_len=%d)", who, d->N, seg_len);
    LDM_CHECK_LAUNCH(who);
    return LDM_OK;
}

}  // namespace

extern "C" int ldm_gemm_bf16(const ldm_gemm_desc *d, int out_bf16, void *stream)
{
    return gemm_bf16_impl("ldm_gemm_bf16", d, out_bf16, 0, nullptr, nullptr, nullptr, stream);
}

extern "C" int ldm_gemm_bf16_gate_fwd(const ldm_gemm_desc *d, void *a_pre, void *b_pre, void *stream)
{
    return gemm_bf16_impl("ldm_gemm_bf16_gate_fwd", d, 1, 1, a_pre, b_pre, nullptr, stream);
}

extern "C" int ldm_gemm_bf16_gate_bwd(const ldm_gemm_desc *d, const void *a_pre, const void *b_pre, void *db, void *stream)
{
    return gemm_bf16_impl("ldm_gemm_bf16_gate_bwd", d, 1, 2, (void *)a_pre, (void *)b_pre, db, stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// TN: out[s][n][k] = sum over rows m of split s of A[m][n] * B[m][k]   (dW = dY^T X, autograd of every 1x1 conv / Linear)
//
// Output tile 128 (n) x 128 KT (k) per workgroup, KT = 1 | 2; 4 waves of 64 x 64 KT.  A stage = 32 contraction rows of both
// operands ([32][128] bf16 sub-tiles with 256-byte LDS rows), filled by LDS-DMA as the rows lie in memory, into a ring of
// NS = 4 (KT = 1, 16 KB stages) or 3 (KT = 2, 24 KB) stages: the DMA runs NS - 1 stages ahead behind COUNTED vmcnt waits and
// one raw barrier per stage (the 16 ... 32 MFMAs of a stage are far shorter than an HBM round trip, so a one-stage lookahead
// left the waves parked 60-70 % of the time).  The MFMA wants, per lane, 8 consecutive m of ONE column: two
// ds_read_b64_tr_b16 (each hands a lane 4 rows of its column out of a 4 x 16 block read by 16 lanes).  Chunk swizzle on the
// DMA source side and on the reads: physical chunk = logical ^ (((row & 3) << 2) | ((row >> 2) & 3)) -- conflict-free for the
// transposed 32x32x16 operand reads.  Split over M on the grid; partial planes summed by the caller in a fixed order.
// ------------------------------------------------------------------------------------------------------------------------
namespace {

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

struct TnP16 {
    const unsigned short *a, *b;
    float *out, *colsum;
    long long lda, ldb;
    int M, N, K, ms, splits, ntn, ntk;
};

constexpr int TBT = 128;             // sub-tile edge (columns of one [32][128] LDS image)
constexpr int TBR = 32;              // contraction rows per stage
constexpr int TSUB = TBR * TBT;      // bf16 elements of one sub-tile image

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }

// LDS-DMA from inline asm (16 bytes per lane, destination = wave-uniform LDS byte address + lane * 16).  M0 carries the
// destination and is compiler-reserved: saved and restored inside the one statement that uses it.
// Source = wave-uniform base (SGPR pair) + this lane's 32-bit byte offset, a loop invariant: no 64-bit vector address arithmetic per
// instruction (the first form took the whole address in a VGPR pair: a 64-bit multiply-add per DMA instruction and lane).
__device__ __forceinline__ void glds16_asm(const char *base, int off, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void tn_wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

template <int KT>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(const TnP16 p)
{
    constexpr int NS = KT == 1 ? 4 : 3;                 // ring stages
    constexpr int LOOK = NS - 1;                        // stages in flight
    constexpr int PW = 2 + 2 * KT;                      // LDS-DMA instructions per wave per stage
    constexpr int STAGE = (1 + KT) * TSUB;              // bf16 elements per stage: A sub-tile, then KT B sub-tiles
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int T = p.ntn * p.ntk;
    int tile, split;
    if (p.splits % 8 == 0) {               // tiles of one split on one XCD (they read the same operand rows): see gemm_tn.hip
        const int b = (int)blockIdx.x, blk = b / (8 * T), in = b - blk * 8 * T;
        split = blk * 8 + (in & 7);
        tile = in >> 3;
    } else {
        tile = (int)blockIdx.x % T;
        split = (int)blockIdx.x / T;
    }
    const int n0 = (tile / p.ntk) * TBT, k0 = (tile % p.ntk) * TBT * KT;
    const long long row0 = (long long)split * p.ms;
    const int nsteps = p.ms / TBR;

    // DMA: one instruction = 64 lanes x 16 B = 4 rows of one sub-tile; a wave moves rows 4 (4 i + wave) .. + 3, i < 2, of every sub-tile.
    // Issued from inline asm: hipcc then does not know that LDS is written behind its back and leaves the counted vmcnt
    // waits below alone (with the builtin it put s_waitcnt vmcnt(0) in front of the first transposing read of every stage).
    const int drow = lane >> 4, dchunk = lane & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short *)lds16;
    int a_voff[2], b_voff[2];                                         // byte offsets of this lane's chunks inside a stage's rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (4 * i + wave) + drow;
        const int csrc = (dchunk ^ tn_swz(row)) * 8;
        a_voff[i] = (int)((row * p.lda + n0 + csrc) * 2);
        b_voff[i] = (int)((row * p.ldb + k0 + csrc) * 2);
    }
    auto issue = [&](int step) {
        const unsigned dst0 = lds_base + (unsigned)(((step % NS) * STAGE + wave * 512) * 2);       // bytes; + i * 4096 + subtile * 8192
        const long long mbase = row0 + (long long)step * TBR;
        const char *abase = (const char *)(p.a + mbase * p.lda), *bbase = (const char *)(p.b + mbase * p.ldb);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16_asm(abase, a_voff[i], __builtin_amdgcn_readfirstlane(dst0 + i * 4096));
#pragma unroll
            for (int j = 0; j < KT; ++j)
                glds16_asm(bbase + j * TBT * 2, b_voff[i], __builtin_amdgcn_readfirstlane(dst0 + i * 4096 + (1 + j) * TSUB * 2));
        }
    };

    f32x16 acc[2][2 * KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2 * KT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transposed fragment read.  Lane l: group g = l >> 4 (0..3), q = (l >> 2) & 3, pp = l & 3.  The operand lane (r = l & 31,
    // hh = l >> 5) needs rows m = mb + 8 hh + {0..7} of column c0 + r: group g covers columns c0 + 16 (g & 1) .. + 15 and rows
    // mb + 8 (g >> 1) + {0..3} (first read) / + {4..7} (second read).  The address a lane SUPPLIES is row q of that block, columns 4 pp .. 4 pp + 3.
    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto frag = [&](const unsigned short *base, int mb, int c0) -> s16x8 {
        const int col = c0 + 16 * (grp & 1) + 4 * pp;
        const int rowa = mb + 8 * (grp >> 1) + q, rowb = rowa + 4;
        const unsigned short *pa = base + rowa * TBT + (((col >> 3) ^ tn_swz(rowa)) << 3) + (col & 7);
        const unsigned short *pb = base + rowb * TBT + (((col >> 3) ^ tn_swz(rowb)) << 3) + (col & 7);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pa);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pb);
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    const bool want_cs = p.colsum != nullptr && k0 == 0 && wn == 0;      // one workgroup column and one wave column own each A column
    float cs[2] = {0.f, 0.f};            // column sums of A: this lane's column of each of its two A tiles, its 8 rows per 16-row slice
#pragma unroll
    for (int s0 = 0; s0 < LOOK; ++s0)
        if (s0 < nsteps) issue(s0);
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        // this wave's pieces of stage `step` have landed once at most the younger stages' instructions are outstanding
        const int younger = nsteps - 1 - step < LOOK - 1 ? nsteps - 1 - step : LOOK - 1;
        if (younger >= 2) tn_wait_vmcnt<2 * PW>();
        else if (younger == 1) tn_wait_vmcnt<PW>();
        else tn_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // ... and its reads of the slot about to be refilled are complete
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + LOOK < nsteps) issue(step + LOOK);                  // into the slot every wave finished reading before this barrier
        const unsigned short *As = lds16 + (step % NS) * STAGE;
#pragma unroll
        for (int s = 0; s < TBR / 16; ++s) {
            s16x8 a[2], b[2 * KT];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag(As, 16 * s, wm * 64 + 32 * i);
#pragma unroll
            for (int j = 0; j < 2 * KT; ++j) {
                const int c = wn * 64 * KT + 32 * j;                    // column inside the workgroup's K range
                b[j] = frag(As + (1 + c / TBT) * TSUB, 16 * s, c % TBT);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2 * KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (want_cs) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[i] += bf16_lo((unsigned)(unsigned short)a[i][e]);
            }
        }
    }

    const int r = lane & 31, h = lane >> 5;
    if (want_cs) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float s_ = cs[i] + __shfl_xor(cs[i], 32);
            if (h == 0) p.colsum[(long long)split * p.N + n0 + wm * 64 + 32 * i + r] = s_;
        }
    }
    // C/D map: column = lane & 31 (k index), row = (e & 3) + 8 (e >> 2) + 4 h (n index)
    float *obase = p.out + (long long)split * p.N * p.K + (long long)(n0 + wm * 64) * p.K + k0 + wn * 64 * KT + r;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < 2 * KT; ++j) obase[(long long)row * p.K + 32 * j] = acc[i][j][e];
        }
}


// ---- 256 x 256 tiles, ONE workgroup of eight waves per CU ("TN ring") -------------------------------------------------------------------
// The 128 x 256 kernel above moves 24.6 KB of operands per stage for 2.1 MFLOP (85 FLOP per staged byte): at its 640-680 TFLOP/s it
// pulls 7.5 TB/s through L2 -- every A panel is re-read by K / 256 tiles and every B panel by N / 128.  A 256 x 256 tile halves the staged
// bytes per FLOP (170 FLOP / B): eight waves as 2 (n) x 4 (k), each 128 x 64 (the same eight accumulator tiles as above), a stage =
// 32 contraction rows of two A sub-tiles and two B sub-tiles (32 KB), FOUR stages (three in flight behind counted vmcnt waits, one raw
// barrier per stage), four LDS-DMA instructions per wave and stage.  Same sub-tile images, swizzle and transposing reads; the k order
// of every output element is that of the kernel above -> bit-identical results.
__global__ __launch_bounds__(512, 1) void gemm_tn_bf16_ring_kernel(const TnP16 p)
{
    constexpr int NS = 4, LOOK = NS - 1, PW = 4;
    constexpr int STAGE = 4 * TSUB;                      // bf16 elements: A sub-tiles 0, 1, then B sub-tiles 0, 1
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int T = p.ntn * p.ntk;
    int tile, split;
    if (p.splits % 8 == 0) {               // tiles of one split on one XCD (they read the same operand rows)
        const int b = (int)blockIdx.x, blk = b / (8 * T), in = b - blk * 8 * T;
        split = blk * 8 + (in & 7);
        tile = in >> 3;
    } else {
        tile = (int)blockIdx.x % T;
        split = (int)blockIdx.x / T;
    }
    const int n0 = (tile / p.ntk) * 256, k0 = (tile % p.ntk) * 256;
    const long long row0 = (long long)split * p.ms;
    const int nsteps = p.ms / TBR;

    // DMA: one instruction = 4 rows of one sub-tile; a sub-tile's 32 rows are 8 instructions, the stage's 4 sub-tiles 32: wave w moves
    // instruction pieces 4 w .. 4 w + 3 (sub-tile = piece >> 3, rows 4 (piece & 7) .. + 3)
    const int drow = lane >> 4, dchunk = lane & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short *)lds16;
    int voff[PW];                                                     // byte offset of this lane's chunk from the stage's first row of its operand
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int piece = 4 * wave + i;
        const int sub = piece >> 3, row = 4 * (piece & 7) + drow;
        const int csrc = (dchunk ^ tn_swz(row)) * 8;
        voff[i] = (int)((sub < 2 ? row * p.lda + n0 + sub * TBT + csrc : row * p.ldb + k0 + (sub - 2) * TBT + csrc) * 2);
    }
    auto issue = [&](int step) {
        const long long mbase = row0 + (long long)step * TBR;
        const char *abase = (const char *)(p.a + mbase * p.lda), *bbase = (const char *)(p.b + mbase * p.ldb);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = 4 * wave + i;
            const int sub = piece >> 3;                               // wave-uniform: waves 0-3 move A, 4-7 move B
            glds16_asm(sub < 2 ? abase : bbase, voff[i], __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(((step % NS) * STAGE + sub * TSUB + 4 * (piece & 7) * TBT) * 2)));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto frag = [&](const unsigned short *base, int mb, int c0) -> s16x8 {
        const int col = c0 + 16 * (grp & 1) + 4 * pp;
        const int rowa = mb + 8 * (grp >> 1) + q, rowb = rowa + 4;
        const unsigned short *pa = base + rowa * TBT + (((col >> 3) ^ tn_swz(rowa)) << 3) + (col & 7);
        const unsigned short *pb = base + rowb * TBT + (((col >> 3) ^ tn_swz(rowb)) << 3) + (col & 7);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pa);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pb);
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    const bool want_cs = p.colsum != nullptr && k0 == 0 && wn == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < LOOK; ++s0)
        if (s0 < nsteps) issue(s0);
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        const int younger = nsteps - 1 - step < LOOK - 1 ? nsteps - 1 - step : LOOK - 1;
        if (younger >= 2) tn_wait_vmcnt<2 * PW>();
        else if (younger == 1) tn_wait_vmcnt<PW>();
        else tn_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + LOOK < nsteps) issue(step + LOOK);
        const unsigned short *St = lds16 + (step % NS) * STAGE;
#pragma unroll
        for (int s = 0; s < TBR / 16; ++s) {
            s16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag(St + wm * TSUB, 16 * s, 32 * i);                // A sub-tile wm: this wave's 128 n-columns
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wn * 64 + 32 * j;                                                  // column inside the tile's 256 k-columns
                b[j] = frag(St + (2 + c / TBT) * TSUB, 16 * s, c % TBT);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (want_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[i] += bf16_lo((unsigned)(unsigned short)a[i][e]);
            }
        }
    }

    const int r = lane & 31, h = lane >> 5;
    if (want_cs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s_ = cs[i] + __shfl_xor(cs[i], 32);
            if (h == 0) p.colsum[(long long)split * p.N + n0 + wm * 128 + 32 * i + r] = s_;
        }
    }
    float *obase = p.out + (long long)split * p.N * p.K + (long long)(n0 + wm * 128) * p.K + k0 + wn * 64 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) obase[(long long)row * p.K + 32 * j] = acc[i][j][e];
        }
}

int g_tn_ring = 1;        // 1 (default): 256 x 256 tiles where N, K allow and the grid fills the CUs; 0: the 128-row kernel everywhere (bit-identical)

template <int KT>
void tn_launch(const TnP16 &p, long long blocks, hipStream_t st)
{
    constexpr int NS = KT == 1 ? 4 : 3;
    constexpr size_t smem = (size_t)NS * (1 + KT) * TSUB * sizeof(unsigned short);
    static LdmLdsOptIn opt_in;
    (void)opt_in((const void *)gemm_tn_bf16_kernel<KT>, smem);
    ldm_launch(gemm_tn_bf16_kernel<KT>, dim3((unsigned)blocks), dim3(256), smem, st, p);
}

}  // namespace

extern "C" int ldm_gemm_tn_bf16(const void *a, long long lda, const void *b, long long ldb, float *out, float *colsum_a, int M, int N, int K, int splits,
                                void *stream)
{
    LDM_REQUIRE(a && b && out, "ldm_gemm_tn_bf16: null pointer");
    LDM_REQUIRE(M > 0 && N > 0 && K > 0 && N % TBT == 0 && K % TBT == 0, "ldm_gemm_tn_bf16: N=%d and K=%d must be multiples of 128", N, K);
    LDM_REQUIRE(splits >= 1 && M % splits == 0 && (M / splits) % 64 == 0, "ldm_gemm_tn_bf16: M=%d must split into %d runs of a multiple of 64 rows", M, splits);
    LDM_REQUIRE(lda >= N && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && ldm_aligned16(a) && ldm_aligned16(b),
                "ldm_gemm_tn_bf16: operands must be 16-byte addressable (lda=%lld ldb=%lld)", lda, ldb);
    LDM_REQUIRE(lda < (1ll << 24) && ldb < (1ll << 24), "ldm_gemm_tn_bf16: row strides must stay below 2^24 elements (32-bit lane offsets)");
    TnP16 p{};
    p.a = (const unsigned short *)a; p.b = (const unsigned short *)b; p.out = out; p.colsum = colsum_a; p.lda = lda; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K; p.ms = M / splits; p.splits = splits; p.ntn = N / TBT;
    void *rec = ldm_prof_begin(LDM_PROF_TN_BF16, 2.0 * M * (double)N * K, (hipStream_t)stream, 2.0 * M * ((double)N + K) + 4.0 * N * (double)K * splits);
    // 256 x 256 tiles, one workgroup per CU, where N and K allow and the tiles x splits still give every CU a workgroup
    const long long ring_blocks = (long long)(N / 256) * (K / 256) * splits;
    static LdmLdsOptIn ring_opt;
    constexpr size_t ring_smem = 4ull * 4 * TSUB * sizeof(unsigned short);                      // 128 KiB
    if (g_tn_ring && N % 256 == 0 && K % 256 == 0 && ring_blocks >= (long long)ldm_cu_count() * 3 / 4 && ring_blocks <= 0x7fffffffLL &&
        ring_opt((const void *)gemm_tn_bf16_ring_kernel, ring_smem)) {
        p.ntn = N / 256; p.ntk = K / 256;
        ldm_launch(gemm_tn_bf16_ring_kernel, dim3((unsigned)ring_blocks), dim3(512), ring_smem, (hipStream_t)stream, p);
        ldm_prof_end(rec, (hipStream_t)stream);
        LDM_CHECK_LAUNCH("ldm_gemm_tn_bf16");
        return LDM_OK;
    }
    // 128 x 256 tiles where K allows and the grid stays full: a third less operand traffic per output, twice the MFMAs per barrier
    const bool wide = K % (2 * TBT) == 0 && (long long)(N / TBT) * (K / (2 * TBT)) * splits >= 256;     // at least one workgroup per CU
    p.ntk = wide ? K / (2 * TBT) : K / TBT;
    const long long blocks = (long long)p.ntn * p.ntk * splits;
    if (blocks > 0x7fffffffLL) { ldm_prof_end(rec, (hipStream_t)stream); ldm_set_error("ldm_gemm_tn_bf16: grid too large"); return LDM_EINVAL; }
    if (wide) tn_launch<2>(p, blocks, (hipStream_t)stream);
    else tn_launch<1>(p, blocks, (hipStream_t)stream);
    ldm_prof_end(rec, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_gemm_tn_bf16");
    return LDM_OK;
}

// kernel behind ldm_gemm_tn_bf16: 1 (default) = 256 x 256 tiles, one workgroup per CU, where N and K are multiples of 256 and tiles x splits fill
// the chip; 0 = the 128-row kernel everywhere.  Bit-identical results (A/B and test knob).  Returns the previous setting.
extern "C" int ldm_gemm_tn_ring(int v)
{
    const int old = g_tn_ring;
    if (v == 0 || v == 1) g_tn_ring = v;
    return old;
}
