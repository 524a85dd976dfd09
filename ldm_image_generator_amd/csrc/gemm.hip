// fp32 GEMM family on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, bit-for-bit an fmaf chain).
//
//     out = act(A . W^T + bias) (+ addend)
//
// One 256-thread workgroup (4 waves) owns a BM x BN output tile; each wave owns
// TM x TN accumulator tiles of 32x32.  K is walked in steps of 32 floats: both
// operands are K-contiguous ("NT"), so a tile row is one full 128-B line.
// Global -> registers -> LDS staging, two LDS stages, one barrier per K-step
// (loads for step k+1 are issued before the MFMAs of step k and written to LDS
// after them).  LDS rows are 128 B with a 16-B-chunk XOR swizzle
// (chunk ^= (row>>1)&7) that makes both the ds_write_b128 staging stores and the
// ds_read_b128 fragment reads bank-conflict free.  A fragment read hands each lane
// four consecutive k of its row; they feed four MFMA k-steps (the k order inside
// a 32-float step is a fixed permutation, identical for A and W).
//
// Variants (template): tile shape, GATE (two weight matrices per tile, epilogue
// a*relu(b): the ReGLU of modules.py:15), A addressing (rows | implicit 3x3
// im2col).  Runtime: weight segments selected by pointer (RandomMoE experts are
// never copied), bias/activation/addend, output scatter (rows | ConvTranspose
// 2x2 | nearest-x2 replicate), groups on grid.y.
#include "gemm_common.h"
#include <cstdlib>

using namespace ldmgemm;

namespace {

template <int WM, int WN, int TM, int TN, bool GATE, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmP p)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NB = GATE ? 2 * BN : BN;
    constexpr int A_F4 = BM / 32, B_F4 = NB / 32;
    constexpr int STAGE = (BM + NB) * 32;
    constexpr int NACC = GATE ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int g = blockIdx.y;
    const int ntn = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m, tile_n;
    tile_from_id(tile, (p.M + BM - 1) / BM, ntn, tile_m, tile_n);
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    const int chunk = t & 7, lrow = t >> 3;
    const int nk = p.K >> 5;

    // ---- per-thread A rows -------------------------------------------------
    const float *a_base = p.a + g * p.a_gstride + chunk * 4;
    long long a_off[A_F4];
    bool a_ok[A_F4];
    int a_y[A_F4], a_x[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.M;
        a_off[i] = (long long)m * p.lda;
        if (AMODE == LDM_A_CONV3X3) {
            a_x[i] = m % p.W;
            a_y[i] = (m / p.W) % p.H;
        }
    }
    // ---- per-thread W rows -------------------------------------------------
    // rows [0,BN) come from w, rows [BN,2BN) (GATE) from w2
    const int seg_n = (p.seg_mode == LDM_SEG_N) ? n0 / p.seg_len : 0;
    const int nloc0 = (p.seg_mode == LDM_SEG_N) ? n0 - seg_n * p.seg_len : n0;

    f32x4 areg[A_F4], breg[B_F4];

    auto load_tiles = [&](int kt) {
        if (AMODE == LDM_A_CONV3X3) {
            const int tap = kt / p.cpt;
            const int c0 = (kt - tap * p.cpt) << 5;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const long long shift = (long long)(dy * p.W + dx) * p.lda + c0;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const bool ok = a_ok[i] && (unsigned)(a_y[i] + dy) < (unsigned)p.H && (unsigned)(a_x[i] + dx) < (unsigned)p.W;
                areg[i] = ok ? *(const f32x4 *)(a_base + a_off[i] + shift) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_F4; ++i)
                areg[i] = a_ok[i] ? *(const f32x4 *)(a_base + a_off[i] + ((long long)kt << 5)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int seg = seg_n;
        long long kcol = (long long)kt << 5;
        if (p.seg_mode == LDM_SEG_K) {
            seg = (kt << 5) / p.seg_len;
            kcol -= (long long)seg * p.seg_len;
        }
        const float *wa = (p.use_table ? p.wtab[g] : p.w[seg] + g * p.w_gstride) + kcol + chunk * 4;
        const float *wb = GATE ? p.w2[seg] + g * p.w_gstride + kcol + chunk * 4 : nullptr;
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int row = lrow + 32 * i;
            if (GATE && row >= BN)
                breg[i] = *(const f32x4 *)(wb + (long long)(nloc0 + row - BN) * p.ldw);
            else
                breg[i] = *(const f32x4 *)(wa + (long long)(nloc0 + row) * p.ldw);
        }
    };
    auto store_tiles = [&](int stage) {
        float *As = lds + stage * STAGE, *Bs = As + BM * 32;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *(f32x4 *)(As + swz(lrow + 32 * i, chunk)) = areg[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) *(f32x4 *)(Bs + swz(lrow + 32 * i, chunk)) = breg[i];
    };

    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][i][j][e] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float *As = lds + cur * STAGE, *Bs = As + BM * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 2 * j + h;
            f32x4 af[TM], bf[NACC][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4 *)(As + swz((wm * TM + i) * 32 + r, c));
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int i = 0; i < TN; ++i) bf[q][i] = *(const f32x4 *)(Bs + swz(q * BN + (wn * TN + i) * 32 + r, c));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < NACC; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
                            acc[q][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[q][jj][e], acc[q][i][jj], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    float no_pre[TM][TN][16];
    EpiCols<TN> cols;
    gemm_epilogue_cols<WN, TN, GATE>(p, cols, n0, g, seg_n, wn, r);
    gemm_epilogue<WM, WN, TM, TN, GATE>(p, acc, m0, wm, h, cols, no_pre, false);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, bool GATE, int AMODE>
int launch(const GemmP &p, int groups, hipStream_t st)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NB = GATE ? 2 * BN : BN;
    constexpr size_t smem = 2ull * (BM + NB) * 32 * sizeof(float);
    static LdmLdsOptIn opt_in;
    auto kern = gemm_f32_kernel<WM, WN, TM, TN, GATE, AMODE>;
    (void)opt_in((const void *)kern, smem);
    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    dim3 grid(ntm * ntn, groups, 1);
    ldm_launch(kern, grid, dim3(256), smem, st, p);
    return 0;
}

template <bool GATE, int AMODE>
int dispatch(const GemmP &p, int groups, hipStream_t st)
{
    const int unit = (p.seg_mode == LDM_SEG_N) ? p.seg_len : p.N;     // BN must divide this
    if (GATE) {
        if (unit % 64 == 0) return launch<2, 2, 2, 1, GATE, AMODE>(p, groups, st);
        return launch<4, 1, 1, 1, GATE, AMODE>(p, groups, st);
    }
    if (p.M <= 32 && unit % 128 == 0) return launch<1, 4, 1, 1, GATE, AMODE>(p, groups, st);
    if (unit % 128 == 0) return launch<2, 2, 2, 2, GATE, AMODE>(p, groups, st);
    if (unit % 64 == 0) return launch<2, 2, 2, 1, GATE, AMODE>(p, groups, st);
    return launch<4, 1, 1, 1, GATE, AMODE>(p, groups, st);
}

int g_variant = 1;      // 0: tile-per-block kernel, 1: persistent LDS-DMA stream kernel, 2: stream kernel, bf16x3 split consumer

}  // namespace

int ldm_gemm_stream_dispatch(const ldmgemm::GemmP &p, int groups, bool gate, int amode, hipStream_t st, bool split);
int ldm_gemm_ring_dispatch_f32(const ldmgemm::GemmP &p, int groups, bool gate, int amode, hipStream_t st);      // gemm_ring.hip
int ldm_gconv3x3_dispatch(const ldmgemm::GemmP &p, int groups, bool gate, int amode, hipStream_t st);

namespace {

void launch_any(const GemmP &p, int groups, bool gate, int a_mode, hipStream_t st)
{
    if (g_variant >= 1 && ldm_gconv3x3_dispatch(p, groups, gate, a_mode, st)) return;
    if (g_variant == 1 && ldm_gemm_ring_dispatch_f32(p, groups, gate, a_mode, st)) return;       // large rows problems: 256-row tiles, one workgroup per CU
    if (g_variant >= 1 && ldm_gemm_stream_dispatch(p, groups, gate, a_mode, st, g_variant == 2)) return;
    if (gate)
        dispatch<true, LDM_A_ROWS>(p, groups, st);
    else if (a_mode == LDM_A_CONV3X3)
        dispatch<false, LDM_A_CONV3X3>(p, groups, st);
    else
        dispatch<false, LDM_A_ROWS>(p, groups, st);
}

// ---------------------------------------------------------------------------------------------------
// split-K for problems with few output tiles and a long reduction (small batch: M <= 128 rows against the
// deep stages' K = 512 ... 3072, the tiny-M Encodings GEMMs): one tile per CU would stream its whole K
// range serially at the per-CU fetch rate; instead the reduction is cut into S grid groups that write
// fp32 partial tiles into the caller's workspace, and a second kernel sums them in a FIXED order and
// applies bias / gate / activation / addend.  Deterministic; S depends only on (N, K), never on M.
// ---------------------------------------------------------------------------------------------------
struct SplitEpiP {
    const float *pa, *pb;            // partials [S][M][N] (pb: gate "b" half)
    int S, M, N, seg_mode, nseg, seg_len, act;
    float slope;
    const float *bias[LDM_MAX_SEG], *bias2[LDM_MAX_SEG];
    const float *addend;
    long long ldadd, ldo;
    float *out;
};

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const SplitEpiP q)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n4n = q.N >> 2;
    if (idx >= (long long)q.M * n4n) return;
    const int m = (int)(idx / n4n), n = (int)(idx - (long long)m * n4n) * 4;
    const long long plane = (long long)q.M * q.N;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < q.S; ++s) {
        const f32x4 v = *(const f32x4 *)(q.pa + s * plane + (long long)m * q.N + n);
        a += v;
        if (q.pb) b += *(const f32x4 *)(q.pb + s * plane + (long long)m * q.N + n);
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = n + e;
        float b1 = 0.f, b2 = 0.f;
        if (q.seg_mode == LDM_SEG_K) {
            for (int sg = 0; sg < q.nseg; ++sg)
                if (q.bias[sg]) b1 += q.bias[sg][col];
            // same association as the fused epilogue: ((b0 + b1) + b2) + b3
        } else {
            const int sg = col / q.seg_len, loc = col - sg * q.seg_len;
            if (q.bias[sg]) b1 = q.bias[sg][loc];
            if (q.pb && q.bias2[sg]) b2 = q.bias2[sg][loc];
        }
        float v = a[e] + b1;
        if (q.pb)
            v = v * fmaxf(b[e] + b2, 0.f);
        else if (q.act == LDM_ACT_RELU)
            v = fmaxf(v, 0.f);
        else if (q.act == LDM_ACT_LRELU)
            v = v > 0.f ? v : v * q.slope;
        if (q.addend) v += q.addend[(long long)m * q.ldadd + col];
        o[e] = v;
    }
    float *dst = q.out + (long long)m * q.ldo + n;
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
}

bool splitk_launch(const ldm_gemm_desc &d, const GemmP &p, bool gate, hipStream_t st)
{
    if (d.o_mode != LDM_O_ROWS || d.a_mode != LDM_A_ROWS || d.w_table || (d.groups > 1) || d.M > 128) return false;
    if (d.lda != d.K) return false;                                   // A columns must be the K axis, densely
    const int nk = d.K >> 5, ntn = d.N / 64;
    if (ntn < 1 || ntn > 96 || nk < 8) return false;
    const int nseg_k = (p.seg_mode == LDM_SEG_K) ? p.nseg : 1;
    const int steps_per_seg = nk / nseg_k;
    int s2 = 1;                                                       // splits inside one K-segment: power of two
    while (ntn * nseg_k * s2 * 2 <= 512 && steps_per_seg % (s2 * 2) == 0 && steps_per_seg / (s2 * 2) >= 2) s2 *= 2;
    const int S = nseg_k * s2;
    if (S < 2 || S > LDM_MAX_TABLE) return false;
    const size_t plane = (size_t)d.M * d.N * sizeof(float);
    if (plane * S * (gate ? 2 : 1) > (size_t)d.workspace_bytes) return false;
    const int ks = (d.K / nseg_k) / s2;

    auto partial = [&](const float *const *wsrc, float *dst) {
        GemmP q = p;
        q.K = ks; q.act = LDM_ACT_NONE; q.addend = nullptr; q.out = dst; q.ldo = d.N; q.o_mode = LDM_O_ROWS;
        q.a_gstride = ks; q.o_gstride = (long long)d.M * d.N; q.b_gstride = 0;
        q.wide_ok = ldm_aligned16(dst) && q.o_gstride % 4 == 0;
        for (int i = 0; i < LDM_MAX_SEG; ++i) { q.bias[i] = q.bias2[i] = nullptr; q.w2[i] = nullptr; q.w[i] = i < p.nseg ? wsrc[i] : nullptr; }
        if (p.seg_mode == LDM_SEG_K) {                                // one group per (segment, sub-range): per-group weight pointers
            q.use_table = 1; q.nseg = 1; q.seg_mode = LDM_SEG_N; q.seg_len = d.N; q.w_gstride = 0;
            for (int g = 0; g < S; ++g) { q.wtab[g] = wsrc[g / s2] + (long long)(g % s2) * ks; q.btab[g] = nullptr; }
        } else {
            q.w_gstride = ks;                                         // every N-segment's weights advance along K with the group
        }
        launch_any(q, S, false, LDM_A_ROWS, st);
    };
    float *pa = (float *)d.workspace, *pb = gate ? pa + (size_t)S * d.M * d.N : nullptr;
    partial(p.w, pa);
    if (gate) partial(p.w2, pb);
    SplitEpiP e{};
    e.pa = pa; e.pb = pb; e.S = S; e.M = d.M; e.N = d.N; e.seg_mode = p.seg_mode; e.nseg = p.nseg; e.seg_len = p.seg_len;
    e.act = d.act; e.slope = d.slope; e.addend = d.addend; e.ldadd = d.ldadd; e.ldo = d.ldo; e.out = d.out;
    for (int i = 0; i < LDM_MAX_SEG; ++i) { e.bias[i] = p.bias[i]; e.bias2[i] = p.bias2[i]; }
    const long long work = (long long)d.M * (d.N / 4);
    ldm_launch(splitk_epilogue_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, e);
    return true;
}

}  // namespace

int ldm_gemm_stream_wide(int v);

extern int g_gconv_wide;
extern "C" int ldm_gemm_wide_epilogue(int v)
{
    if (v == 0 || v == 1) g_gconv_wide = v;          // the grouped conv's 16-byte row epilogue follows the same switch
    return ldm_gemm_stream_wide(v);
}

extern "C" int ldm_gemm_variant(int v)
{
    const int old = g_variant;
    if (v >= 0 && v <= 2) g_variant = v;
    return old;
}

static int gemm_f32_impl(const ldm_gemm_desc *d, float *a_pre, float *b_pre, void *stream, float *db = nullptr);

extern "C" int ldm_gemm_f32(const ldm_gemm_desc *d, void *stream) { return gemm_f32_impl(d, nullptr, nullptr, stream); }

// ReGLU backward of the fp32 training step in the epilogue of dh = dY . Wc: d describes that plain GEMM (rows in, rows out [M, ldo], no
// activation, no addend); its result is NOT stored -- d->out receives da = dh * relu(b_pre), db receives dh * a_pre * (b_pre > 0), all fp32
// [M, ldo].  Bit-identical to ldm_gemm_f32 + ldm_gate_bwd_f32.  Returns 0 when launched, 1 when no kernel instance takes the shape.
extern "C" int ldm_gemm_f32_gate_bwd(const ldm_gemm_desc *d, const float *a_pre, const float *b_pre, float *db, void *stream)
{
    LDM_REQUIRE(d && a_pre && b_pre && db, "ldm_gemm_f32_gate_bwd: null pointer");
    LDM_REQUIRE(d->act == LDM_ACT_NONE && !d->addend && d->a_mode == LDM_A_ROWS && d->o_mode == LDM_O_ROWS, "ldm_gemm_f32_gate_bwd: plain rows problem without addend only");
    LDM_REQUIRE(ldm_aligned16(a_pre) && ldm_aligned16(b_pre) && ldm_aligned16(db) && ldm_aligned16(d->out) && d->ldo % 4 == 0,
                "ldm_gemm_f32_gate_bwd: operands must be 16-byte addressable");
    return gemm_f32_impl(d, (float *)a_pre, (float *)b_pre, stream, db);
}

// ReGLU forward of the fp32 training step in ONE launch: out = (A Wa^T + ba) * relu(A Wb^T + bb) AND a_pre / b_pre (the two
// pre-activations its backward needs), all fp32 [M, ldo].  d as for ldm_gemm_f32 with act = LDM_ACT_GATE, rows in, rows out, no addend.
// Returns LDM_OK when launched, 1 when no kernel instance takes the shape (the caller then runs two plain GEMMs + ldm_gate_fwd_f32).
extern "C" int ldm_gemm_f32_gate_fwd(const ldm_gemm_desc *d, float *a_pre, float *b_pre, void *stream)
{
    LDM_REQUIRE(d && a_pre && b_pre, "ldm_gemm_f32_gate_fwd: null pointer");
    LDM_REQUIRE(d->act == LDM_ACT_GATE && !d->addend && d->a_mode == LDM_A_ROWS && d->o_mode == LDM_O_ROWS, "ldm_gemm_f32_gate_fwd: gated rows problem without addend only");
    LDM_REQUIRE(ldm_aligned16(a_pre) && ldm_aligned16(b_pre) && ldm_aligned16(d->out) && d->ldo % 4 == 0, "ldm_gemm_f32_gate_fwd: outputs must be 16-byte addressable");
    return gemm_f32_impl(d, a_pre, b_pre, stream);
}

static int gemm_f32_impl(const ldm_gemm_desc *d, float *a_pre, float *b_pre, void *stream, float *db)
{
    LDM_REQUIRE(d != nullptr, "ldm_gemm_f32: null descriptor");
    LDM_REQUIRE(d->a && d->out, "ldm_gemm_f32: null operand");
    LDM_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "ldm_gemm_f32: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    LDM_REQUIRE(d->N % 32 == 0 && d->K % 32 == 0, "ldm_gemm_f32: N=%d and K=%d must be multiples of 32", d->N, d->K);
    LDM_REQUIRE(d->nseg >= 1 && d->nseg <= LDM_MAX_SEG, "ldm_gemm_f32: nseg=%d", d->nseg);
    LDM_REQUIRE(d->lda % 4 == 0 && d->ldw % 4 == 0, "ldm_gemm_f32: lda/ldw must be multiples of 4 floats");
    LDM_REQUIRE(ldm_aligned16(d->a), "ldm_gemm_f32: A not 16-byte aligned");
    const bool gate = d->act == LDM_ACT_GATE;
    const int groups = d->groups > 0 ? d->groups : 1;
    LDM_REQUIRE(d->a_gstride % 4 == 0 && d->w_gstride % 4 == 0, "ldm_gemm_f32: group strides must be multiples of 4 floats");
    LDM_REQUIRE(d->seg_mode == LDM_SEG_N || d->seg_mode == LDM_SEG_K, "ldm_gemm_f32: seg_mode=%d", d->seg_mode);
    const int seg_total = d->seg_mode == LDM_SEG_N ? d->N : d->K;
    const int seg_len = d->nseg == 1 ? seg_total : d->seg_len;
    LDM_REQUIRE(seg_len > 0 && seg_len % 32 == 0 && (long long)seg_len * d->nseg == seg_total,
                "ldm_gemm_f32: segments (%d x %d) do not cover %d", d->nseg, seg_len, seg_total);
    LDM_REQUIRE(!(gate && d->seg_mode == LDM_SEG_K && d->nseg > 1), "ldm_gemm_f32: GATE with K-segments is not supported");
    if (d->w_table) LDM_REQUIRE(d->nseg == 1 && !gate, "ldm_gemm_f32: pointer-table mode needs nseg == 1 and no GATE");
    for (int s = 0; s < d->nseg && !d->w_table; ++s) {
        LDM_REQUIRE(d->w[s] && ldm_aligned16(d->w[s]), "ldm_gemm_f32: weight segment %d null/unaligned", s);
        if (gate) LDM_REQUIRE(d->w2[s] && ldm_aligned16(d->w2[s]), "ldm_gemm_f32: gate weight segment %d null/unaligned", s);
    }
    GemmP p{};
    p.a = d->a; p.lda = d->lda; p.M = d->M; p.N = d->N; p.K = d->K;
    p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.cpt = 1;
    if (d->a_mode == LDM_A_CONV3X3) {
        LDM_REQUIRE(d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % 32 == 0 && d->K == 9 * d->Cin, "ldm_gemm_f32: conv3x3 needs K == 9*Cin, Cin %% 32 == 0");
        LDM_REQUIRE(d->M % (d->H * d->W) == 0, "ldm_gemm_f32: conv3x3 M=%d is not a multiple of H*W", d->M);
        p.cpt = d->Cin / 32;
    } else {
        LDM_REQUIRE(d->a_mode == LDM_A_ROWS, "ldm_gemm_f32: a_mode=%d", d->a_mode);
    }
    p.nseg = d->nseg; p.seg_mode = d->seg_mode;
    p.seg_len = seg_len;
    for (int s = 0; s < LDM_MAX_SEG; ++s) {
        p.w[s] = s < d->nseg ? d->w[s] : nullptr;
        p.w2[s] = s < d->nseg ? d->w2[s] : nullptr;
        p.bias[s] = s < d->nseg ? d->bias[s] : nullptr;
        p.bias2[s] = s < d->nseg ? d->bias2[s] : nullptr;
    }
    p.ldw = d->ldw; p.act = d->act; p.slope = d->slope;
    p.addend = d->addend; p.ldadd = d->ldadd; p.out = d->out; p.ldo = d->ldo;
    p.o_mode = d->o_mode; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
    if (d->o_mode != LDM_O_ROWS) {
        LDM_REQUIRE(d->OH > 0 && d->OW > 0 && d->M % (d->OH * d->OW) == 0, "ldm_gemm_f32: scatter output needs M %% (OH*OW) == 0");
        if (d->o_mode == LDM_O_CONVT2X2) LDM_REQUIRE(d->Cout > 0 && d->N == 4 * d->Cout, "ldm_gemm_f32: convT2x2 needs N == 4*Cout");
    }
    p.a_gstride = d->a_gstride; p.w_gstride = d->w_gstride; p.o_gstride = d->o_gstride; p.b_gstride = d->b_gstride;
    p.use_table = d->w_table ? 1 : 0;
    p.wide_ok = d->o_mode == LDM_O_ROWS && ldm_aligned16(d->out) && d->ldo % 4 == 0 && d->o_gstride % 4 == 0 &&
                (!d->addend || (ldm_aligned16(d->addend) && d->ldadd % 4 == 0));
    p.scat_ok = d->o_mode != LDM_O_ROWS && ldm_aligned16(d->out) && d->ldo % 4 == 0 && d->o_gstride % 4 == 0 && d->M < (1 << 23) &&
                (!d->addend || (ldm_aligned16(d->addend) && d->ldadd % 4 == 0)) && (d->o_mode != LDM_O_CONVT2X2 || d->Cout % 4 == 0);
    p.inv_ow = d->o_mode != LDM_O_ROWS ? 1.0f / (float)d->OW : 0.f;
    if (d->w_table) {
        LDM_REQUIRE(groups <= LDM_MAX_TABLE, "ldm_gemm_f32: pointer-table mode supports at most %d groups", LDM_MAX_TABLE);
        for (int i = 0; i < groups; ++i) {
            LDM_REQUIRE(d->w_table[i] && ldm_aligned16(d->w_table[i]), "ldm_gemm_f32: table weight %d null/unaligned", i);
            p.wtab[i] = d->w_table[i];
            p.btab[i] = d->bias_table ? d->bias_table[i] : nullptr;
        }
    }

    hipStream_t st = (hipStream_t)stream;
    // algorithmic HBM bytes of the launch: every operand element once (the 3x3 im2col re-reads and the x2 replicate of the
    // addressing modes are not algorithmic), weights once per group
    const double a_elems = (double)d->M * (d->a_mode == LDM_A_CONV3X3 ? d->Cin : d->K) * (d->a_gstride || groups == 1 ? groups : 1);
    const double o_elems = (double)d->M * d->N * groups * (d->o_mode == LDM_O_UP2 ? 4.0 : 1.0);
    const double algo_bytes = 4.0 * (a_elems + (double)d->N * d->K * groups * (gate ? 2.0 : 1.0) + o_elems * (d->addend ? 2.0 : 1.0));
    LDM_REQUIRE(!(gate && d->a_mode == LDM_A_CONV3X3), "ldm_gemm_f32: GATE with conv3x3 unsupported");       // before the profiler opens a record
    void *rec = ldm_prof_begin(LDM_PROF_GEMM, 2.0 * d->M * (double)d->N * d->K * groups * (gate ? 2.0 : 1.0), st,
                               algo_bytes + (a_pre ? 8.0 * o_elems : 0.0));
    if (a_pre) {                                                  // fused ReGLU forward / backward: a ring-kernel instance or nothing
        if (db) {
            p.in2 = a_pre;
            p.in3 = b_pre;
            p.out2 = db;
        } else {
            p.out2 = a_pre;
            p.out3 = b_pre;
        }
        const int taken = g_variant == 1 ? ldm_gemm_ring_dispatch_f32(p, groups, db == nullptr, d->a_mode, st) : 0;
        ldm_prof_end(rec, st);
        if (!taken) return 1;
        LDM_CHECK_LAUNCH(db ? "ldm_gemm_f32_gate_bwd" : "ldm_gemm_f32_gate_fwd");
        return LDM_OK;
    }
    if (!(d->workspace && splitk_launch(*d, p, gate, st))) launch_any(p, groups, gate, d->a_mode, st);
    ldm_prof_end(rec, st);
    LDM_CHECK_LAUNCH("ldm_gemm_f32");
    return LDM_OK;
}
