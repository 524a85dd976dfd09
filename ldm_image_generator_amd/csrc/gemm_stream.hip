// Persistent, LDS-DMA-fed fp32-MFMA GEMM ("stream" kernel).
//
// Same math and operand conventions as gemm_f32_kernel (gemm.hip), different schedule:
//   * resident workgroups (2 per CU) walk a static, XCD-aware list of output tiles; the K-steps of ALL
//     tiles of a workgroup form one continuous stream, so a tile's first K-step and its epilogue never
//     expose memory latency (short-K shapes such as C = 128 otherwise live in exactly those two places);
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write pass) into
//     a 2-stage ring; the DMA for step s+1 is issued at the top of step s;
//   * ONE sync point per K-step, placed before the LAST quarter of the step's MFMAs: the fragments of that
//     quarter are already in registers, so after the barrier a wave still owns 16 MFMAs (1024 cycles) of
//     work under which it fetches the next step's first fragments -- the matrix pipe never drains at a
//     step boundary;
//   * bias / addend values of a tile are prefetched (branch-free) under its last K-step.
// LDS-DMA writes lane-linear (wave-uniform base + lane*16 B), so the 16-B-chunk XOR swizzle is applied to
// the per-lane SOURCE address; fragment reads use the same swizzle.  Rows past M are clamped (their results
// are never stored); zero padding of the implicit 3x3 im2col reads a 64-B block of zeros.
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

using namespace ldmgemm;

namespace {

constexpr int NS = 2;      // LDS ring stages
int g_wide = 1;            // wide (LDS-transposed, 16-byte-per-lane) epilogue for rows outputs; 0 keeps the direct one (A/B tests)

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void glds16(const float *src, float *lds_dst)
{
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}


// ---- "split" consumer (SPLIT = true) -------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores: every fp32 operand value is cut, exactly, into three bf16 pieces
// x = hi + mid + lo (8 + 8 + 8 mantissa bits, truncating), and a product tile is six
// v_mfma_f32_32x32x16_bf16 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; small terms first; fp32
// accumulate).  Only mid*lo, lo*mid, lo*lo are dropped (<= 2^-24 relative each), so results carry
// fp32-level error but are NOT bit-identical to the v_mfma_f32_32x32x2_f32 schedule.  Operands still arrive
// as fp32 (same LDS ring, same swizzle, weights in place): the split happens in registers per fragment.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Split3 {
    bf16x8 hi, mid, lo;
};

__device__ __forceinline__ void split3(const f32x4 &c0, const f32x4 &c1, Split3 &o)
{
    u32x4 H, M, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh[2], mm[2], ll[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = 2 * i + j;
            const float v = e < 4 ? c0[e & 3] : c1[e & 3];
            hh[j] = __float_as_uint(v) & 0xFFFF0000u;
            const float r1 = v - __uint_as_float(hh[j]);          // exact
            mm[j] = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(mm[j]);         // exact, <= 8 significant bits: a bf16
            ll[j] = __float_as_uint(r2);
        }
        H[i] = __builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u);   // two high halves -> one dword
        M[i] = __builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u);
        L[i] = __builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u);
    }
    o.hi = __builtin_bit_cast(bf16x8, H);
    o.mid = __builtin_bit_cast(bf16x8, M);
    o.lo = __builtin_bit_cast(bf16x8, L);
}

__device__ __forceinline__ void mfma6(const Split3 &a, const Split3 &b, f32x16 &c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
}

// ET = 1: both operands are bf16 (rows of 128 B = 64 k), one v_mfma_f32_32x32x16_bf16 per fragment pair; the loader, the LDS
// ring and the swizzle are unchanged because they only ever move 16-byte chunks of 128-byte rows -- the host passes K, lda,
// ldw and the group strides in units of 4 bytes (two bf16).  OBF: the (wide) epilogue rounds the result to bf16 (RNE).
template <int WM, int WN, int TM, int TN, bool GATE, int AMODE, int SPLIT = 0, bool WIDE = false, int ET = 0, bool OBF = false, bool SCAT = false, bool GBW = false>
__global__ __launch_bounds__(256, 2) void gemm_stream_kernel(const GemmP p, int ntm, int ntn, int total_tiles)
{
    static_assert(ET == 0 || SPLIT == 0, "bf16 operands: no split consumer");
    static_assert(!OBF || WIDE, "bf16 output goes through the wide epilogue");
    static_assert(!SCAT || (WIDE && !GATE && !OBF && ET == 0), "wide scatter epilogue: plain fp32 instances only");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NB = GATE ? 2 * BN : BN;
    constexpr int A_F4 = BM / 32, B_F4 = NB / 32;
    constexpr int STAGE = (BM + NB) * 32;
    constexpr int NACC = GATE ? 2 : 1;
    static_assert(!WIDE || A_F4 + B_F4 >= 4 * TN, "wide epilogue: not enough per-wave LDS slices in one ring stage");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int chunk = t & 7, lrow = t >> 3;
    const int nk = p.K >> 5;
    const int per_group = ntm * ntn;

    // tiles of this workgroup: logical ids xcd_remap(blockIdx.x + i * gridDim.x)
    const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_steps = my_tiles * nk;

    auto tile_coords = [&](int i, int &g, int &m0, int &n0) {
        const int id = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, total_tiles);
        g = id / per_group;
        int tm_, tn_;
        tile_from_id(id - g * per_group, ntm, ntn, tm_, tn_);
        m0 = tm_ * BM;
        n0 = tn_ * BN;
    };

    // ---- loader cursor -------------------------------------------------------
    // Wave-uniform running base pointers (SGPRs) + constant per-lane 32-bit offsets: advancing the stream
    // by one K-step is a handful of scalar adds, no per-lane 64-bit address arithmetic and no division.
    int l_tile = 0, l_kt = 0, l_step = 0, l_g = 0;
    int l_seg = 0, l_kin = 0;                    // weight segment of the current K-step, K-steps done inside it
    int l_tap = 0, l_cin = 0;                    // conv: tap index, 32-channel chunk inside the tap
    const float *a_cur = nullptr;                // A tile base (+ K offset of the current step)
    const float *w_row0 = nullptr, *w2_row0 = nullptr;   // weight rows of this tile at K offset 0 of the segment
    int a_voff[A_F4], b_voff[B_F4];              // per-lane element offsets (row * stride + swizzled chunk)
    int a_y[A_F4], a_x[A_F4];
    const int seg_steps = p.seg_mode == LDM_SEG_K ? p.seg_len >> 5 : 0x7fffffff;   // K-steps per weight segment
    int l_nloc0 = 0;
    int l_m0 = 0, l_n0 = 0, l_seg_n = 0;         // coordinates of the loader's current tile
    int i_g = 0, i_m0 = 0, i_n0 = 0, i_seg_n = 0;  // ... of the tile whose FIRST K-step was issued last (the consumer takes them over)

    auto weight_rows = [&]() {
        const long long off = (long long)l_nloc0 * p.ldw;
        w_row0 = (p.use_table ? p.wtab[l_g] : p.w[l_seg] + l_g * p.w_gstride) + off;
        if (GATE) w2_row0 = p.w2[l_seg] + l_g * p.w_gstride + off;
    };
    auto loader_setup = [&]() {
        int m0, n0;
        tile_coords(l_tile, l_g, m0, n0);
        l_m0 = m0;
        l_n0 = n0;
        l_seg = (p.seg_mode == LDM_SEG_N) ? n0 / p.seg_len : 0;
        l_seg_n = l_seg;
        l_nloc0 = (p.seg_mode == LDM_SEG_N) ? n0 - l_seg * p.seg_len : n0;
        l_kin = 0;
        l_tap = 0;
        l_cin = 0;
        a_cur = p.a + l_g * p.a_gstride + (long long)m0 * p.lda;
        weight_rows();
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = lrow + 32 * i;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            a_voff[i] = (m - m0) * (int)p.lda + (chunk ^ ((row >> 1) & 7)) * 4;
            if (AMODE == LDM_A_CONV3X3) {
                a_x[i] = m % p.W;
                a_y[i] = (m / p.W) % p.H;
            }
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int row = lrow + 32 * i;
            const int nrow = (GATE && row >= BN) ? row - BN : row;
            b_voff[i] = nrow * (int)p.ldw + (chunk ^ ((row >> 1) & 7)) * 4;
        }
    };

    auto loader_issue = [&]() {
        float *As = lds + (l_step & 1) * STAGE, *Bs = As + BM * 32;
        const bool first = l_kt == 0;                 // branch-free hand-over of the tile coordinates
        i_g = first ? l_g : i_g;
        i_m0 = first ? l_m0 : i_m0;
        i_n0 = first ? l_n0 : i_n0;
        i_seg_n = first ? l_seg_n : i_seg_n;
        if (AMODE == LDM_A_CONV3X3) {
            const int dy = l_tap / 3 - 1, dx = l_tap - (l_tap / 3) * 3 - 1;
            const float *a_tap = a_cur + ((long long)(dy * p.W + dx) * p.lda + (l_cin << 5));
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const bool ok = (unsigned)(a_y[i] + dy) < (unsigned)p.H && (unsigned)(a_x[i] + dx) < (unsigned)p.W;
                glds16(ok ? a_tap + a_voff[i] : ldm_zero_block + (chunk & 3) * 4, As + (i * 4 + wave) * 256);
            }
            if (++l_cin == p.cpt) {
                l_cin = 0;
                ++l_tap;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) glds16(a_cur + a_voff[i], As + (i * 4 + wave) * 256);
            a_cur += 32;
        }
        const float *wa = w_row0 + (l_kin << 5);
        const float *wb = GATE ? w2_row0 + (l_kin << 5) : nullptr;
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const bool second = GATE && (32 * i >= BN);
            glds16((second ? wb : wa) + b_voff[i], Bs + (i * 4 + wave) * 256);
        }
        ++l_step;
        if (++l_kt == nk) {
            l_kt = 0;
            ++l_tile;
            if (l_tile < my_tiles) loader_setup();
        } else if (++l_kin == seg_steps) {           // next K-segment (LDM_SEG_K): switch weight pointers
            l_kin = 0;
            ++l_seg;
            weight_rows();
        }
    };

    // ---- consumer ------------------------------------------------------------
    f32x16 acc[NACC][TM][TN];
    float pre[TM][TN][16];
    const WideLane<TN> wl = wide_lane<TN>(lane);
    f32x4 fa0[TM], fb0[NACC][TN], fa1[TM], fb1[NACC][TN];        // two fragment sets (ping-pong over j); SPLIT: the two chunks of a half
    Split3 sa[SPLIT ? TM : 1], sb[SPLIT ? NACC : 1][SPLIT ? TN : 1];   // SPLIT: 0 exact fp32, 1 split, 2 split with paced interleave
    if (my_tiles == 0) return;
    // accumulators are cleared here and again right after each tile's epilogue -- NOT by a per-step
    // "c_kt == 0 ? 0 : acc" select, which would put 16 VALU selects per accumulator tile (each waiting
    // for the previous step's last MFMA) at the head of every K-step and drain the matrix pipe there
    auto clear_acc = [&]() {
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[q][i][j][e] = 0.f;
    };
    clear_acc();

    auto read_frags = [&](int step, int j, f32x4 (&af)[TM], f32x4 (&bf)[NACC][TN]) {
        const float *As = lds + (step & 1) * STAGE, *Bs = As + BM * 32;
        const int c = 2 * j + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f32x4 *)(As + swz((wm * TM + i) * 32 + r, c));
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < TN; ++i) bf[q][i] = *(const f32x4 *)(Bs + swz(q * BN + (wn * TN + i) * 32 + r, c));
    };
    auto mma = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[NACC][TN]) {
        if constexpr (ET == 1) {
            // a 16-byte chunk = 8 consecutive k of the lane's row: exactly the A / B fragment of the 32x32x16 instruction
            // (lane half h holds k = 8 h .. 8 h + 7 of the 16-k slice 2 j + h -> slice j)
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        acc[q][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[q][jj]),
                                                                                acc[q][i][jj], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        acc[q][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[q][jj][e], acc[q][i][jj], 0, 0, 0);
    };
    // steps 0..2 of the K-step: three quarters of the MFMAs, all remaining fragment reads of this stage
    auto quarters_0_to_2 = [&](int s) {
        read_frags(s, 1, fa1, fb1);
        mma(fa0, fb0);
        read_frags(s, 2, fa0, fb0);
        mma(fa1, fb1);
        read_frags(s, 3, fa1, fb1);
        mma(fa0, fb0);
    };
    // the sync point: this wave's DMA for step s+1 has landed and its reads of stage s are complete;
    // after the barrier that holds for every wave, so stage s may be overwritten and stage s+1 read
    auto sync_point = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // SPLIT: a K-step is two halves of 16 k each; lane group h owns chunks {4*hf + h, 4*hf + 2 + h} of a row
    // (any assignment works as long as A and W use the same one)
    auto read_half = [&](int step, int hf) {
        const float *As = lds + (step & 1) * STAGE, *Bs = As + BM * 32;
        const int c0 = 4 * hf + h, c1 = c0 + 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            fa0[i] = *(const f32x4 *)(As + swz((wm * TM + i) * 32 + r, c0));
            fa1[i] = *(const f32x4 *)(As + swz((wm * TM + i) * 32 + r, c1));
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                fb0[q][i] = *(const f32x4 *)(Bs + swz(q * BN + (wn * TN + i) * 32 + r, c0));
                fb1[q][i] = *(const f32x4 *)(Bs + swz(q * BN + (wn * TN + i) * 32 + r, c1));
            }
    };
    // One half (16 k) of a K-step of the split consumer.  A split is 44 VALU instructions, a tile 6 MFMAs (192 matrix
    // cycles, of which the issuing wave is held for 48): each tile's MFMAs are interleaved with the split of the NEXT
    // operand some later tile needs (sched_group_barrier: 1 MFMA, 8 VALU, ...), so only the first two splits of a half
    // are exposed.  `mid` runs at the point where all raw fragment registers have been consumed (the next half's reads,
    // or the step's sync point, go there).
    auto split_b = [&](int bi) { split3(fb0[bi / TN][bi % TN], fb1[bi / TN][bi % TN], sb[bi / TN][bi % TN]); };
    auto tile_mma = [&](int a, int bi) { mfma6(sa[a], sb[bi / TN][bi % TN], acc[bi / TN][a][bi % TN]); };
    auto interleave = [&]() {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
    };
    auto split_half = [&](auto mid, auto paced, auto mid_early) {
        if constexpr (SPLIT) {
            constexpr int NB = NACC * TN;
            static_assert(TM == 2 && (NB == 1 || NB == 2), "split consumer: 2 x {1, 2} tiles per wave");
            if constexpr (!decltype(paced)::value || SPLIT != 2) {   // unpaced: all splits, then all MFMAs; the scheduler is free
                split3(fa0[0], fa1[0], sa[0]);
                split3(fa0[1], fa1[1], sa[1]);
#pragma unroll
                for (int bi = 0; bi < NB; ++bi) split_b(bi);
                if constexpr (decltype(mid_early)::value) mid();      // first half: the second half's reads fly under all MFMAs
#pragma unroll
                for (int bi = 0; bi < NB; ++bi) tile_mma(0, bi);
                if constexpr (!decltype(mid_early)::value) mid();     // second half: the sync point sits before the last MFMAs
#pragma unroll
                for (int bi = 0; bi < NB; ++bi) tile_mma(1, bi);
                return;
            }
            split3(fa0[0], fa1[0], sa[0]);
            split_b(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NB == 2) {
                split_b(1);
                tile_mma(0, 0);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
            split3(fa0[1], fa1[1], sa[1]);
            tile_mma(0, NB - 1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            mid();
            tile_mma(1, 0);
            if constexpr (NB == 2) tile_mma(1, 1);
        }
    };

    // ---- prologue: stage 0 <- step 0 -------------------------------------------
    loader_setup();
    loader_issue();
    sync_point();
    if constexpr (SPLIT) read_half(0, 0); else read_frags(0, 0, fa0, fb0);

    // Tile loop outside, K-steps inside: the accumulators are loop-carried in FIXED registers through a
    // single-path inner loop (one merged loop with a "last step?" branch made hipcc shuttle all accumulator
    // registers through v_mov copies behind an s_nop at every K-step, draining the matrix pipe each time).
    int s = 0;                                            // position in this workgroup's K-step stream
#pragma unroll 1
    for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
        // the loader runs exactly one K-step ahead: the last FIRST-step it issued belongs to this tile.  Take its
        // coordinates over instead of recomputing them (tile order = an xcd remap and three integer divisions)
        const int c_g = i_g, c_m0 = i_m0, c_n0 = i_n0, c_seg_n = i_seg_n;
        clear_acc();
#pragma unroll 1
        for (int kt = 0; kt < nk - 1; ++kt, ++s) {
            loader_issue();                               // step s+1 (exists: this is not the tile's last step)
            if constexpr (SPLIT) {
                split_half([&]() { read_half(s, 1); }, std::true_type{}, std::true_type{});
                split_half([&]() {
                    sync_point();
                    read_half(s + 1, 0);
                }, std::true_type{}, std::false_type{});
            } else {
                quarters_0_to_2(s);
                sync_point();
                read_frags(s + 1, 0, fa0, fb0);
                mma(fa1, fb1);
            }
        }
        // last K-step of the tile: bias/addend prefetch, MFMAs and epilogue on ONE control path, so the
        // only wait for the prefetched registers sits in front of their first use
        const bool more = s + 1 < total_steps;
        if (more) loader_issue();
        const bool use_pre = p.addend != nullptr && p.o_mode == LDM_O_ROWS;
        const int seg_n = c_seg_n;
        EpiCols<TN> cols;
        gemm_epilogue_cols<WN, TN, GATE>(p, cols, c_n0, c_g, seg_n, wn, r);
        if constexpr (WIDE) {
            if (use_pre) gemm_prefetch_addend_wide<WM, WN, TM, TN>(p, pre, c_m0, c_n0, c_g, wm, wn, wl);
        } else {
            if (use_pre) gemm_prefetch_addend<WM, WN, TM, TN>(p, pre, c_m0, c_n0, c_g, wm, wn, r, h);
        }
        if constexpr (SPLIT) {
            split_half([&]() { read_half(s, 1); }, std::false_type{}, std::true_type{});
            split_half([&]() {
                sync_point();
                if (more) read_half(s + 1, 0);
            }, std::false_type{}, std::false_type{});
        } else {
            quarters_0_to_2(s);
            sync_point();
            if (more) read_frags(s + 1, 0, fa0, fb0);
            mma(fa1, fb1);
        }
        if constexpr (WIDE)      // scratch: this wave's slices of the stage the tile's last step has just released
            gemm_epilogue_wide<WM, WN, TM, TN, GATE, OBF, SCAT, GBW>(p, acc, c_m0, c_n0, c_g, wm, wn, cols, pre, use_pre, wl, lds + (s & 1) * STAGE + wave * 256);
        else
            gemm_epilogue<WM, WN, TM, TN, GATE>(p, acc, c_m0, wm, h, cols, pre, use_pre);
        ++s;
    }
}

template <int WM, int WN, int TM, int TN, bool GATE, int AMODE, int SPLIT = 0, bool WIDE = false, int ET = 0, bool OBF = false, bool SCAT = false, bool GBW = false>
int launch_stream(const GemmP &p, int groups, hipStream_t st)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NB = GATE ? 2 * BN : BN;
    constexpr size_t smem = (size_t)NS * (BM + NB) * 32 * sizeof(float);
    static LdmLdsOptIn opt_in;
    static std::atomic<int> per_cu_cache{0};
    auto kern = gemm_stream_kernel<WM, WN, TM, TN, GATE, AMODE, SPLIT, WIDE, ET, OBF, SCAT, GBW>;
    (void)opt_in((const void *)kern, smem);
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 256, smem) != hipSuccess || per_cu < 1) per_cu = 2;
        per_cu_cache.store(per_cu, std::memory_order_relaxed);
    }
    const int slots = ldm_cu_count() * (per_cu > 4 ? 4 : per_cu);          // resident workgroups (no grid barrier: a wrong guess only skews load)
    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    const long long total = (long long)ntm * ntn * groups;
    if (total > 0x7fffffffLL) return 0;
    int grid = (int)(total < slots ? total : slots);
    if (grid > 8) grid &= ~7;                                          // keep blockIdx % 8 == XCD label across rounds
    ldm_launch(kern, dim3(grid), dim3(256), smem, st, p, ntm, ntn, (int)total);
    return 1;
}

template <int WM, int WN, int TM, int TN, bool GATE, int AMODE, int SPLIT = 0>
int launch_stream_w(const GemmP &p, int groups, hipStream_t st)
{
    if (p.wide_ok && g_wide) return launch_stream<WM, WN, TM, TN, GATE, AMODE, SPLIT, true>(p, groups, st);
    return launch_stream<WM, WN, TM, TN, GATE, AMODE, SPLIT, false>(p, groups, st);
}

}  // namespace

int ldm_gemm_stream_wide(int v)
{
    const int old = g_wide;
    if (v == 0 || v == 1) g_wide = v;
    return old;
}

// Chooses a stream-kernel instance for the problem; returns 1 if it launched, 0 if the caller should
// fall back to the tile-per-block kernel.
int ldm_gemm_stream_dispatch(const GemmP &p, int groups, bool gate, int amode, hipStream_t st, bool split)
{
    const int unit = (p.seg_mode == LDM_SEG_N) ? p.seg_len : p.N;
    if (split && p.M > 128) {
        // split schedule: wave tiles of 64x64 (or 64x32 x two gate matrices) so that each split fragment feeds two
        // tiles; shapes it does not cover (grouped conv N = 32, tiny M) fall through to the exact-fp32 instances.
        // "paced" (SPLIT = 2) pins each tile's MFMAs over the next operand's split with sched_group_barrier: +3-8 %
        // when the A operand is cache-resident (deep stages, conv taps), -10 % when it streams from HBM (the fences
        // take away the scheduler's freedom around the DMA wait), hence the size test
        const bool paced = amode == LDM_A_CONV3X3 || (long long)p.M * p.K * 4 <= (48ll << 20);
        if (gate && amode == LDM_A_ROWS && unit % 64 == 0)
            return paced ? launch_stream_w<2, 2, 2, 1, true, LDM_A_ROWS, 2>(p, groups, st) : launch_stream_w<2, 2, 2, 1, true, LDM_A_ROWS, 1>(p, groups, st);
        if (!gate && amode == LDM_A_CONV3X3 && unit % 128 == 0) return launch_stream_w<2, 2, 2, 2, false, LDM_A_CONV3X3, 2>(p, groups, st);
        if (!gate && amode == LDM_A_ROWS && unit % 128 == 0)
            return paced ? launch_stream_w<2, 2, 2, 2, false, LDM_A_ROWS, 2>(p, groups, st) : launch_stream_w<2, 2, 2, 2, false, LDM_A_ROWS, 1>(p, groups, st);
        // N = 64 (VAE stage 3): 64x32 per wave, three splits feed two tiles -- still ahead of the exact instruction
        if (!gate && amode == LDM_A_CONV3X3 && unit % 64 == 0) return launch_stream_w<2, 2, 2, 1, false, LDM_A_CONV3X3, 2>(p, groups, st);
        if (!gate && amode == LDM_A_ROWS && unit % 64 == 0) return launch_stream_w<2, 2, 2, 1, false, LDM_A_ROWS, 1>(p, groups, st);
    }
    if (gate) {
        if (amode != LDM_A_ROWS) return 0;
        if (unit % 64 == 0) return launch_stream_w<2, 2, 2, 1, true, LDM_A_ROWS>(p, groups, st);
        return launch_stream_w<4, 1, 1, 1, true, LDM_A_ROWS>(p, groups, st);
    }
    if (amode == LDM_A_CONV3X3) {
        if (unit % 128 == 0 && (long long)((p.M + 127) / 128) * (p.N / 128) * groups >= 512)
            return launch_stream_w<2, 2, 2, 2, false, LDM_A_CONV3X3>(p, groups, st);
        if (unit % 64 == 0) return launch_stream_w<2, 2, 2, 1, false, LDM_A_CONV3X3>(p, groups, st);
        return launch_stream_w<4, 1, 1, 1, false, LDM_A_CONV3X3>(p, groups, st);
    }
    if (p.o_mode != LDM_O_ROWS && p.scat_ok && g_wide && p.M > 32) {
        // convT 2x2 / up x2: the wide epilogue with scattered row addresses (16-byte stores; the direct one is 4-byte stores behind
        // three integer divisions per accumulator element)
        if (unit % 128 == 0 && (long long)((p.M + 127) / 128) * (p.N / 128) * groups >= 512)
            return launch_stream<2, 2, 2, 2, false, LDM_A_ROWS, 0, true, 0, false, true>(p, groups, st);
        if (unit % 64 == 0) return launch_stream<2, 2, 2, 1, false, LDM_A_ROWS, 0, true, 0, false, true>(p, groups, st);
    }
    if (p.M <= 32 && unit % 128 == 0) return launch_stream_w<1, 4, 1, 1, false, LDM_A_ROWS>(p, groups, st);
    // 33..64 rows against many weight matrices (the FiLM MLPs of the 8 x 8 level: 64 rows x 18 blocks x 8 M weights): a
    // weight-streaming launch; 64 x 128 tiles waste no rows and keep 16-KB weight slices per K-step
    if (p.M <= 64 && unit % 128 == 0) return launch_stream<2, 2, 1, 2, false, LDM_A_ROWS, 0, false>(p, groups, st);      // direct epilogue: the tile has too few LDS slices for the wide one
    // 128x128 tiles where N allows and the tile count still fills the chip (two workgroups per CU): a third fewer LDS-DMA instructions per MFMA (with the wide
    // epilogue the instance fits two workgroups per CU without spills); measured +5-6 % at C = 256 / 512, neutral elsewhere
    if (unit % 128 == 0 && (long long)((p.M + 127) / 128) * (p.N / 128) * groups >= 512) return launch_stream_w<2, 2, 2, 2, false, LDM_A_ROWS>(p, groups, st);
    if (unit % 64 == 0) return launch_stream_w<2, 2, 2, 1, false, LDM_A_ROWS>(p, groups, st);
    return launch_stream_w<4, 1, 1, 1, false, LDM_A_ROWS>(p, groups, st);
}


// bf16 operands (ldm_gemm_bf16): plain rows in, rows out (fp32 or bf16); p is already in 4-byte units along K.
int ldm_gemm_stream_dispatch_bf16(const GemmP &p, int groups, bool out_bf16, hipStream_t st, bool gate, int amode)
{
    const int unit = (p.seg_mode == LDM_SEG_N) ? p.seg_len : p.N;
    if (amode == LDM_A_CONV3X3) {        // dense 3x3 of the VAE in bf16 (decode under autocast): implicit im2col in the loader, bf16 rows out
        if (gate || !out_bf16 || !p.wide_ok) return 0;
        if (unit % 128 == 0 && (long long)((p.M + 127) / 128) * (p.N / 128) * groups >= 512)
            return launch_stream<2, 2, 2, 2, false, LDM_A_CONV3X3, 0, true, 1, true>(p, groups, st);
        if (unit % 64 == 0) return launch_stream<2, 2, 2, 1, false, LDM_A_CONV3X3, 0, true, 1, true>(p, groups, st);
        return 0;
    }
    if (gate) {          // a * relu(b) from two weight matrices per tile (128 x 64 x 2), bf16 out (+ the saved pre-activations)
        if (!out_bf16 || unit % 64) return 0;
        return launch_stream<2, 2, 2, 1, true, LDM_A_ROWS, 0, true, 1, true>(p, groups, st);
    }
    const bool big = unit % 128 == 0 && (long long)((p.M + 127) / 128) * (p.N / 128) * groups >= 512;
    if (out_bf16) {
        if (!p.wide_ok) return 0;
        if (p.in2) {                     // ReGLU backward in the epilogue (ldm_gemm_bf16_gate_bwd): its own instances
            if (p.act != LDM_ACT_NONE) return 0;
            if (big) return launch_stream<2, 2, 2, 2, false, LDM_A_ROWS, 0, true, 1, true, false, true>(p, groups, st);
            if (unit % 64 == 0) return launch_stream<2, 2, 2, 1, false, LDM_A_ROWS, 0, true, 1, true, false, true>(p, groups, st);
            return 0;
        }
        if (big) return launch_stream<2, 2, 2, 2, false, LDM_A_ROWS, 0, true, 1, true>(p, groups, st);
        if (unit % 64 == 0) return launch_stream<2, 2, 2, 1, false, LDM_A_ROWS, 0, true, 1, true>(p, groups, st);
        return 0;
    }
    if (p.wide_ok && g_wide) {
        if (big) return launch_stream<2, 2, 2, 2, false, LDM_A_ROWS, 0, true, 1, false>(p, groups, st);
        if (unit % 64 == 0) return launch_stream<2, 2, 2, 1, false, LDM_A_ROWS, 0, true, 1, false>(p, groups, st);
        return 0;
    }
    if (big) return launch_stream<2, 2, 2, 2, false, LDM_A_ROWS, 0, false, 1, false>(p, groups, st);
    if (unit % 64 == 0) return launch_stream<2, 2, 2, 1, false, LDM_A_ROWS, 0, false, 1, false>(p, groups, st);
    return 0;
}
