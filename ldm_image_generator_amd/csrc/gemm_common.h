// Shared pieces of the two fp32-MFMA GEMM kernels (tile-per-block `gemm_f32_kernel`
// and the persistent LDS-DMA stream `gemm_stream_kernel`).
#pragma once
#include "common.h"

namespace ldmgemm {

struct GemmP {
    const float *a;
    long long lda;
    int M, N, K;
    int H, W, Cin, cpt;          // conv: chunks (of 32 channels) per tap
    int nseg, seg_mode, seg_len;
    const float *w[LDM_MAX_SEG];
    const float *w2[LDM_MAX_SEG];
    const float *bias[LDM_MAX_SEG];
    const float *bias2[LDM_MAX_SEG];
    long long ldw;
    int act;
    float slope;
    const float *addend;
    long long ldadd;
    float *out;
    long long ldo;
    int o_mode, OH, OW, Cout;
    long long a_gstride, w_gstride, o_gstride, b_gstride;
    int use_table;                              // pointer-table mode: per-group weights / biases below
    int wide_ok;                                // rows output (and addend) 16-byte addressable: the stream kernel may use its wide epilogue
    int scat_ok;                                // scatter output (convT 2x2 / up x2) 16-byte addressable and M < 2^23: wide scatter epilogue
    float inv_ow;                               // 1 / OW (row -> (image row, x) by reciprocal multiply in that epilogue)
    const float *wtab[LDM_MAX_TABLE];
    const float *btab[LDM_MAX_TABLE];
    // bf16 training fusions of the wide epilogue (ldm_gemm_bf16_gate_fwd / _bwd), all bf16 [M, ldo] like `out`:
    //   gate forward : out = a * relu(b) AND out2 = a, out3 = b (the pre-activations the backward needs)
    //   gate backward: the GEMM result is dh; in2 = a, in3 = b;  out = dh * relu(b),  out2 = dh * a * (b > 0)
    void *out2, *out3;
    const void *in2, *in3;
    // bf16 sampling / decode: a bf16 addend [M, ldadd] for the bf16 output of the wide epilogue (VAE ResBlock skip: vae.py:65),
    // added in fp32 after the activation, before the one rounding to bf16
    const void *addend16;
};

// 64 B of zeros (one copy per translation unit): target of "absent operand" loads, so that optional
// biases and conv zero-padding need no branch around a load
static __device__ __attribute__((aligned(64))) float ldm_zero_block[16];

// Logical tile id -> (tile_m, tile_n), "banded" order: ids sweep a band of GM M-tiles column by column
// (m fastest inside the band, then n), so the ~64-96 tiles an XCD works on at one time form a compact
// GM x (64/GM) patch: every A panel is shared by the patch's N-tiles and every W panel by its M-tiles
// while they stream through K together -> both operands are served by the XCD's L2 instead of the
// fabric (measured with the n-fastest order: 13-19x the algorithmic fetch at C = 512 / 1024, W re-read
// from beyond L2 for every M-tile).
#ifndef LDM_BAND_M
#define LDM_BAND_M 8
#endif
constexpr int kBandM = LDM_BAND_M;      // (-DLDM_BAND_M=n: probe builds of tools/band_probe.py)
__device__ __forceinline__ void tile_from_id(int rem, int ntm, int ntn, int &tile_m, int &tile_n)
{
    const int band = rem / (kBandM * ntn);
    const int first = band * kBandM;
    const int gsz = (ntm - first) < kBandM ? (ntm - first) : kBandM;
    const int r2 = rem - band * kBandM * ntn;
    tile_n = r2 / gsz;
    tile_m = first + (r2 - tile_n * gsz);
}

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 5) + ((chunk ^ ((row >> 1) & 7)) << 2); }


// Epilogue of one BM x BN tile: bias, activation / gate, optional addend, rows | convT 2x2 | up2 addressing.
// acc layout (32x32 MFMA C/D map): column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
// `pre` holds addend values prefetched by the caller (o_mode == ROWS) when use_pre is set.
template <int TN>
struct EpiCols {
    float braw[LDM_MAX_SEG][TN];     // loaded, not yet summed: no arithmetic (hence no wait) at the load site
    float b2[TN];
    int q4[TN];
    long long ocol[TN];
};

// Column-side epilogue state (bias values, output columns): plain global loads, so the stream kernel
// issues them together with the addend prefetch, ahead of the tile's last K-step.
template <int WN, int TN, bool GATE>
__device__ __forceinline__ void gemm_epilogue_cols(const GemmP &p, EpiCols<TN> &c, int n0, int g, int seg_n, int wn, int r)
{
    const long long gcol = g * p.o_gstride;
    const long long gb = g * p.b_gstride;
    const float *tbias = p.use_table ? p.btab[g] : nullptr;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int nloc = n0 + (wn * TN + jn) * 32 + r;      // column inside this group's N
        int bidx = (p.seg_mode == LDM_SEG_N) ? nloc - seg_n * p.seg_len : nloc;
        int co = nloc;
        c.q4[jn] = 0;
        if (p.o_mode == LDM_O_CONVT2X2) {
            c.q4[jn] = nloc / p.Cout;
            co = nloc - c.q4[jn] * p.Cout;
            bidx = co;
        }
        c.ocol[jn] = gcol + co;
        // unconditional loads through SELECTED pointers (absent bias -> zero block): one straight-line
        // path with no branch, so all loads issue back to back and are waited for once, at their first
        // use after the MFMAs (a branch here makes hipcc drain the LDS-DMA queue with vmcnt(0))
        const bool ksum = p.seg_mode == LDM_SEG_K && !p.use_table;
        const float *first = p.use_table ? tbias : p.bias[seg_n];
        const long long off0 = (p.use_table ? 0 : gb) + bidx;
#pragma unroll
        for (int sgi = 0; sgi < LDM_MAX_SEG; ++sgi) {
            const float *base = ksum ? p.bias[sgi] : (sgi == 0 ? first : nullptr);
            const float *bp = base ? base + (ksum ? gb + bidx : off0) : ldm_zero_block;
            c.braw[sgi][jn] = *bp;
        }
        if (GATE) {
            const float *bq = p.bias2[seg_n] ? p.bias2[seg_n] + gb + bidx : ldm_zero_block;
            c.b2[jn] = *bq;
        } else {
            c.b2[jn] = 0.f;
        }
    }
}

// Addend prefetch for o_mode == ROWS (called ahead of the tile's last K-step): one 64-bit base per lane,
// then compile-time row offsets times the (uniform) row stride.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_prefetch_addend(const GemmP &p, float (&pre)[TM][TN][16], int m0, int n0, int g, int wm, int wn,
                                                     int r, int h)
{
    const int row0 = m0 + wm * TM * 32 + 4 * h;
    const int lda_ = (int)p.ldadd;
    const bool full = m0 + WM * TM * 32 <= p.M;
    const int rclamp = row0 < p.M ? row0 : p.M - 1;
    const float *base = p.addend + (long long)rclamp * p.ldadd + g * p.o_gstride + n0 + wn * TN * 32 + r;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int roff = im * 32 + (e & 3) + 8 * (e >> 2);
            if (!full) roff = row0 + roff < p.M ? roff : (p.M - 1 - rclamp);        // clamp to the last valid row
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) pre[im][jn][e] = base[roff * lda_ + jn * 32];
        }
}

// ACT is a template parameter: with the activation tested at run time inside the fully unrolled element loops, hipcc emitted
// two scalar branches per ELEMENT (338 branches, ~900 instructions per 128 x 128 tile of the plain instances) -- the wave-uniform
// switch now happens once per tile, in the wrappers below.
template <int WM, int WN, int TM, int TN, bool GATE, int ACT>
__device__ __forceinline__ void gemm_epilogue_act(const GemmP &p, f32x16 (&acc)[GATE ? 2 : 1][TM][TN], int m0, int wm, int h,
                                                  const EpiCols<TN> &c, const float (&pre)[TM][TN][16], bool use_pre)
{
    constexpr int NACC = GATE ? 2 : 1;
    float b1[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) b1[jn] = ((c.braw[0][jn] + c.braw[1][jn]) + c.braw[2][jn]) + c.braw[3][jn];
    auto value = [&](int im, int jn, int e) {
        float v = acc[0][im][jn][e] + b1[jn];
        if (GATE) {
            const float gt = acc[NACC - 1][im][jn][e] + c.b2[jn];
            v = v * fmaxf(gt, 0.f);
        } else if (ACT == LDM_ACT_RELU) {
            v = fmaxf(v, 0.f);
        } else if (ACT == LDM_ACT_LRELU) {
            v = v > 0.f ? v : v * p.slope;
        }
        // prefetched addend: consumed on every path (also for clamped, non-stored rows) so that the
        // compiler retires those loads here and never drains the LDS-DMA queue elsewhere for them
        if (use_pre) v += pre[im][jn][e];
        return v;
    };
    if (p.o_mode == LDM_O_ROWS) {
        // fast addressing: one 64-bit base per lane + (compile-time row offset) x (uniform 32-bit row stride)
        const int row0 = m0 + wm * TM * 32 + 4 * h;
        const int ldo_ = (int)p.ldo, lda_ = (int)p.ldadd;
        float *obase = p.out + (long long)row0 * p.ldo;
        const float *abase = p.addend ? p.addend + (long long)row0 * p.ldadd : nullptr;
        const bool late_add = !use_pre && p.addend != nullptr;
        if (m0 + WM * TM * 32 <= p.M) {                      // whole tile inside M: no per-element predicate
            // running pointers (one 64-bit add of the uniform row stride per row) instead of 32 distinct
            // "row offset x stride" products, which cost an SGPR each and spill
            float *orow = obase;
            const float *arow = abase;
#pragma unroll
            for (int im = 0; im < TM; ++im)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        float v = value(im, jn, e);
                        if (late_add) v += arow[c.ocol[jn]];
                        orow[c.ocol[jn]] = v;
                    }
                    // rows advance 0,1,2,3, 8,9,10,11, 16,... : +1 inside a group of four, +5 to the next group
                    const int step = ((e & 3) == 3) ? ((e == 15) ? 5 : 5) : 1;
                    orow += (long long)step * ldo_;
                    if (late_add) arow += (long long)step * lda_;
                }
        } else {
#pragma unroll
            for (int im = 0; im < TM; ++im)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int roff = im * 32 + (e & 3) + 8 * (e >> 2);
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        float v = value(im, jn, e);
                        if (row0 + roff < p.M) {
                            if (late_add) v += abase[roff * lda_ + c.ocol[jn]];
                            obase[roff * ldo_ + c.ocol[jn]] = v;
                        }
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int im = 0; im < TM; ++im) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + (wm * TM + im) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const bool live = m < p.M;
            const int xx = m % p.OW;
            const int yy = (m / p.OW) % p.OH;
            const long long bb = m / (p.OW * p.OH);
            const int ox2 = 2 * p.OW;
            const long long orow0 = (bb * 2 * p.OH + 2 * yy) * ox2 + 2 * xx;        // top-left fine pixel
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                float v = value(im, jn, e);
                if (live) {
                    if (p.o_mode == LDM_O_UP2) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const long long orow = orow0 + (d >> 1) * ox2 + (d & 1);
                            float o = v;
                            if (p.addend) o += p.addend[orow * p.ldadd + c.ocol[jn]];
                            p.out[orow * p.ldo + c.ocol[jn]] = o;
                        }
                    } else {
                        const long long orow = orow0 + (c.q4[jn] >> 1) * ox2 + (c.q4[jn] & 1);
                        if (p.addend) v += p.addend[orow * p.ldadd + c.ocol[jn]];
                        p.out[orow * p.ldo + c.ocol[jn]] = v;
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool GATE>
__device__ __forceinline__ void gemm_epilogue(const GemmP &p, f32x16 (&acc)[GATE ? 2 : 1][TM][TN], int m0, int wm, int h,
                                              const EpiCols<TN> &c, const float (&pre)[TM][TN][16], bool use_pre)
{
    if constexpr (GATE) {
        gemm_epilogue_act<WM, WN, TM, TN, GATE, LDM_ACT_GATE>(p, acc, m0, wm, h, c, pre, use_pre);
    } else {
        if (p.act == LDM_ACT_RELU) gemm_epilogue_act<WM, WN, TM, TN, GATE, LDM_ACT_RELU>(p, acc, m0, wm, h, c, pre, use_pre);
        else if (p.act == LDM_ACT_LRELU) gemm_epilogue_act<WM, WN, TM, TN, GATE, LDM_ACT_LRELU>(p, acc, m0, wm, h, c, pre, use_pre);
        else gemm_epilogue_act<WM, WN, TM, TN, GATE, LDM_ACT_NONE>(p, acc, m0, wm, h, c, pre, use_pre);
    }
}

// ---------------------------------------------------------------------------------------------------
// "Wide" epilogue of the stream kernel (o_mode == ROWS): the MFMA C/D map gives a lane ONE column and 16 rows, so a
// direct epilogue is 16 global_store_dword (+ 16 global_load_dword for an addend) per 32x32 tile and wave -- 128-byte
// row fragments, store-issue bound.  Here each wave passes its values through LDS (its own 1-KiB slices of the ring
// stage the tile has just finished with: no other wave writes them before this wave's next DMA) and leaves as
// 16-byte-per-lane rows: 4 global_store_dwordx4 (+ 4 global_load_dwordx4) per tile and wave.  Arithmetic and its
// order are unchanged (bias, activation / gate, then addend): results stay bit-identical to the direct epilogue.
//
// Scratch layout per wave: local row rho (0..31) of the strip sits in "slot" = rho with bits 0 and 2 swapped (rows
// rho and rho + 4 -- the two half-waves of one ds_write -- land in different bank halves); slots are rows of 32*TN
// floats packed into the wave's 1-KiB slices (slice i of wave w starts at float (i * 4 + w) * 256).
template <int TN>
struct WideLane {
    int cc, rsub;          // this lane's 16-byte chunk column inside the strip row, row inside an instruction's row group
    int rd_off;            // float offset of its b128 reads inside the wave's scratch (without the per-k constant)
    int wr_off;            // float offset of its b32 writes (without the per-(e, jn) constant)
};

template <int TN>
__device__ __forceinline__ WideLane<TN> wide_lane(int lane)
{
    constexpr int CPR = 8 * TN;
    WideLane<TN> w;
    w.cc = lane % CPR;
    w.rsub = lane / CPR;
    const int r = lane & 31, h = lane >> 5;
    if (TN == 1) {
        const int sw = (w.rsub & 2) | ((w.rsub & 1) << 2) | ((w.rsub >> 2) & 1);
        w.rd_off = sw * 32 + 4 * w.cc;
        w.wr_off = h * 32 + r;
    } else {
        w.rd_off = (w.rsub & 1) * 1024 + (w.rsub & 2) * 64 + 4 * w.cc;
        w.wr_off = h * 64 + r;
    }
    return w;
}

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_prefetch_addend_wide(const GemmP &p, float (&pre)[TM][TN][16], int m0, int n0, int g, int wm, int wn,
                                                          const WideLane<TN> &wl)
{
    constexpr int RPI = 8 / TN;                 // rows per instruction
    const int row0 = m0 + wm * TM * 32 + wl.rsub;
    const int lda_ = (int)p.ldadd;
    const bool full = m0 + WM * TM * 32 <= p.M;
    const int rclamp = row0 < p.M ? row0 : p.M - 1;
    const float *base = p.addend + (long long)rclamp * p.ldadd + g * p.o_gstride + n0 + wn * TN * 32 + 4 * wl.cc;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int k = 0; k < 4 * TN; ++k) {
            int roff = im * 32 + k * RPI;
            if (!full) roff = row0 + roff < p.M ? roff : (p.M - 1 - rclamp);
            const f32x4 v = *(const f32x4 *)(base + roff * lda_);
#pragma unroll
            for (int c = 0; c < 4; ++c) pre[im][(4 * k + c) >> 4][(4 * k + c) & 15] = v[c];
        }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of two bf16, round to nearest even, NaN stays NaN (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// GBW: the ReGLU backward in this epilogue (bf16 rows out, p.in2 / p.in3 / p.out2 set) -- a compile-time flag since round 3: as a
// run-time one its a / b buffers sat in every bf16-output instance and made the hot ones spill registers in their K loop
template <int WM, int WN, int TM, int TN, bool GATE, bool OBF, int ACT, bool SCAT = false, bool GBW = false>
__device__ __forceinline__ void gemm_epilogue_wide_act(const GemmP &p, f32x16 (&acc)[GATE ? 2 : 1][TM][TN], int m0, int n0, int g, int wm, int wn,
                                                   const EpiCols<TN> &c, const float (&pre)[TM][TN][16], bool use_pre,
                                                   const WideLane<TN> &wl, float *scratch /* stage base + wave * 256 */)
{
    constexpr int NACC = GATE ? 2 : 1;
    constexpr int RPI = 8 / TN;
    float b1[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) b1[jn] = ((c.braw[0][jn] + c.braw[1][jn]) + c.braw[2][jn]) + c.braw[3][jn];
    const int row0 = m0 + wm * TM * 32 + wl.rsub;
    const bool full = m0 + WM * TM * 32 <= p.M;
    const int ldo_ = (int)p.ldo;
    const long long oelem = (long long)row0 * p.ldo + g * p.o_gstride + n0 + wn * TN * 32 + 4 * wl.cc;     // same element offset for both output types
    float *obase = p.out + oelem;
    unsigned short *obase16 = (unsigned short *)p.out + oelem;
    float *wr = scratch + wl.wr_off;
    const float *rd = scratch + wl.rd_off;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    // scatter outputs (SCAT): coarse row m = q1 * OW + x lands on fine rows 4 OW q1 + 2 x + {0, 1, 2 OW, 2 OW + 1} -- all four for
    // up x2, the one its column quadrant q4 = n / Cout names for convT 2x2 (a lane's four columns share q4: Cout % 4 == 0)
    const int scol = g * (int)p.o_gstride + n0 + wn * TN * 32 + 4 * wl.cc;
    int s_q4 = 0, s_col = scol;
    if (SCAT && p.o_mode == LDM_O_CONVT2X2) {
        const int nloc = n0 + wn * TN * 32 + 4 * wl.cc;
        s_q4 = nloc / p.Cout;
        s_col = g * (int)p.o_gstride + nloc - s_q4 * p.Cout;
    }
    // passes over the tile: 1, or 3 for the gate forward that also saves its pre-activations (hid, a, b)
    const int npass = (GATE && OBF && p.out2 != nullptr) ? 3 : 1;
    static_assert(!GBW || (!GATE && OBF && !SCAT && ACT == LDM_ACT_NONE), "gate backward: plain bf16-output instance without activation");
    constexpr bool gate_bwd = GBW;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        unsigned short *ob16 = pass == 0 ? obase16 : (unsigned short *)(pass == 1 ? p.out2 : p.out3) + oelem;
        // gate backward: the saved pre-activations of this lane's row pieces, half a strip (2 TN pieces) at a time.  The loads of half
        // hh + 1 are issued BEFORE half hh is stored: loads issued behind stores cannot retire (vmcnt is in order) until those stores
        // are acknowledged -- a write round trip per strip with the loads at the strip's top.
        constexpr int HP = 2 * TN;                                        // row pieces per half strip
        u32x2_t ga[2][HP], gb[2][HP];
        auto load_ab = [&](int hh) {
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                int roff = (hh >> 1) * 32 + ((hh & 1) * HP + k) * RPI;
                if (!full && row0 + roff >= p.M) roff = p.M - 1 - row0;          // clamp (the result is not stored)
                ga[hh & 1][k] = *(const u32x2_t *)((const unsigned short *)p.in2 + oelem + (long long)roff * ldo_);
                gb[hh & 1][k] = *(const u32x2_t *)((const unsigned short *)p.in3 + oelem + (long long)roff * ldo_);
            }
        };
        if (gate_bwd) load_ab(0);
        const bool add16 = OBF && !GATE && !gate_bwd && p.addend16 != nullptr;
#pragma unroll
        for (int im = 0; im < TM; ++im) {
            // bf16 addend of this strip (4 TN row pieces of 4 columns): issued before the strip's LDS round trip, consumed after it
            u32x2_t a16[4 * TN];
            if constexpr (OBF && !GATE) {
                if (add16) {
                    const unsigned short *ab = (const unsigned short *)p.addend16 + (long long)row0 * p.ldadd + g * p.o_gstride + n0 + wn * TN * 32 + 4 * wl.cc;
#pragma unroll
                    for (int k = 0; k < 4 * TN; ++k) {
                        int roff = im * 32 + k * RPI;
                        if (!full && row0 + roff >= p.M) roff = p.M - 1 - row0;
                        a16[k] = *(const u32x2_t *)(ab + (long long)roff * p.ldadd);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    float v = acc[0][im][jn][e] + b1[jn];
                    if (GATE) {
                        const float gt = acc[NACC - 1][im][jn][e] + c.b2[jn];
                        if (pass == 0) v = v * fmaxf(gt, 0.f);
                        else if (pass == 2) v = gt;
                    } else if (ACT == LDM_ACT_RELU) {
                        v = fmaxf(v, 0.f);
                    } else if (ACT == LDM_ACT_LRELU) {
                        v = v > 0.f ? v : v * p.slope;
                    }
                    // local row (e & 3) + 8 (e >> 2) + 4 h  ->  slot (bits 0, 2 swapped) = h | (e & 2) | (e & 1) << 2 | 8 (e >> 2)
                    int off;
                    if (TN == 1) off = (e >> 2) * 1024 + ((e & 2) | ((e & 1) << 2)) * 32;
                    else off = (2 * (e >> 2) + (e & 1)) * 1024 + (e & 2) * 64 + jn * 32;
                    wr[off] = v;
                }
#pragma unroll
            for (int k = 0; k < 4 * TN; ++k) {
                const int hh = 2 * im + k / HP;                       // half strip of this piece
                if (gate_bwd && k % HP == 0 && hh + 1 < 2 * TM) load_ab(hh + 1);
                const int off = (TN == 1) ? k * 1024 : (k >> 1) * 2048 + (k & 1) * 64;
                f32x4 v = *(const f32x4 *)(rd + off);
                if (use_pre) {
#pragma unroll
                    for (int cix = 0; cix < 4; ++cix) v[cix] += pre[im][(4 * k + cix) >> 4][(4 * k + cix) & 15];
                }
                const int roff = im * 32 + k * RPI;
                if constexpr (SCAT) {
                    const int m = row0 + roff;
                    if (full || m < p.M) {
                        int q1 = (int)((float)m * p.inv_ow);                      // exact after one correction each way (m < 2^23)
                        int xx = m - q1 * p.OW;
                        q1 += xx >= p.OW ? 1 : (xx < 0 ? -1 : 0);
                        xx = m - q1 * p.OW;
                        const long long orow0 = 4ll * p.OW * q1 + 2 * xx;
                        const int ox2 = 2 * p.OW;
                        if (p.o_mode == LDM_O_UP2) {
                            f32x4 ad[4];
                            if (p.addend) {
#pragma unroll
                                for (int d = 0; d < 4; ++d) ad[d] = *(const f32x4 *)(p.addend + (orow0 + (d >> 1) * ox2 + (d & 1)) * p.ldadd + scol);
                            }
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                f32x4 o = v;
                                if (p.addend) o += ad[d];
                                *(f32x4 *)(p.out + (orow0 + (d >> 1) * ox2 + (d & 1)) * p.ldo + scol) = o;
                            }
                        } else {
                            const long long orow = orow0 + (s_q4 >> 1) * ox2 + (s_q4 & 1);
                            if (p.addend) v += *(const f32x4 *)(p.addend + orow * p.ldadd + s_col);
                            *(f32x4 *)(p.out + orow * p.ldo + s_col) = v;
                        }
                    }
                } else if (full || row0 + roff < p.M) {
                    if constexpr (OBF) {
                        if (gate_bwd) {
                            float da[4], db[4];
#pragma unroll
                            for (int cix = 0; cix < 4; ++cix) {
                                const unsigned wa = ga[hh & 1][k % HP][cix >> 1], wb = gb[hh & 1][k % HP][cix >> 1];
                                const float av = __uint_as_float((cix & 1) ? (wa & 0xFFFF0000u) : (wa << 16));
                                const float bv = __uint_as_float((cix & 1) ? (wb & 0xFFFF0000u) : (wb << 16));
                                da[cix] = v[cix] * fmaxf(bv, 0.f);
                                db[cix] = bv > 0.f ? v[cix] * av : 0.f;
                            }
                            *(u32x2_t *)(ob16 + roff * ldo_) = u32x2_t{pack_bf16x2(da[0], da[1]), pack_bf16x2(da[2], da[3])};
                            *(u32x2_t *)((unsigned short *)p.out2 + oelem + (long long)roff * ldo_) =
                                u32x2_t{pack_bf16x2(db[0], db[1]), pack_bf16x2(db[2], db[3])};
                        } else {
                            if constexpr (!GATE) {
                                if (add16) {
                                    v[0] += __uint_as_float(a16[k][0] << 16);
                                    v[1] += __uint_as_float(a16[k][0] & 0xFFFF0000u);
                                    v[2] += __uint_as_float(a16[k][1] << 16);
                                    v[3] += __uint_as_float(a16[k][1] & 0xFFFF0000u);
                                }
                            }
                            *(u32x2_t *)(ob16 + roff * ldo_) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        }
                    } else {
                        *(f32x4 *)(obase + roff * ldo_) = v;
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool GATE, bool OBF = false, bool SCAT = false, bool GBW = false>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmP &p, f32x16 (&acc)[GATE ? 2 : 1][TM][TN], int m0, int n0, int g, int wm, int wn,
                                                   const EpiCols<TN> &c, const float (&pre)[TM][TN][16], bool use_pre, const WideLane<TN> &wl, float *scratch)
{
    if constexpr (GBW) {
        gemm_epilogue_wide_act<WM, WN, TM, TN, false, true, LDM_ACT_NONE, false, true>(p, acc, m0, n0, g, wm, wn, c, pre, false, wl, scratch);
    } else if constexpr (SCAT) {
        if (p.act == LDM_ACT_RELU) gemm_epilogue_wide_act<WM, WN, TM, TN, false, false, LDM_ACT_RELU, true>(p, acc, m0, n0, g, wm, wn, c, pre, false, wl, scratch);
        else if (p.act == LDM_ACT_LRELU) gemm_epilogue_wide_act<WM, WN, TM, TN, false, false, LDM_ACT_LRELU, true>(p, acc, m0, n0, g, wm, wn, c, pre, false, wl, scratch);
        else gemm_epilogue_wide_act<WM, WN, TM, TN, false, false, LDM_ACT_NONE, true>(p, acc, m0, n0, g, wm, wn, c, pre, false, wl, scratch);
    } else if constexpr (GATE) {
        gemm_epilogue_wide_act<WM, WN, TM, TN, GATE, OBF, LDM_ACT_GATE>(p, acc, m0, n0, g, wm, wn, c, pre, use_pre, wl, scratch);
    } else {
        if (p.act == LDM_ACT_RELU) gemm_epilogue_wide_act<WM, WN, TM, TN, GATE, OBF, LDM_ACT_RELU>(p, acc, m0, n0, g, wm, wn, c, pre, use_pre, wl, scratch);
        else if (p.act == LDM_ACT_LRELU) gemm_epilogue_wide_act<WM, WN, TM, TN, GATE, OBF, LDM_ACT_LRELU>(p, acc, m0, n0, g, wm, wn, c, pre, use_pre, wl, scratch);
        else gemm_epilogue_wide_act<WM, WN, TM, TN, GATE, OBF, LDM_ACT_NONE>(p, acc, m0, n0, g, wm, wn, c, pre, use_pre, wl, scratch);
    }
}

}  // namespace ldmgemm
