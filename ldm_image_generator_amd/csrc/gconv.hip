// Grouped 3x3 convolution, 32 channels in / 32 out per group (unet.py:30,44), as its own fp32-MFMA kernel.
//
// Through the generic stream GEMM this op is an implicit-im2col problem with N = 32: every K-step (one tap) re-fetches
// a 128-pixel x 32-channel A tile for only 16 MFMAs per wave, i.e. 5 LDS-DMA instructions per 1024 MFMA cycles -- the
// kernel sits at MFMA-busy 0.50, paced by DMA issue.  Here a workgroup loads, ONCE per tile, the flattened pixel range
// [m0 - W - 1, m0 + 256 + W + 1) of its group's 32 channels (every 3x3 neighbour of pixel m is pixel m + dy*W + dx of
// that range) and, once per group, the 9 x 32 x 32 weights, then runs all nine taps out of LDS: about 5 DMA
// instructions per wave per 9216 MFMA cycles.  Image borders are handled at fragment-read time: an invalid neighbour
// reads a zero row.
//
// Same k-order as the stream / tile GEMM kernels (tap-major, then the (quarter, e, half-wave) map of their K-steps),
// so results are bit-identical to the generic path.
#include "gemm_common.h"

using namespace ldmgemm;

namespace {

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void glds16(const float *src, float *lds_dst)
{
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

// LDS fragment reads as inline asm: hipcc's waitcnt pass puts "s_waitcnt vmcnt(0)" in front of every LDS read that
// follows an LDS-DMA it cannot disambiguate -- which would wait for the NEXT tile's pixel block at the first fragment
// of the current one.  Reads issued this way are invisible to that pass; lds_wait() is the matching explicit wait and
// ties the fragment registers to it so that no consumer can be scheduled ahead of the wait.
__device__ __forceinline__ f32x4 lds_read16(unsigned byte_addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ void lds_wait(f32x4 &a, f32x4 &b)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

constexpr int NW = 8;                      // waves per workgroup: two per SIMD, so one wave's address / epilogue work
                                           // runs under the other's MFMAs (each wave is ONE dependent MFMA chain)
constexpr int TM = 1;                      // 32-row fragments per wave
constexpr int BM = NW * TM * 32;           // pixels per workgroup

// Persistent workgroups (one per CU): each walks a contiguous run of (group, pixel-tile) ids.  LDS: one zero row, TWO
// pixel blocks [NPp slots][32 floats] (the next tile's block is fetched by LDS-DMA while the current one is in the
// MFMAs -- its DMA instructions are spread over the MFMA quarters), the group's weights [9 taps][32 out][32 in]
// (re-fetched only when the run crosses into another group); all rows 128 B, 16-byte chunks XOR-swizzled by
// (row >> 1) & 7 like the GEMM tiles.
__global__ __launch_bounds__(NW * 64) void gconv3x3_kernel(const GemmP p, int ntm, int total, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = p.W, H = p.H;
    const int NP = BM + 2 * W + 2;
    const int NPp = (NP + 7) & ~7;                             // DMA instructions cover 8 slots: pad so the tail stays inside
    const int npieces = NPp / 8;                               // DMA instructions per pixel block (all waves together)
    float *Zs = lds, *Ab = lds + 32, *Ws = Ab + 2 * NPp * 32;
    const int first = (int)blockIdx.x * chunk;
    const int last = first + chunk < total ? first + chunk : total;
    if (first >= last) return;

    // one DMA instruction of a pixel block: piece k covers slots [8k, 8k + 8)
    auto a_piece = [&](int id, int k, float *dst) {
        const int g = id / ntm, m0 = (id - g * ntm) * BM;
        const int slot = 8 * k + (lane >> 3), cpos = lane & 7;
        int m = m0 - W - 1 + slot;
        m = m < 0 ? 0 : (m < p.M ? m : p.M - 1);
        glds16(p.a + g * p.a_gstride + (long long)m * p.lda + ((cpos ^ ((slot >> 1) & 7)) << 2), dst + k * 256);
    };
    auto w_load = [&](int g) {
        const float *wbase = p.w[0] + g * p.w_gstride;
        for (int q0 = wave * 64; q0 < 9 * 32 * 8; q0 += NW * 64) {    // 2304 chunks: exact multiple of 64
            const int q = q0 + lane;
            const int row = q >> 3, cpos = q & 7;                  // row = tap * 32 + n
            const int tap = row >> 5, n = row & 31;
            glds16(wbase + (long long)n * p.ldw + tap * 32 + ((cpos ^ ((row >> 1) & 7)) << 2), Ws + q0 * 4);
        }
    };

    if (t < 8) *(f32x4 *)(Zs + t * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur_g = first / ntm;
    w_load(cur_g);
    for (int k = wave; k < npieces; k += NW) a_piece(first, k, Ab);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int wsw = (r >> 1) & 7;                              // ((tap * 32 + r) >> 1) & 7
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = r * 32 + (((2 * j + h) ^ wsw) << 2);

#pragma unroll 1
    for (int id = first; id < last; ++id) {
        const int g = id / ntm, m0 = (id - g * ntm) * BM;
        const float *As = Ab + ((id - first) & 1) * NPp * 32;
        float *An = Ab + ((id - first + 1) & 1) * NPp * 32;
        if (g != cur_g) {                                      // the run crossed into another group: everyone is past its W reads
            cur_g = g;
            w_load(g);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        if (id + 1 < last)                                     // next tile's pixel block: lands under this tile's MFMAs
            for (int k = wave; k < npieces; k += NW) a_piece(id + 1, k, An);

        // ---- this lane's pixels (one per fragment): LDS row and swizzle of each of their nine neighbours ---------
        int aoff[TM][9], asw[TM][9];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wave * TM + i) * 32 + r;
            const int x = m % W, y = (m / W) % H;
            const bool live = m < p.M;
            const int slot0 = (wave * TM + i) * 32 + r + W + 1;    // the pixel's own slot
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const bool ok = live && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const int slot = ok ? slot0 + dy * W + dx : -1;    // outside the image: the zero row (in front of the blocks)
                aoff[i][tap] = ok ? (int)(As - lds) + slot * 32 : 0;
                asw[i][tap] = ok ? (slot >> 1) & 7 : 0;
            }
        }

        // bias / addend of the epilogue: issued now, consumed after the MFMAs (rows past M are clamped, never stored)
        const int col = g * (int)p.o_gstride + r;
        const float bias = *(p.bias[0] ? p.bias[0] + g * p.b_gstride + r : ldm_zero_block);
        const int row_first = m0 + wave * TM * 32 + 4 * h;
        const bool full = m0 + BM <= p.M;
        const int rclamp = row_first < p.M ? row_first : p.M - 1;
        const float *abase_e = p.addend ? p.addend + (long long)rclamp * p.ldadd + col : ldm_zero_block;
        const int lda_e = p.addend ? (int)p.ldadd : 0;
        float pre[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int roff = i * 32 + (e & 3) + 8 * (e >> 2);
                if (!full) roff = row_first + roff < p.M ? roff : (p.M - 1 - rclamp);
                pre[i][e] = abase_e[roff * lda_e];
            }

        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        // Fragments are fetched one quarter of a tap ahead of their use; each accumulator is ONE dependent MFMA chain in
        // the generic GEMM's k-order, the two chains of a wave interleave.  sched_barrier keeps hipcc from sinking the
        // LDS reads behind the MFMA group that is meant to cover their latency.
        const unsigned ws_addr = (unsigned)(size_t)(lptr_t)Ws, lds_addr = (unsigned)(size_t)(lptr_t)lds;
        auto frag = [&](int idx, f32x4 (&af)[TM], f32x4 &bf) {
            const int tap = idx >> 2, j = idx & 3;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = lds_read16(lds_addr + 4u * (unsigned)(aoff[i][tap] + (((2 * j + h) ^ asw[i][tap]) << 2)));
            bf = lds_read16(ws_addr + 4u * (unsigned)(tap * 1024 + woff[j]));
        };
        auto mma = [&](const f32x4 (&af)[TM], const f32x4 &bf) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], acc[i], 0, 0, 0);
        };
        static_assert(TM == 1, "lds_wait ties exactly one A fragment and one W fragment");
        f32x4 a0[TM], a1[TM], b0, b1;
        frag(0, a0, b0);
#pragma unroll
        for (int idx = 0; idx < 36; idx += 2) {
            lds_wait(a0[0], b0);                       // fragments of quarter idx (issued one MFMA group ago)
            frag(idx + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);                // reads stay in front of the MFMA group that hides them
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            lds_wait(a1[0], b1);
            if (idx + 2 < 36) frag(idx + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }

        // the next block has landed for this wave and this wave is done reading the current one; after the barrier that
        // holds for all waves.  Placed BEFORE the epilogue so that its stores drain under the next tile's MFMAs.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- epilogue (MFMA C/D map: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h) -----------------------
        float *obase = p.out + (long long)row_first * p.ldo + col;
        const int ldo_e = (int)p.ldo;
        const float slope = p.act == LDM_ACT_LRELU ? p.slope : 1.f;
        const bool relu = p.act == LDM_ACT_RELU, has_add = p.addend != nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int roff = i * 32 + (e & 3) + 8 * (e >> 2);
                float v = acc[i][e] + bias;
                const float vr = fmaxf(v, 0.f), vl = v > 0.f ? v : v * slope;
                v = relu ? vr : vl;
                if (has_add) v += pre[i][e];
                acc[i][e] = v;
            }
        if (full) {                                            // whole tile inside M: no per-row predicate
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) obase[(i * 32 + (e & 3) + 8 * (e >> 2)) * ldo_e] = acc[i][e];
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int roff = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (row_first + roff < p.M) obase[roff * ldo_e] = acc[i][e];
                }
        }
    }
}

// ---- software-pipelined variant (M a multiple of 256: every tile is full) ---------------------------------------------
// With one dependent MFMA chain per wave, a wave can only do other work in the gaps between its MFMAs, and the per-tile
// workgroup barrier keeps all waves of a workgroup in phase -- so the ~2000 instructions of per-tile set-up and epilogue
// of the kernel above are exposed.  Here they are cut into chunks of about a dozen instructions and placed behind each
// 4-MFMA group of the K loop (where the matrix pipe is busy for another 64 cycles anyway):
//   * quarter 0 .. 11  : the NEXT tile's pixel coordinates (reciprocal-multiply division) and its nine neighbour slots,
//   * quarter 12 .. 27 : the NEXT tile's addend loads,
//   * quarter 20 .. 35 : the PREVIOUS tile's 16 output stores (its values were finalised right after its MFMAs).
struct GcTile {
    int aoff[9], asw[9];
    float pre[16];
    float bias;
    float *obase;
};

// WIDE: addend and output travel as 16-byte row pieces (4 + 4 VMEM instructions per tile and wave instead of 16 + 16 dwords): the
// finished values cross from the MFMA C/D map to rows through a private 32 x 36-float LDS slice per wave.  Same arithmetic.
constexpr int GC_SROW = 36;
template <bool WIDE>
__global__ __launch_bounds__(NW * 64) void gconv3x3_pipe_kernel(const GemmP p, int ntm, int total, int chunk, float inv_w, float inv_h)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = p.W, H = p.H;
    const int NP = BM + 2 * W + 2;
    const int NPp = (NP + 7) & ~7;
    const int npieces = NPp / 8;
    float *Zs = lds, *Ab = lds + 32, *Ws = Ab + 2 * NPp * 32;
    float *Sw = Ws + 9 * 32 * 32 + wave * (32 * GC_SROW);       // WIDE: this wave's transpose slice
    const int rsub = lane >> 3, cc = lane & 7;                 // WIDE: row inside an 8-row group, 16-byte chunk of the 32 columns
    const int first = (int)blockIdx.x * chunk;
    const int last = first + chunk < total ? first + chunk : total;
    if (first >= last) return;

    auto a_piece = [&](int id, int k, float *dst) {
        const int g = id / ntm, m0 = (id - g * ntm) * BM;
        const int slot = 8 * k + (lane >> 3), cpos = lane & 7;
        int m = m0 - W - 1 + slot;
        m = m < 0 ? 0 : (m < p.M ? m : p.M - 1);
        glds16(p.a + g * p.a_gstride + (long long)m * p.lda + ((cpos ^ ((slot >> 1) & 7)) << 2), dst + k * 256);
    };
    auto w_load = [&](int g) {
        const float *wbase = p.w[0] + g * p.w_gstride;
        for (int q0 = wave * 64; q0 < 9 * 32 * 8; q0 += NW * 64) {
            const int q = q0 + lane;
            const int row = q >> 3, cpos = q & 7;
            const int tap = row >> 5, n = row & 31;
            glds16(wbase + (long long)n * p.ldw + tap * 32 + ((cpos ^ ((row >> 1) & 7)) << 2), Ws + q0 * 4);
        }
    };
    // exact x / d for 0 <= x < 2^24 via a float reciprocal and one correction step each way
    auto divmod = [](int x, int d, float inv, int &q, int &rem) {
        q = (int)((float)x * inv);
        rem = x - q * d;
        const bool hi = rem >= d, lo = rem < 0;
        q += hi ? 1 : (lo ? -1 : 0);
        rem += hi ? -d : (lo ? d : 0);
    };
    // per-tile pieces -------------------------------------------------------------------------------------------------
    int nx = 0, ny = 0, nrow = 0, nslot0 = 0, nbuf = 0;      // next tile's pixel of this lane
    const float *n_abase = nullptr;
    int n_lda = 0, n_g = 0, n_m0 = 0;
    // three chunks, one per MFMA group: tile -> (group, first pixel) and x; y; output / addend pointers
    auto next_coords0 = [&](int id) {
        n_g = id / ntm;
        n_m0 = (id - n_g * ntm) * BM;
        divmod(n_m0 + wave * 32 + r, W, inv_w, nrow, nx);
    };
    auto next_coords1 = [&](int bufsel) {
        int tmp;
        divmod(nrow, H, inv_h, tmp, ny);
        nslot0 = wave * 32 + r + W + 1;
        nbuf = 32 + bufsel * NPp * 32;                        // float offset of the pixel block inside lds
    };
    auto next_coords2 = [&](GcTile &T) {
        const int col = n_g * (int)p.o_gstride + r;
        const long long row_first = n_m0 + wave * 32 + 4 * h;
        T.bias = *(p.bias[0] ? p.bias[0] + n_g * p.b_gstride + r : ldm_zero_block);
        if (WIDE) {
            const long long rw = n_m0 + wave * 32 + rsub;
            const int cw = n_g * (int)p.o_gstride + 4 * cc;
            T.obase = p.out + rw * p.ldo + cw;
            n_abase = p.addend ? p.addend + rw * p.ldadd + cw : ldm_zero_block;
        } else {
            T.obase = p.out + row_first * p.ldo + col;
            n_abase = p.addend ? p.addend + row_first * p.ldadd + col : ldm_zero_block;
        }
        n_lda = p.addend ? (int)p.ldadd : 0;
    };
    auto next_tap = [&](int tap, GcTile &T) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const bool ok = (unsigned)(ny + dy) < (unsigned)H && (unsigned)(nx + dx) < (unsigned)W;
        const int slot = nslot0 + dy * W + dx;
        T.aoff[tap] = ok ? nbuf + slot * 32 : 0;              // outside the image: the zero row at lds[0]
        T.asw[tap] = ok ? (slot >> 1) & 7 : 0;
    };
    auto next_pre = [&](int e, GcTile &T) { T.pre[e] = n_abase[((e & 3) + 8 * (e >> 2)) * n_lda]; };
    auto next_pre4 = [&](int k, GcTile &T) {                  // WIDE: rows 8 k + rsub, four columns
        const f32x4 v = *(const f32x4 *)(n_abase + 8 * k * n_lda);
#pragma unroll
        for (int c = 0; c < 4; ++c) T.pre[4 * k + c] = v[c];
    };

    if (t < 8) *(f32x4 *)(Zs + t * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur_g = first / ntm;
    w_load(cur_g);
    for (int k = wave; k < npieces; k += NW) a_piece(first, k, Ab);
    GcTile C, N;
    next_coords0(first);
    next_coords1(0);
    next_coords2(C);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) next_tap(tap, C);
    if (WIDE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) next_pre4(k, C);
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) next_pre(e, C);
    }
    N = C;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int wsw = (r >> 1) & 7;
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = r * 32 + (((2 * j + h) ^ wsw) << 2);
    const unsigned ws_addr = (unsigned)(size_t)(lptr_t)Ws, lds_addr = (unsigned)(size_t)(lptr_t)lds;
    const float slope = p.act == LDM_ACT_LRELU ? p.slope : 1.f;
    const bool relu = p.act == LDM_ACT_RELU, has_add = p.addend != nullptr;
    const int ldo_e = (int)p.ldo;

    float outv[16];                                           // previous tile's finished values, stored under this tile's MFMAs
    float *obase_prev = nullptr;
    bool pending = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) outv[e] = 0.f;

#pragma unroll 1
    for (int id = first; id < last; ++id) {
        const int g = id / ntm;
        float *An = Ab + ((id - first + 1) & 1) * NPp * 32;
        if (g != cur_g) {
            cur_g = g;
            w_load(g);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const int idn = id + 1 < last ? id + 1 : id;           // clamped: the last tile "prepares" itself again (unused)
        if (id + 1 < last)
            for (int k = wave; k < npieces; k += NW) a_piece(id + 1, k, An);

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        auto frag = [&](int idx, f32x4 &af, f32x4 &bf) {
            const int tap = idx >> 2, j = idx & 3;
            af = lds_read16(lds_addr + 4u * (unsigned)(C.aoff[tap] + (((2 * j + h) ^ C.asw[tap]) << 2)));
            bf = lds_read16(ws_addr + 4u * (unsigned)(tap * 1024 + woff[j]));
        };
        auto mma = [&](const f32x4 &af, const f32x4 &bf) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        };
        // the chunk of side work that rides behind MFMA group q
        auto extra = [&](int q) {
            if (q == 0) next_coords0(idn);
            if (q == 1) next_coords1((id - first + 1) & 1);
            if (q == 2) next_coords2(N);
            if (q >= 3 && q <= 11) next_tap(q - 3, N);
            if (WIDE) {
                if (q >= 12 && q <= 15) next_pre4(q - 12, N);
                if (q >= 20 && q <= 23 && pending)
                    *(f32x4 *)(obase_prev + 8 * (q - 20) * ldo_e) = f32x4{outv[4 * (q - 20)], outv[4 * (q - 20) + 1], outv[4 * (q - 20) + 2], outv[4 * (q - 20) + 3]};
            } else {
                if (q >= 12 && q <= 27) next_pre(q - 12, N);
                if (q >= 20 && pending) obase_prev[(((q - 20) & 3) + 8 * ((q - 20) >> 2)) * ldo_e] = outv[q - 20];
            }
        };
        f32x4 a0, a1, b0, b1;
        frag(0, a0, b0);
#pragma unroll
        for (int idx = 0; idx < 36; idx += 2) {
            lds_wait(a0, b0);
            frag(idx + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            extra(idx);
            __builtin_amdgcn_sched_barrier(0);
            lds_wait(a1, b1);
            if (idx + 2 < 36) frag(idx + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            extra(idx + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();

        // finalise this tile's values (stored under the next tile's MFMAs, or by the flush below)
        if (WIDE) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[e] + C.bias;
                const float vr = fmaxf(v, 0.f), vl = v > 0.f ? v : v * slope;
                Sw[((e & 3) + 8 * (e >> 2) + 4 * h) * GC_SROW + r] = relu ? vr : vl;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = *(const f32x4 *)(Sw + (8 * k + rsub) * GC_SROW + 4 * cc);
#pragma unroll
                for (int c = 0; c < 4; ++c) outv[4 * k + c] = has_add ? o[c] + C.pre[4 * k + c] : o[c];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[e] + C.bias;
                const float vr = fmaxf(v, 0.f), vl = v > 0.f ? v : v * slope;
                v = relu ? vr : vl;
                if (has_add) v += C.pre[e];
                outv[e] = v;
            }
        }
        obase_prev = C.obase;
        pending = true;
        C = N;
    }
    if (WIDE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4 *)(obase_prev + 8 * k * ldo_e) = f32x4{outv[4 * k], outv[4 * k + 1], outv[4 * k + 2], outv[4 * k + 3]};
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) obase_prev[((e & 3) + 8 * (e >> 2)) * ldo_e] = outv[e];
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
// dW[g][co][tap][ci] = sum over pixels m of dy[m][32 g + co] * x[nbr(m, tap)][32 g + ci]  (autograd of unet.py:30,44).
// A "TN" MFMA problem per group: both operands have the contraction (the pixels) on their rows.  A workgroup owns one
// group and a run of pixels; per 128-pixel tile it loads the x block with halo and the dy block into LDS (rows as they
// lie in memory), and each wave walks its 32 pixels two at a time: ONE dy value per lane (row = co) against NINE x
// values (column = ci, the nine neighbours; an out-of-image neighbour reads the zero row) -> nine independent
// accumulator tiles, one per tap.  No transposed im2col copy (9x the activation), no per-group launches.
// Each wave writes its own partial plane: out[(split * 4 + wave)][C][288]; the caller sums the planes in fixed order.
constexpr int WG_BM = 128;

__global__ __launch_bounds__(256, 2) void gconv3x3_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                float *__restrict__ out, int M, int H, int W, int C, int ms,
                                                                float inv_w, float inv_h)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int g = (int)blockIdx.x, split = (int)blockIdx.y;
    const int NP = WG_BM + 2 * W + 2;
    const int NPp = (NP + 7) & ~7;
    float *Zs = lds, *Xs = lds + 32, *Ds = Xs + NPp * 32;       // zero row | x block [NPp][32] | dy block [128][32]
    if (t < 8) *(f32x4 *)(Zs + t * 4) = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

    const float *xg = x + g * 32, *dg = dy + g * 32;
    const int row_lo = split * ms;
#pragma unroll 1
    for (int m0 = row_lo; m0 < row_lo + ms; m0 += WG_BM) {
        __syncthreads();                                       // everyone is done with the previous tile's blocks
        for (int k = wave; k < NPp / 8; k += 4) {             // x block: 8 slots (1 KiB) per DMA instruction
            const int slot = 8 * k + (lane >> 3);
            int m = m0 - W - 1 + slot;
            m = m < 0 ? 0 : (m < M ? m : M - 1);
            glds16(xg + (long long)m * C + (lane & 7) * 4, Xs + k * 256);
        }
        for (int k = wave; k < WG_BM / 8; k += 4)
            glds16(dg + (long long)(m0 + 8 * k + (lane >> 3)) * C + (lane & 7) * 4, Ds + k * 256);
        // this lane's first pixel of the tile: m0 + 32 wave + h  (then + 2 per step)
        int px, py, row, tmp;
        {
            const int m = m0 + wave * 32 + h;
            row = (int)((float)m * inv_w);
            px = m - row * W;
            row += px >= W ? 1 : (px < 0 ? -1 : 0);
            px += px >= W ? -W : (px < 0 ? W : 0);
            tmp = (int)((float)row * inv_h);
            py = row - tmp * H;
            py += py >= H ? -H : (py < 0 ? H : 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const unsigned xs_addr = (unsigned)(size_t)(lptr_t)Xs, ds_addr = (unsigned)(size_t)(lptr_t)Ds;
        int slot = wave * 32 + h + W + 1;                      // own slot of the lane's current pixel
#pragma unroll 2
        for (int st = 0; st < 16; ++st) {
            const float a = Ds[(wave * 32 + 2 * st + h) * 32 + r];
            const bool up = py > 0, dn = py < H - 1, lf = px > 0, rt = px < W - 1;
            float bv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int ddy = k / 3 - 1, ddx = k % 3 - 1;
                const bool ok = (ddy < 0 ? up : (ddy > 0 ? dn : true)) && (ddx < 0 ? lf : (ddx > 0 ? rt : true));
                const int sl = slot + ddy * W + ddx;
                bv[k] = ok ? Xs[sl * 32 + r] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[k], acc[k], 0, 0, 0);
            slot += 2;
            px += 2;
            if (px >= W) {                                     // W >= 2: at most one wrap per step
                px -= W;
                py += 1;
                if (py >= H) py = 0;
            }
        }
        (void)xs_addr; (void)ds_addr;
    }

    // acc[k] element e of lane (ci = r, h): row co = (e & 3) + 8 (e >> 2) + 4 h  ->  out[plane][32 g + co][32 k + ci]
    float *ob = out + ((long long)(split * 4 + wave) * C + g * 32) * 288 + r;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) ob[((e & 3) + 8 * (e >> 2) + 4 * h) * 288 + 32 * k] = acc[k][e];
}

}  // namespace

int g_gconv_wide = 1;          // 0: dword addend loads / output stores in the pipelined kernel (A/B tests; see ldm_gemm_wide_epilogue)

// Returns 1 if the problem is a 32-in / 32-out-per-group 3x3 convolution this kernel covers (and launches it), else 0.
int ldm_gconv3x3_dispatch(const GemmP &p, int groups, bool gate, int amode, hipStream_t st)
{
    if (gate || amode != LDM_A_CONV3X3 || groups < 2 || p.N != 32 || p.Cin != 32 || p.K != 288) return 0;
    if (p.o_mode != LDM_O_ROWS || p.use_table || p.nseg != 1 || p.a_gstride != 32 || p.W > 64) return 0;
    if (p.lda > 0x7fffff || p.ldo > 0x7fffff || p.o_gstride > 0x7fffff) return 0;
    const int ntm = (p.M + BM - 1) / BM;
    const long long total = (long long)ntm * groups;
    if (total > 0x3fffffffLL) return 0;
    const int NPp = (BM + 2 * p.W + 2 + 7) & ~7;
    const size_t smem = ((size_t)32 + 2 * (size_t)NPp * 32 + 9 * 32 * 32) * sizeof(float);
    static LdmLdsOptIn opt_a, opt_b, opt_c;
    (void)opt_a((const void *)gconv3x3_kernel, 150 * 1024);
    (void)opt_b((const void *)gconv3x3_pipe_kernel<false>, 150 * 1024);
    (void)opt_c((const void *)gconv3x3_pipe_kernel<true>, 160 * 1024);
    const int cus = ldm_cu_count();
    const int wgs = (int)(total < cus ? total : cus);
    const int chunk = (int)((total + wgs - 1) / wgs);
    const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
    if (p.M % BM == 0 && p.M < (1 << 24) && p.ldo * 64ll < 0x7fffffffLL) {        // every tile full: the software-pipelined variant
        const size_t smem_w = smem + (size_t)NW * 32 * GC_SROW * sizeof(float);
        const bool wide = g_gconv_wide && smem_w <= 160 * 1024 && ldm_aligned16(p.out) && p.ldo % 4 == 0 && p.o_gstride % 4 == 0 &&
                          (!p.addend || (ldm_aligned16(p.addend) && p.ldadd % 4 == 0));
        if (wide)
            ldm_launch(gconv3x3_pipe_kernel<true>, dim3(grid), dim3(NW * 64), smem_w, st, p, ntm, (int)total, chunk, 1.0f / (float)p.W,
                               1.0f / (float)p.H);
        else
            ldm_launch(gconv3x3_pipe_kernel<false>, dim3(grid), dim3(NW * 64), smem, st, p, ntm, (int)total, chunk, 1.0f / (float)p.W,
                               1.0f / (float)p.H);
    } else
        ldm_launch(gconv3x3_kernel, dim3(grid), dim3(NW * 64), smem, st, p, ntm, (int)total, chunk);
    return 1;
}

extern "C" int ldm_gconv3x3_wgrad_f32(const float *x, const float *dy, float *out_planes, int B, int H, int W, int C, int splits, void *stream)
{
    LDM_REQUIRE(x && dy && out_planes, "ldm_gconv3x3_wgrad_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W >= 2 && W <= 96 && C >= 32 && C % 32 == 0, "ldm_gconv3x3_wgrad_f32: bad shape (2 <= W <= 96, C %% 32 == 0)");
    const long long M = (long long)B * H * W;
    LDM_REQUIRE(M < (1 << 24) && splits >= 1 && splits <= 65535 && M % ((long long)splits * WG_BM) == 0,
                "ldm_gconv3x3_wgrad_f32: B*H*W=%lld must split into %d runs of a multiple of 128 pixels", M, splits);
    LDM_REQUIRE(ldm_aligned16(x) && ldm_aligned16(dy), "ldm_gconv3x3_wgrad_f32: unaligned pointer");
    const int NPp = (WG_BM + 2 * W + 2 + 7) & ~7;
    const size_t smem = ((size_t)32 + (size_t)NPp * 32 + WG_BM * 32) * sizeof(float);
    static LdmLdsOptIn opt_in;
    (void)opt_in((const void *)gconv3x3_wgrad_kernel, 80 * 1024);
    void *rec = ldm_prof_begin(LDM_PROF_GCONV_WG, 2.0 * (double)M * C * 288.0, (hipStream_t)stream,
                               8.0 * (double)M * C + 4.0 * splits * (double)C * 288.0);
    ldm_launch(gconv3x3_wgrad_kernel, dim3(C / 32, splits), dim3(256), smem, (hipStream_t)stream, x, dy, out_planes, (int)M, H, W, C,
                       (int)(M / splits), 1.0f / (float)W, 1.0f / (float)H);
    ldm_prof_end(rec, (hipStream_t)stream);
    LDM_CHECK_LAUNCH("ldm_gconv3x3_wgrad_f32");
    return LDM_OK;
}
