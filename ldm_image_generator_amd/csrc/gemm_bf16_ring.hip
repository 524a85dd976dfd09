// bf16 NT GEMM with 256 x 256 output tiles and a four-stage LDS ring ("ring" kernel): out = act(A . W^T + bias) (+ addend).
//
// Why a second NT kernel.  The stream kernel (gemm_stream.hip, ET = 1) keeps two workgroups of 128 x 128 per CU with ONE K-step
// of look-ahead.  With bf16 operands a K-step is 16 MFMAs (512 cycles per wave), far shorter than an L2 / Infinity-Cache round
// trip under load: every step waits for its own LDS-DMA and the deep layers of the training step ran at 540-760 TFLOP/s
// (22-30 % of the bf16 matrix peak).  Probes: halving the DMA volume bought +8 %, removing the W stream +22 % -- latency, not
// bytes.  Two things fix that, and both need the whole CU for one workgroup:
//   * 256 x 256 tiles: 128 FLOP per staged byte instead of 64 (32 KiB of LDS-DMA per 1 024 matrix-pipe cycles at full rate,
//     about what one CU's L2 -> LDS path sustains);
//   * a ring of four 32-KiB stages (A [256][32] + W [256][32] bf16, 64-byte rows): three steps are in flight while one is
//     consumed, behind COUNTED vmcnt waits and one raw s_barrier per step; the K-steps of all tiles of a (persistent)
//     workgroup form one stream, so a tile's first steps were fetched under the previous tile's last ones and its epilogue.
// Eight waves of 128 x 64 (4 x 2 accumulator tiles of v_mfma_f32_32x32x16_bf16, 128 accumulator registers).  A step is two
// 16-k slices; the barrier sits BETWEEN the slices: slice 0's MFMAs cover the fragment reads of slice 1, slice 1's MFMAs
// cover the next step's slice-0 reads, so the matrix pipe only sees barrier skew.  Same slices in the same order as the
// stream kernel -> bit-identical results (asserted by tests/test_gpu_bf16.py).
//
// LDS image: 64-byte rows, 16-byte chunk c of row r stored at slot c ^ ((r >> 2) & 3) (applied to the DMA's SOURCE address,
// the destination of an LDS-DMA is lane-linear): the 8 lanes one LDS clock serves (rows r .. r + 7, same logical chunk) hit
// 8 distinct 16-byte slots of the 256-byte bank row.
// Epilogue: a wave's 32 x 32 (fp32 out) or 32 x 64 (bf16 out) pieces pass through its private 4-KiB scratch and leave as
// 16-byte-per-lane row segments of 128 bytes (the MFMA C/D map gives a lane one column).
#include "gemm_common.h"

using namespace ldmgemm;

namespace {

constexpr int RT = 256;                      // tile edge
constexpr int RNS = 4;                       // ring stages
constexpr int RSTAGE = 2 * RT * 64;          // bytes per stage: A rows then W rows, 64 B each
constexpr int RSCRATCH = 8 * 4096;           // epilogue scratch, 4 KiB per wave
constexpr int RPW = 4;                       // LDS-DMA instructions per wave and step (2 x A, 2 x W)
constexpr size_t RSMEM = (size_t)RNS * RSTAGE + RSCRATCH;      // 160 KiB: the whole LDS of a CU

int g_ring = 1;                              // 0 off, 1 auto (large problems), 2 whenever the shape is legal (tests)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ring_glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// waits until at most `younger` x RPW of this wave's vector-memory operations are outstanding (in-order retirement: everything
// older has landed); operations the compiler issues on its own in between (epilogue loads / stores) only make the wait stricter
__device__ __forceinline__ void ring_wait(int younger)
{
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool OBF>
__global__ __launch_bounds__(512, 1) void gemm_bf16_ring_kernel(const GemmP p, int ntm, int ntn, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char rlds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int nk = p.K >> 4;                                       // steps per tile (p.K counts 4-byte units: 32 bf16 = 16 units)
    const long long lda_b = p.lda * 4, ldw_b = p.ldw * 4;           // row strides in bytes
    const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total_steps = my_tiles * nk;
    const int seg_steps = p.seg_mode == LDM_SEG_K ? p.seg_len >> 4 : 0x7fffffff;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)rlds;

    auto tile_coords = [&](int i, int &m0, int &n0) {
        const int id = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, total_tiles);
        int tm_, tn_;
        tile_from_id(id, ntm, ntn, tm_, tn_);
        m0 = tm_ * RT;
        n0 = tn_ * RT;
    };

    // ---- loader: this wave moves rows [32 wave, 32 wave + 32) of the A tile and of the W tile, 16 rows per instruction --------
    int l_tile = 0, l_kt = 0, l_step = 0, l_seg = 0, l_kin = 0, l_nloc0 = 0;
    const char *a_cur = nullptr, *w_cur = nullptr;
    int a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + 16 * i + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        a_off[i] = row * (int)lda_b + c * 16;
        w_off[i] = row * (int)ldw_b + c * 16;
    }
    auto weight_rows = [&]() { w_cur = (const char *)p.w[l_seg] + (long long)l_nloc0 * ldw_b; };
    auto loader_setup = [&]() {
        int m0, n0;
        tile_coords(l_tile, m0, n0);
        l_seg = (p.seg_mode == LDM_SEG_N) ? n0 / p.seg_len : 0;
        l_nloc0 = (p.seg_mode == LDM_SEG_N) ? n0 - l_seg * p.seg_len : n0;
        l_kin = 0;
        a_cur = (const char *)p.a + (long long)m0 * lda_b;
        weight_rows();
    };
    // One step's DMA = four instructions per wave (part 0..3: A rows 0-15, W rows 0-15, A rows 16-31, W rows 16-31 of the
    // wave's 32), issued one at a time between the consumer's MFMAs, then the cursor advance.  EVERY step of the consumer issues
    // its four instructions: past the end of the stream they re-fetch the first rows of A / W into the stage nobody reads any
    // more, so the number of instructions in flight is the same at every wait (one fixed vmcnt, no branches in the K loop).
    unsigned ld_dst = 0;
    const char *ld_a = nullptr, *ld_w = nullptr;
    auto issue_begin = [&](bool live) {
        ld_dst = lds_base + (unsigned)((l_step & (RNS - 1)) * RSTAGE + wave * 2048);
        ld_a = live ? a_cur : (const char *)p.a;
        ld_w = live ? w_cur : (const char *)p.w[0];
    };
    auto issue_part = [&](int part) {
        const int i = part >> 1;
        if (part & 1) ring_glds16(ld_w + w_off[i], __builtin_amdgcn_readfirstlane(ld_dst + RT * 64 + i * 1024));
        else ring_glds16(ld_a + a_off[i], __builtin_amdgcn_readfirstlane(ld_dst + i * 1024));
    };
    auto issue_advance = [&](bool live) {
        ++l_step;
        if (!live) return;
        a_cur += 64;
        w_cur += 64;
        if (++l_kt == nk) {
            l_kt = 0;
            ++l_tile;
            if (l_tile < my_tiles) loader_setup();
        } else if (++l_kin == seg_steps) {
            l_kin = 0;
            ++l_seg;
            weight_rows();
        }
    };
    auto issue = [&](bool live) {
        issue_begin(live);
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(part);
        issue_advance(live);
    };

    // ---- consumer ---------------------------------------------------------------------------------------------------------------
    f32x16 acc[4][2];
    u32x4 fa0[4], fb0[2], fa1[4], fb1[2];
    const int sw = (r >> 2) & 3;
    const int lo0 = r * 64 + ((h ^ sw) << 4), lo1 = r * 64 + (((2 + h) ^ sw) << 4);      // this lane's chunk of slice 0 / 1 inside a 32-row block
    auto read_frags = [&](int step, int slice, u32x4 (&fa)[4], u32x4 (&fb)[2]) {
        const char *st = rlds + (step & (RNS - 1)) * RSTAGE;
        const int lo = slice ? lo1 : lo0;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const u32x4 *)(st + (wm * 128 + i * 32) * 64 + lo);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *(const u32x4 *)(st + RT * 64 + (wn * 64 + j * 32) * 64 + lo);
    };
    auto mma = [&](const u32x4 (&fa)[4], const u32x4 (&fb)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    };
    // one fragment / one MFMA at a time, in the order the MFMAs consume the fragments: A0 W0 W1 A1 A2 A3
    auto read_one = [&](const char *st, int slice, int idx, u32x4 (&fa)[4], u32x4 (&fb)[2]) {
        const int lo = slice ? lo1 : lo0;
        if (idx == 0) fa[0] = *(const u32x4 *)(st + (wm * 128) * 64 + lo);
        else if (idx == 1) fb[0] = *(const u32x4 *)(st + RT * 64 + (wn * 64) * 64 + lo);
        else if (idx == 2) fb[1] = *(const u32x4 *)(st + RT * 64 + (wn * 64 + 32) * 64 + lo);
        else fa[idx - 2] = *(const u32x4 *)(st + (wm * 128 + (idx - 2) * 32) * 64 + lo);
    };
    auto mma_one = [&](const u32x4 (&fa)[4], const u32x4 (&fb)[2], int q) {
        const int i = q >> 1, j = q & 1;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    };
    auto clear_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };

    // ---- prologue: the whole ring in flight, then step 0's first slice into registers ---------------------------------------------
    loader_setup();
#pragma unroll 1
    for (int s0 = 0; s0 < RNS; ++s0) issue(s0 < total_steps);
    ring_wait(RNS - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0, fa0, fb0);

    // epilogue scratch and lane roles
    char *scr = rlds + RNS * RSTAGE + wave * 4096;
    const int rsub = lane >> 3, cc = lane & 7;
    const int rrow = (rsub & 2) | ((rsub & 1) << 2) | ((rsub >> 2) & 1);        // row (inside a group of 8) whose slot this lane reads back

    int s = 0;
#pragma unroll 1
    for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
        int c_m0, c_n0;
        tile_coords(c_tile, c_m0, c_n0);
        clear_acc();
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++s) {
            const char *st_cur = rlds + (s & (RNS - 1)) * RSTAGE, *st_nxt = rlds + ((s + 1) & (RNS - 1)) * RSTAGE;
            // slice 0: one fragment read of slice 1 behind each of the first six MFMAs (all eight waves run in step: six reads in a
            // row from every wave at once is 48 KiB of LDS traffic in front of the second MFMA)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                mma_one(fa0, fb0, q);
                if (q < 6) read_one(st_cur, 1, q, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // step s + 1 has landed once only the two younger steps' instructions are outstanding; this wave's reads of stage s are
            // complete: after the barrier stage s may be refilled and stage s + 1 read
            ring_wait(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // slice 1: the DMA of step s + 4 (into stage s) and the next step's slice-0 fragments, one instruction per MFMA gap
            const bool live = s + RNS < total_steps;
            issue_begin(live);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                mma_one(fa1, fb1, q);
                if (q < 6) read_one(st_nxt, 0, q, fa0, fb0);        // past the last step: stale LDS, never used
                if (q & 1) issue_part(q >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_advance(live);
        }

        // ---- epilogue of this tile ----------------------------------------------------------------------------------------------
        const int seg_n = (p.seg_mode == LDM_SEG_N) ? c_n0 / p.seg_len : 0;
        float b1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = c_n0 + wn * 64 + j * 32 + r;
            const int bidx = (p.seg_mode == LDM_SEG_N) ? col - seg_n * p.seg_len : col;
            const bool ksum = p.seg_mode == LDM_SEG_K;
            float braw[LDM_MAX_SEG];
#pragma unroll
            for (int sg = 0; sg < LDM_MAX_SEG; ++sg) {
                const float *base = ksum ? p.bias[sg] : (sg == 0 ? p.bias[seg_n] : nullptr);
                braw[sg] = base ? base[bidx] : 0.f;
            }
            b1[j] = ((braw[0] + braw[1]) + braw[2]) + braw[3];
        }
        const int act = p.act;
        const float slope = p.slope;
        auto value = [&](float a, float b) {
            float v = a + b;
            if (act == LDM_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == LDM_ACT_LRELU) v = v > 0.f ? v : v * slope;
            return v;
        };
        const long long orow0 = (long long)(c_m0 + wm * 128);
        if constexpr (OBF) {
            unsigned short *o16 = (unsigned short *)p.out;
            const int ldo_ = (int)p.ldo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // 32 x 64 strip as bf16 into the scratch: row (e & 3) + 8 (e >> 2) + 4 h sits in slot h | (e & 2) | (e & 1) << 2 | 8 (e >> 2)
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float v = value(acc[i][j][e], b1[j]);
                        const int slot = h | (e & 2) | ((e & 1) << 2) | ((e >> 2) << 3);
                        *(unsigned short *)(scr + slot * 128 + (j * 32 + r) * 2) = (unsigned short)(pack_bf16x2(v, v) & 0xFFFFu);
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 v = *(const u32x4 *)(scr + (8 * k + rsub) * 128 + cc * 16);
                    const long long row = orow0 + i * 32 + 8 * k + rrow;
                    *(u32x4 *)(o16 + row * ldo_ + c_n0 + wn * 64 + cc * 8) = v;
                }
            }
        } else {
            float *o32 = (float *)p.out;
            const int ldo_ = (int)p.ldo, lda_ = (int)p.ldadd;
            const bool add = p.addend != nullptr;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const long long col = c_n0 + wn * 64 + j * 32 + cc * 4;
                    f32x4 ad[4];
                    if (add) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) ad[k] = *(const f32x4 *)(p.addend + (orow0 + i * 32 + 8 * k + rrow) * lda_ + col);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int slot = h | (e & 2) | ((e & 1) << 2) | ((e >> 2) << 3);
                        *(float *)(scr + slot * 128 + r * 4) = value(acc[i][j][e], b1[j]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f32x4 v = *(const f32x4 *)(scr + (8 * k + rsub) * 128 + cc * 16);
                        if (add) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += ad[k][q];
                        }
                        *(f32x4 *)(o32 + (orow0 + i * 32 + 8 * k + rrow) * ldo_ + col) = v;
                    }
                }
        }
    }
    // the trailing (dummy) DMA must have landed before the workgroup's LDS is handed to the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool OBF>
int ring_launch(const GemmP &p, hipStream_t st)
{
    static int state = 0, cus = 256;                         // 0 unknown, 1 usable, -1 the device refuses 160 KiB of LDS per workgroup
    auto kern = gemm_bf16_ring_kernel<OBF>;
    if (state == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        state = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RSMEM) == hipSuccess ? 1 : -1;
        (void)hipGetLastError();
    }
    if (state < 0) return 0;
    const int ntm = p.M / RT, ntn = p.N / RT;
    const int total = ntm * ntn;
    int grid = total < cus ? total : cus;
    if (grid > 8) grid &= ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), RSMEM, st, p, ntm, ntn, total);
    return 1;
}

}  // namespace

extern "C" int ldm_gemm_bf16_ring(int v)
{
    const int old = g_ring;
    if (v >= 0 && v <= 2) g_ring = v;
    return old;
}

// Takes the plain bf16 NT problems whose shape fits the 256 x 256 ring kernel; returns 1 if it launched.
int ldm_gemm_ring_dispatch_bf16(const GemmP &p, int groups, bool out_bf16, hipStream_t st)
{
    if (g_ring == 0 || groups != 1) return 0;
    if (p.M % RT || p.N % RT || (p.K & 15) || p.K < 16) return 0;
    if (p.seg_mode == LDM_SEG_N && p.nseg > 1 && p.seg_len % RT) return 0;
    if (p.seg_mode == LDM_SEG_K && p.nseg > 1 && (p.seg_len & 15)) return 0;
    if (p.act != LDM_ACT_NONE && p.act != LDM_ACT_RELU && p.act != LDM_ACT_LRELU) return 0;
    if (p.in2 || p.out2 || p.use_table) return 0;
    if (p.lda * 4 * RT >= (1ll << 31) || p.ldw * 4 * RT >= (1ll << 31)) return 0;
    if (out_bf16) {
        if (!ldm_aligned16(p.out) || p.ldo % 8 || p.addend) return 0;
    } else {
        if (!ldm_aligned16(p.out) || p.ldo % 4) return 0;
        if (p.addend && (!ldm_aligned16(p.addend) || p.ldadd % 4)) return 0;
    }
    const long long tiles = (long long)(p.M / RT) * (p.N / RT);
    // auto: at least ~3/4 of the CUs get a tile and a tile is at least 8 steps long (below that the two-workgroup stream kernel's
    // finer tiles balance better / its shorter prologue wins)
    if (g_ring == 1 && (tiles < 192 || p.K < 128)) return 0;
    return out_bf16 ? ring_launch<true>(p, st) : ring_launch<false>(p, st);
}
