// NT GEMM with 256-row output tiles, ONE workgroup of eight waves per CU and a four-stage LDS ring ("ring" kernel):
// out = act(A . W^T + bias) (+ addend), or the ReGLU pair a(x) * relu(b(x)).  bf16 operands (training step) and exact fp32
// (sampling) share one schedule.
//
// Why next to the stream kernel (gemm_stream.hip: two workgroups of 128 x 128 per CU, one K-step of look-ahead).
//   bf16: a K-step is 16 MFMAs (512 cycles per wave), far shorter than an L2 / Infinity-Cache round trip under load: every step
//     waited out its own LDS-DMA and the deep layers ran at 540-760 TFLOP/s.  Probes: half the W bytes +8 %, no W bytes +22 % --
//     latency, not volume.
//   fp32: the K-steps are long, but at C = 128 / 256 a tile is only 4-8 of them and every tile pays ~3.9 k cycles of tile switch
//     plus ~1.7 k cycles of LDS-DMA issue per K-step pair (probe builds).
// One workgroup per CU buys: 256 x 256 (or 256 x 128) tiles = half (three quarters) of the staged bytes and LDS-DMA instructions
// per MFMA and a quarter (half) of the tile switches; a ring of four 32-KiB stages of 64-byte rows (32 bf16 / 16 fp32 k): three
// steps in flight while one is consumed, behind ONE fixed counted vmcnt and one raw s_barrier per step.  The K-steps of all tiles of
// a (persistent) workgroup form one stream; past its end every step still issues its LDS-DMA instructions (re-fetching the first
// rows of A / W into the stage nobody reads any more), so no wait ever branches.
// Eight waves as 2 (M) x 4 (N), each 128 rows x 32 NJ columns (4 x NJ accumulator tiles of 32 x 32).  A step is two slices (16 bf16
// / 8 fp32 k); the barrier sits BETWEEN the slices: each slice's MFMAs cover the fragment reads of the next one, and the step's
// LDS-DMA instructions are pinned one per MFMA gap behind the barrier.  Slices, k pairing and order are those of the stream
// kernel -> bit-identical results (tests/test_gpu_bf16.py, tests/test_gpu_kernels.py).
// GATE: the W stage holds 128 rows of the "a" weights and 128 rows of the "b" weights of the same 128 hidden columns; a wave's
// two accumulator columns are a and b of ITS 32 hidden columns, and the epilogue stores (a + ba) * relu(b + bb).
//
// LDS image: 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3) (applied to the DMA's SOURCE address; an LDS-DMA destination is
// lane-linear): the lanes one LDS clock serves hit distinct 16-byte slots of the 256-byte bank row.
// Epilogue: 32 x 32 (fp32 out) / 32 x 64 (bf16 out) pieces through a private 4-KiB scratch per wave, 16 bytes per lane, 128-byte rows.
#include "gemm_common.h"
#include <type_traits>

using namespace ldmgemm;

namespace {

constexpr int RT = 256;                      // rows of A per tile
constexpr int RNS = 4;                       // ring stages
constexpr int RSTAGE = 32768;                // bytes per stage: A rows at 0, W rows at 16384, 64 B each
constexpr int RSCRATCH = 8 * 4096;           // epilogue scratch, 4 KiB per wave
constexpr size_t RSMEM = (size_t)RNS * RSTAGE + RSCRATCH;      // 160 KiB: the whole LDS of a CU

int g_ring = 1;                              // 0 off, 1 auto (large problems), 2 whenever the shape is legal, 3 = 2 on at most 8 workgroups
                                             // (tests: many tiles per workgroup out of small problems -- the stream across tile boundaries)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA instruction: 64 lanes x 16 bytes from base + off (base: wave-uniform, in SGPRs; off: this lane's 32-bit byte offset, a loop
// invariant) to LDS at lds_dst + 16 lane.  The scalar-base form needs no 64-bit vector add per instruction (the earlier form took the
// whole address in a VGPR pair: two VALU instructions per DMA instruction, eight per wave and step, in a loop whose issue slots are
// what the bf16 instances run out of)
// (the address-in-a-VGPR-pair form: kept for the bf16 instances, where the scalar-base form measured 0.8 % SLOWER over a sampling pass --
// 446.9 vs 443.5 ms, two alternations on one box -- while it is worth 3 % to the fp32 instances)
__device__ __forceinline__ void ring_glds16v(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void ring_glds16(const char *base, int off, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(lds_dst)
                 : "memory");
}

// waits until at most N of this wave's vector-memory operations are outstanding (in-order retirement: everything older has landed);
// operations the compiler issues on its own in between (epilogue loads / stores) only make the wait stricter
template <int N>
__device__ __forceinline__ void ring_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// a 4-byte global load the compiler does not see (so that it never puts its own, uncounted, vmcnt wait in front of the use): the
// caller guarantees by the ring's counted waits that the value has arrived before it is read
__device__ __forceinline__ float ring_gload(const float *ptr)
{
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

// ET: 0 exact fp32 (v_mfma_f32_32x32x2_f32), 1 bf16 (v_mfma_f32_32x32x16_bf16); NJ: accumulator columns per wave (tile = 256 x 128 NJ);
// GATE (NJ == 2): tile = 256 x 128 hidden columns; OBF: bf16 output (ET == 1, NJ == 2, plain)
// GF (bf16 gated instance): 0 fp32 output, 1 bf16 hidden only (sampling), 3 bf16 hidden + both pre-activations (training forward)
template <int ET, int NJ, bool GATE, bool OBF, bool ADD, int GF = 0>
__global__ __launch_bounds__(512, 1) void gemm_ring_kernel(const GemmP p, int ntm, int ntn, int total_tiles)
{
    static_assert(!GATE || NJ == 2, "gate: two accumulator columns (a, b)");
    static_assert(!OBF || (ET == 1 && NJ == 2 && !GATE), "bf16 output: plain bf16 instance");
    static_assert(!ADD || (!OBF && !GATE), "in-place addend: plain fp32-output instances");
    constexpr int BN = GATE ? 128 : 128 * NJ;                      // output columns per tile
    constexpr int RPW = 2 + NJ;                                    // LDS-DMA instructions per wave and step (2 x A, NJ x W)
    constexpr int NM = (ET ? 1 : 4) * 4 * NJ;                      // MFMAs of one slice
    // GF == 4 (fp32, plain NJ == 2 instance): the ReGLU BACKWARD in the epilogue -- the GEMM result is dh; da = dh relu(b), db = dh a (b > 0)
    // with a = p.in2, b = p.in3 (fp32 [M, ldo] like the two outputs p.out, p.out2); dh never reaches HBM
    constexpr bool GBW = GF == 4;
    static_assert(!GF || GBW || (GATE && !OBF && !ADD), "gate forward with saved pre-activations: a gated instance");
    static_assert(!GF || ET == 1 || GF == 3 || GBW, "fp32: hidden + both pre-activations (the training forward) or the plain gated instance");
    static_assert(!GBW || (ET == 0 && NJ == 2 && !GATE && !OBF && !ADD), "gate backward: the plain fp32 256 x 256 instance");
    constexpr int NSTORE = GBW ? 32 * NJ : GF ? 16 * GF : (OBF || GATE) ? 16 : 16 * NJ;  // row stores per wave and tile in the epilogue
    constexpr int NBIAS = GATE ? 2 : 4 * NJ;                       // bias loads per lane and tile (issued at the tile's start)
    extern __shared__ __attribute__((aligned(16))) char rlds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int nk = p.K >> 4;                                       // steps per tile: p.K counts 4-byte units (fp32 / two bf16), 16 per step
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
        n0 = tn_ * BN;
    };

    // ---- loader: this wave moves rows [32 wave, + 32) of the A tile and rows [16 NJ wave, + 16 NJ) of the W stage, 16 rows per
    // instruction.  GATE: W-stage rows 0-127 come from the "a" weights (waves 0-3), rows 128-255 from the "b" weights (waves 4-7).
    int l_tile = 0, l_kt = 0, l_step = 0, l_seg = 0, l_kin = 0, l_nloc0 = 0;
    const char *a_cur = nullptr, *w_cur = nullptr;
    int a_off[2], w_off[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + 16 * i + (lane >> 2);
        a_off[i] = row * (int)lda_b + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int row = wave * 16 * NJ + 16 * i + (lane >> 2);                       // row of the W stage
        const int wrow = GATE ? (row & 127) : row;                                   // row of the weight matrix inside the tile
        w_off[i] = wrow * (int)ldw_b + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
    }
    auto weight_rows = [&]() {
        const float *base = (GATE && wave >= 4) ? p.w2[l_seg] : p.w[l_seg];
        w_cur = (const char *)base + (long long)l_nloc0 * ldw_b;
    };
    auto loader_setup = [&]() {
        int m0, n0;
        tile_coords(l_tile, m0, n0);
        l_seg = (p.seg_mode == LDM_SEG_N) ? n0 / p.seg_len : 0;
        l_nloc0 = (p.seg_mode == LDM_SEG_N) ? n0 - l_seg * p.seg_len : n0;
        l_kin = 0;
        a_cur = (const char *)p.a + (long long)m0 * lda_b;
        weight_rows();
    };
    unsigned ld_dst = 0;
    const char *ld_a = nullptr, *ld_w = nullptr;
    auto issue_begin = [&](bool live) {
        ld_dst = lds_base + (unsigned)((l_step & (RNS - 1)) * RSTAGE);
        ld_a = live ? a_cur : (const char *)p.a;
        ld_w = live ? w_cur : (const char *)((GATE && wave >= 4) ? p.w2[0] : p.w[0]);
    };
    // part 0 .. RPW - 1: A rows 0-15, W rows 0-15, A rows 16-31 (, W rows 16-31) of this wave
    auto issue_part = [&](int part) {
        const char *src = (part & 1) ? ld_w : ld_a;
        const int off = part == 0 ? a_off[0] : part == 2 ? a_off[1] : part == 1 ? w_off[0] : w_off[NJ - 1];
        const unsigned dst = part == 0 ? ld_dst + wave * 2048 : part == 2 ? ld_dst + wave * 2048 + 1024
                           : part == 1 ? ld_dst + 16384 + wave * 1024 * NJ : ld_dst + 16384 + wave * 1024 * NJ + 1024;
        if (part == 3 && NJ != 2) return;
        if constexpr (ET == 0) ring_glds16(src, off, __builtin_amdgcn_readfirstlane(dst));
        else ring_glds16v(src + off, __builtin_amdgcn_readfirstlane(dst));
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

    // ---- consumer ---------------------------------------------------------------------------------------------------------------
    f32x16 acc[4][NJ];
    f32x4 fa0[4], fb0[NJ], fa1[4], fb1[NJ];
    const int sw = (r >> 2) & 3;
    const int lo0 = r * 64 + ((h ^ sw) << 4), lo1 = r * 64 + (((2 + h) ^ sw) << 4);      // this lane's chunk of slice 0 / 1 inside a 32-row block
    // one fragment at a time, in the order the MFMAs consume them: A0 W0 (W1) A1 A2 A3
    auto read_one = [&](const char *st, int slice, int idx, f32x4 (&fa)[4], f32x4 (&fb)[NJ]) {
        const int lo = slice ? lo1 : lo0;
        if (idx == 0) {
            fa[0] = *(const f32x4 *)(st + (wm * 128) * 64 + lo);
        } else if (idx <= NJ) {
            const int j = idx - 1;
            const int wrow = GATE ? j * 128 + wn * 32 : wn * 32 * NJ + j * 32;
            fb[j] = *(const f32x4 *)(st + 16384 + wrow * 64 + lo);
        } else {
            const int i = idx - NJ;
            fa[i] = *(const f32x4 *)(st + (wm * 128 + i * 32) * 64 + lo);
        }
    };
    auto mma_one = [&](const f32x4 (&fa)[4], const f32x4 (&fb)[NJ], int q) {
        if constexpr (ET == 1) {
            const int i = q / NJ, j = q % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        } else {
            const int e = q / (4 * NJ), i = (q % (4 * NJ)) / NJ, j = q % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    };
    auto clear_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };

    // ---- prologue: the whole ring in flight, then step 0's first slice into registers ---------------------------------------------
#ifdef LDM_RING_PRIO                                     // probe: static priority for the younger wave group (MI355X_MICROARCH.md, two waves per SIMD)
    if (wave >= 4) __builtin_amdgcn_s_setprio(LDM_RING_PRIO);
#endif
    loader_setup();
#pragma unroll 1
    for (int s0 = 0; s0 < RNS; ++s0) {
        const bool live = s0 < total_steps;
        issue_begin(live);
#pragma unroll
        for (int part = 0; part < RPW; ++part) issue_part(part);
        issue_advance(live);
    }
    ring_wait<3 * RPW>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int idx = 0; idx < 4 + NJ; ++idx) read_one(rlds, 0, idx, fa0, fb0);

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
#ifdef LDM_RING_STAMP      // probe build (tools): cycle stamps of workgroup 0 / wave 0 into the buffer passed as bias2[0] of a plain problem
        long long *stamps = (!GATE && blockIdx.x == 0 && wave == 0 && lane == 0) ? (long long *)p.bias2[0] : nullptr;
        if (stamps && c_tile < 16) stamps[c_tile * 4 + 0] = (long long)__builtin_amdgcn_s_memtime();
        if (stamps && (c_tile == 0 || c_tile == my_tiles - 1)) {
            stamps[200 + (c_tile ? 2 : 0)] = (long long)__builtin_amdgcn_s_memtime();
            stamps[201 + (c_tile ? 2 : 0)] = (long long)__builtin_amdgcn_s_memrealtime();
        }
#endif
        // bias values of this tile's columns: loaded HERE, behind the compiler's back -- a load it knows about gets an uncounted
        // vmcnt(0) in front of its first use, i.e. a wait for every LDS-DMA in flight at the top of the epilogue.  The fourth counted
        // wait of the K loop retires them (they are older than all but the last two steps' DMA).  Absent biases read A's first row.
        const int seg_n = (p.seg_mode == LDM_SEG_N) ? c_n0 / p.seg_len : 0;
        float braw[NJ][GATE ? 1 : LDM_MAX_SEG];
        bool bhas[NJ][GATE ? 1 : LDM_MAX_SEG];
        if constexpr (GATE) {
            const int bidx = c_n0 + wn * 32 + r - seg_n * p.seg_len;
            bhas[0][0] = p.bias[seg_n] != nullptr;
            bhas[1][0] = p.bias2[seg_n] != nullptr;
            braw[0][0] = ring_gload(bhas[0][0] ? p.bias[seg_n] + bidx : p.a);
            braw[1][0] = ring_gload(bhas[1][0] ? p.bias2[seg_n] + bidx : p.a);
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = c_n0 + wn * 32 * NJ + j * 32 + r;
                const int bidx = (p.seg_mode == LDM_SEG_N) ? col - seg_n * p.seg_len : col;
                const bool ksum = p.seg_mode == LDM_SEG_K;
#pragma unroll
                for (int sg = 0; sg < LDM_MAX_SEG; ++sg) {
                    const float *base = ksum ? p.bias[sg] : (sg == 0 ? p.bias[seg_n] : nullptr);
                    bhas[j][sg] = base != nullptr;
                    braw[j][sg] = ring_gload(base ? base + bidx : p.a);
                }
            }
        }
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++s) {
            const char *st_cur = rlds + (s & (RNS - 1)) * RSTAGE, *st_nxt = rlds + ((s + 1) & (RNS - 1)) * RSTAGE;
            // slice 0: one fragment read of slice 1 behind each of the first MFMAs (all eight waves run in step: the reads in a row
            // from every wave at once are 48 KiB of LDS traffic in front of the second MFMA)
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                mma_one(fa0, fb0, q);
                if (q < 4 + NJ) read_one(st_cur, 1, q, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // step s + 1 has landed once only the two younger steps' instructions are outstanding; this wave's reads of stage s are
            // complete: after the barrier stage s may be refilled and stage s + 1 read.  In the first three steps after an epilogue
            // the previous tile's NSTORE row stores are also younger than the DMA waited for (they were issued after the DMA of the
            // tile's first three steps): counting them in keeps the wave from waiting out the whole chip's synchronised store burst
            // (all workgroups finish their equally long tiles together: 32 k cycles per tile before this).
            // The tile's NBIAS bias loads are younger than that DMA in the same three steps.
            if (kt < 3 && nk >= 3) {
                // (vmcnt is a 6-bit counter: a larger count is clamped, which only makes the wait stricter)
                if (c_tile > 0) ring_wait<(2 * RPW + NSTORE + NBIAS < 63 ? 2 * RPW + NSTORE + NBIAS : 63)>();
                else ring_wait<2 * RPW + NBIAS>();
            } else {
                ring_wait<2 * RPW>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // slice 1: the DMA of step s + 4 (into stage s) and the next step's slice-0 fragments, one instruction per MFMA gap
            const bool live = s + RNS < total_steps;
            issue_begin(live);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                mma_one(fa1, fb1, q);
                if (q < 4 + NJ) read_one(st_nxt, 0, q, fa0, fb0);      // past the last step: stale LDS, never used
                if ((q & 1) && (q >> 1) < RPW) issue_part(q >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_advance(live);
#ifdef LDM_RING_STAMP
            if (stamps && c_tile == 1 && kt < 32) stamps[64 + kt] = (long long)__builtin_amdgcn_s_memtime();
#endif
        }
#ifdef LDM_RING_STAMP
        if (stamps && c_tile < 16) stamps[c_tile * 4 + 1] = (long long)__builtin_amdgcn_s_memtime();
#endif

        // ---- epilogue of this tile ----------------------------------------------------------------------------------------------
        if (nk < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // short tiles: the K loop's waits did not cover the bias loads
        float b1[NJ], b2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int sg = 0; sg < (GATE ? 1 : LDM_MAX_SEG); ++sg) {
                asm volatile("" : "+v"(braw[j][sg]));                            // keeps every use of the values below the K loop
                braw[j][sg] = bhas[j][sg] ? braw[j][sg] : 0.f;
            }
        if constexpr (GATE) {
            b1[0] = braw[0][0];
            b1[1] = 0.f;
            b2 = braw[1][0];
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) b1[j] = ((braw[j][0] + braw[j][1]) + braw[j][2]) + braw[j][3];
        }
        const long long orow0 = (long long)(c_m0 + wm * 128);
#if defined(LDM_RING_PROBE) && LDM_RING_PROBE == 1       // probe build: no epilogue at all (results are not stored)
        {
            float dbg = 0.f;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < NJ; ++j) dbg += acc[i][j][(i + j) & 15];
            if (dbg == 12345.678f) ((float *)p.out)[c_m0] = dbg + b1[0] + b2;
            continue;
        }
#endif
        // the activation is wave-uniform: chosen once per tile, not per element
        auto run = [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
            const float slope = p.slope;
            auto value = [&](int i, int j, int e) {
                float v = acc[i][j][e] + b1[j];
                if constexpr (GATE) v = v * fmaxf(acc[i][1][e] + b2, 0.f);
                else if constexpr (ACT == LDM_ACT_RELU) v = fmaxf(v, 0.f);
                else if constexpr (ACT == LDM_ACT_LRELU) v = v > 0.f ? v : v * slope;
                return v;
            };
            if constexpr (GF == 3 && ET == 0) {
                // ReGLU forward of the fp32 training step: hid = (a + ba) relu(b + bb) AND the two pre-activations, all fp32 [M, ldo]:
                // three passes of each 32 x 32 piece through the scratch, 16 bytes per lane and store (unet.py:14-15 + what its
                // autograd keeps); replaces two plain GEMMs and an elementwise pass
                const int ldo_ = (int)p.ldo;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pass = 0; pass < 3; ++pass) {
                        float *o32 = (float *)(pass == 0 ? p.out : (pass == 1 ? p.out2 : p.out3));
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float av = acc[i][0][e] + b1[0], gt = acc[i][1][e] + b2;
                            const float v = pass == 0 ? av * fmaxf(gt, 0.f) : (pass == 1 ? av : gt);
                            const int slot = h | (e & 2) | ((e & 1) << 2) | ((e >> 2) << 3);
                            *(float *)(scr + slot * 128 + r * 4) = v;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 v = *(const f32x4 *)(scr + (8 * k + rsub) * 128 + cc * 16);
                            *(f32x4 *)(o32 + (orow0 + i * 32 + 8 * k + rrow) * ldo_ + c_n0 + wn * 32 + cc * 4) = v;
                        }
                    }
            } else if constexpr (GF && !GBW) {
                // ReGLU forward of the bf16 training step: hid = (a + ba) relu(b + bb) AND the two pre-activations, all bf16 [M, ldo];
                // three passes of each 32 x 32 fp32 piece through the scratch, 4 columns (8 bytes) per lane
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const int ldo_ = (int)p.ldo;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pass = 0; pass < GF; ++pass) {
                        unsigned short *o16 = (unsigned short *)(pass == 0 ? p.out : (pass == 1 ? p.out2 : p.out3));
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float av = acc[i][0][e] + b1[0], gt = acc[i][1][e] + b2;
                            const float v = pass == 0 ? av * fmaxf(gt, 0.f) : (pass == 1 ? av : gt);
                            const int slot = h | (e & 2) | ((e & 1) << 2) | ((e >> 2) << 3);
                            *(float *)(scr + slot * 128 + r * 4) = v;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 v = *(const f32x4 *)(scr + (8 * k + rsub) * 128 + cc * 16);
                            *(u32x2 *)(o16 + (orow0 + i * 32 + 8 * k + rrow) * ldo_ + c_n0 + wn * 32 + cc * 4) =
                                u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        }
                    }
            } else if constexpr (OBF) {
                unsigned short *o16 = (unsigned short *)p.out;
                const int ldo_ = (int)p.ldo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // 32 x 64 strip as bf16 into the scratch: row (e & 3) + 8 (e >> 2) + 4 h sits in slot h | (e & 2) | (e & 1) << 2 | 8 (e >> 2)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const float v = value(i, j, e);
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
                constexpr int NJO = GATE ? 1 : NJ, NFR = 4 * NJO;
                auto col_of = [&](int f) { return (long long)(c_n0 + (GATE ? wn * 32 : wn * 32 * NJ + (f % NJO) * 32) + cc * 4); };
                // The addend of fragment f + 1 is loaded BEFORE fragment f is stored.  The compiler's vmcnt for a load counts only the
                // vector-memory operations it issued itself after it: with the loads of f issued after the stores of f - 1 (or with a
                // run-time "is there an addend" flag, which made it wait in every instance) each fragment waited for the previous
                // fragment's stores to be acknowledged -- 4.7 us per tile, all of the epilogue's measured cost.
                // (half fragments = 16 rows at a time: two 16-register addend buffers did not fit beside 128 accumulators)
                // 256 x 128 tiles (64 accumulator registers) have room for the WHOLE tile's addend: all 16 loads go out at the top of the
                // epilogue.  In-kernel stamps at M = 262144, N = 128, K = 384 (K loop 107 k cycles = 4.4 k per step, 92 % of them MFMA
                // issue, at a 2.05 GHz shader clock): epilogue 10.5 k cycles with the one-ahead prefetch, 8.5 k with this, 4 k without
                // an addend -- what is left is the bandwidth of 256 workgroups reading and writing their 128 KiB at the same moment,
                // not latency (the launch: 239 -> 237 us)
                constexpr bool ADD_ALL = ADD && NJ == 1;
                constexpr int NAD = ADD_ALL ? 2 * NFR : 2;
                f32x4 ad[NAD][2];
                auto load_add = [&](int hh) {                 // half fragment hh = 2 f + kh: row pieces 2 kh, 2 kh + 1 of fragment f
                    const int f = hh >> 1, kh = hh & 1;
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        ad[ADD_ALL ? hh : (hh & 1)][k] = *(const f32x4 *)(p.addend + (orow0 + (f / NJO) * 32 + 8 * (2 * kh + k) + rrow) * lda_ + col_of(f));
                };
                if constexpr (ADD_ALL) {
#pragma unroll
                    for (int hh = 0; hh < 2 * NFR; ++hh) load_add(hh);
                } else if constexpr (ADD) {
                    load_add(0);
                }
                // gate backward: the pre-activations of half fragment hh + 1 are loaded before half fragment hh is stored (as the addend)
                // (one 8-row piece ahead: half a fragment ahead -- 32 registers of a and b -- spilled beside the 128 accumulators)
                f32x4 ga[2], gb[2];
                auto load_gate = [&](int qk) {                // piece qk = 4 f + 2 kh + k
                    const int f = qk >> 2;
                    const long long at = (orow0 + (f / NJO) * 32 + 8 * (qk & 3) + rrow) * ldo_ + col_of(f);
                    ga[qk & 1] = *(const f32x4 *)((const float *)p.in2 + at);
                    gb[qk & 1] = *(const f32x4 *)((const float *)p.in3 + at);
                };
                if constexpr (GBW) load_gate(0);
#pragma unroll
                for (int f = 0; f < NFR; ++f) {
                    const int i = f / NJO, j = f % NJO;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int slot = h | (e & 2) | ((e & 1) << 2) | ((e >> 2) << 3);
                        *(float *)(scr + slot * 128 + r * 4) = value(i, j, e);
                    }
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
                        const int hh = 2 * f + kh;
                        f32x4 v[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) v[k] = *(const f32x4 *)(scr + (8 * (2 * kh + k) + rsub) * 128 + cc * 16);
                        if constexpr (ADD) {
                            if constexpr (!ADD_ALL) {
                                if (hh + 1 < 2 * NFR) load_add(hh + 1);
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k)
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[k][q] += ad[ADD_ALL ? hh : (hh & 1)][k][q];
                        }
                        if constexpr (GBW) {
                            float *o2 = (float *)p.out2;
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const int qk = 2 * hh + k;
                                if (qk + 1 < 4 * NFR) load_gate(qk + 1);
                                f32x4 oa, ob;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float g = v[k][q], av = ga[qk & 1][q], bv = gb[qk & 1][q];
                                    oa[q] = g * fmaxf(bv, 0.f);                      // exactly ldm_gate_bwd_f32's arithmetic
                                    ob[q] = bv > 0.f ? g * av : 0.f;
                                }
                                const long long at = (orow0 + i * 32 + 8 * (2 * kh + k) + rrow) * ldo_ + col_of(f);
                                *(f32x4 *)(o32 + at) = oa;
                                *(f32x4 *)(o2 + at) = ob;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < 2; ++k) *(f32x4 *)(o32 + (orow0 + i * 32 + 8 * (2 * kh + k) + rrow) * ldo_ + col_of(f)) = v[k];
                        }
                    }
                }
            }
        };
        if constexpr (GATE) {
            run(std::integral_constant<int, LDM_ACT_GATE>{});
        } else {
            if (p.act == LDM_ACT_RELU) run(std::integral_constant<int, LDM_ACT_RELU>{});
            else if (p.act == LDM_ACT_LRELU) run(std::integral_constant<int, LDM_ACT_LRELU>{});
            else run(std::integral_constant<int, LDM_ACT_NONE>{});
        }
#ifdef LDM_RING_STAMP
        if (stamps && c_tile < 16) stamps[c_tile * 4 + 2] = (long long)__builtin_amdgcn_s_memtime();
        if (!GATE && wave == 0 && lane == 0 && p.bias2[0]) {                     // every workgroup: realtime at its first tile's start and at its last tile's end
            long long *all = (long long *)p.bias2[0];
            if (c_tile == 0) all[256 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
            if (c_tile == my_tiles - 1) all[257 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
        }
#endif
    }
    // the trailing (dummy) DMA must have landed before the workgroup's LDS is handed to the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int ET, int NJ, bool GATE, bool OBF, bool ADD = false, int GF = 0>
int ring_launch(const GemmP &p, hipStream_t st)
{
    static LdmLdsOptIn opt_in;                               // per device: a device that refuses 160 KiB of LDS per workgroup falls back to the stream kernel
    auto kern = gemm_ring_kernel<ET, NJ, GATE, OBF, ADD, GF>;
    if (!opt_in((const void *)kern, RSMEM)) return 0;
    const int cus = ldm_cu_count();
    const int ntm = p.M / RT, ntn = p.N / (GATE ? 128 : 128 * NJ);
    const int total = ntm * ntn;
    int grid = total < cus ? total : cus;
    if (g_ring == 3 && grid > 8) grid = 8;
    if (grid > 8) grid &= ~7;
    ldm_launch(kern, dim3(grid), dim3(512), RSMEM, st, p, ntm, ntn, total);
    return 1;
}

// shape rules shared by both operand types; returns the accumulator columns per wave (2: 256-column tiles, 1: 128-column tiles)
// the problem should run with, or 0 if the ring kernel does not take it
int ring_shape(const GemmP &p, int groups, bool gate, bool out_bf16, bool fp32, bool gate_fwd = false)
{
    if (g_ring == 0 || groups != 1 || p.use_table || p.in2 || p.addend16) return 0;
    if (gate_fwd ? ((p.out2 != nullptr) != (p.out3 != nullptr)) : (p.out2 != nullptr)) return 0;
    if (p.M % RT || (p.K & 15) || p.K < 16) return 0;
    if (p.seg_mode == LDM_SEG_K && p.nseg > 1 && (p.seg_len & 15)) return 0;
    if (p.lda * 4 * RT >= (1ll << 31) || p.ldw * 4 * RT >= (1ll << 31)) return 0;
    if (!ldm_aligned16(p.out) || p.ldo % (out_bf16 ? 8 : 4)) return 0;
    if (p.addend && (out_bf16 || !ldm_aligned16(p.addend) || p.ldadd % 4)) return 0;
    const int unit = (p.seg_mode == LDM_SEG_N && p.nseg > 1) ? p.seg_len : p.N;       // a tile's columns must lie inside one N-segment
    const long long mt = p.M / RT;
    // auto: the tiles fill whole rounds of the CUs (one workgroup each: 384 tiles on 256 CUs ran 18 % BEHIND the stream kernel, 768
    // ahead of it), a tile is at least 8 steps long, and -- fp32 only -- an in-place addend comes with K >= 384 (its loads are not
    // prefetched under the last K-step as in the stream kernel: out-projections at K = 128 / 256 lost 6 %)
    const bool any = g_ring >= 2;
    const int cus = ldm_cu_count();
    auto fills = [&](long long tiles) {
        if (any) return true;
        const long long rounds = (tiles + cus - 1) / cus;
        return tiles * 4 >= (long long)cus * 3 && tiles * 100 >= rounds * cus * 88;
    };
    if (!any && p.K < (fp32 ? 128 : 64)) return 0;            // bf16: 4 steps (measured: the gated forward at C = 128 runs 1.5x the stream kernel)
    if (!any && fp32 && p.addend && p.K < 384) return 0;
    if (gate) return (unit % 128 == 0 && fills(mt * (p.N / 128))) ? 2 : 0;
    if (unit % 256 == 0 && fills(mt * (p.N / 256))) return 2;
    if (fp32 && unit % 128 == 0 && (any || !p.addend || p.K >= 384) && fills(mt * (p.N / 128))) return 1;
    // (bf16 has no 256 x 128 instance: a bf16 slice of such a tile is 4 MFMAs per wave, fewer than the 5 fragment reads the schedule pins
    // behind them -- the K-segment GEMM of the 8 x 8 level at batch 256, 128 tiles of 256 x 256, stays on the stream kernel)
    return 0;
}

}  // namespace

extern "C" int ldm_gemm_ring(int v)
{
    const int old = g_ring;
    if (v >= 0 && v <= 3) g_ring = v;
    return old;
}

// bf16 operands (ldm_gemm_bf16, plain problems): returns 1 if the ring kernel launched
int ldm_gemm_ring_dispatch_bf16(const GemmP &p, int groups, bool out_bf16, hipStream_t st)
{
    if (p.act != LDM_ACT_NONE && p.act != LDM_ACT_RELU && p.act != LDM_ACT_LRELU) return 0;
    if (ring_shape(p, groups, false, out_bf16, false) != 2) return 0;
    if (out_bf16) return ring_launch<1, 2, false, true>(p, st);
    return p.addend ? ring_launch<1, 2, false, false, true>(p, st) : ring_launch<1, 2, false, false, false>(p, st);
}

// bf16 ReGLU forward with saved pre-activations (ldm_gemm_bf16_gate_fwd): returns 1 if the ring kernel launched
int ldm_gemm_ring_dispatch_bf16_gate(const GemmP &p, int groups, hipStream_t st)
{
    if (p.addend || p.ldo % 4 || (((size_t)p.out | (size_t)p.out2 | (size_t)p.out3) & 7)) return 0;
    if (ring_shape(p, groups, true, false, false, true) != 2) return 0;
    if (!p.out2) return ring_launch<1, 2, true, false, false, 1>(p, st);          // hidden only (sampling: nothing is kept for a backward)
    return ring_launch<1, 2, true, false, false, 3>(p, st);
}

// exact fp32 (ldm_gemm_f32: rows in, rows out, plain or gated): returns 1 if the ring kernel launched
int ldm_gemm_ring_dispatch_f32(const GemmP &p, int groups, bool gate, int amode, hipStream_t st)
{
    if (amode != LDM_A_ROWS || p.o_mode != LDM_O_ROWS) return 0;
    if (!gate && p.act != LDM_ACT_NONE && p.act != LDM_ACT_RELU && p.act != LDM_ACT_LRELU) return 0;
    if (!gate && p.in2 && p.in3 && p.out2) {                         // ldm_gemm_f32_gate_bwd: da, db out of dh = dy . Wc in the epilogue
        GemmP q = p;
        q.out2 = nullptr;
        q.in2 = q.in3 = nullptr;
        if (p.addend || p.act != LDM_ACT_NONE || ring_shape(q, groups, false, false, true) != 2) return 0;
        return ring_launch<0, 2, false, false, false, 4>(p, st);
    }
    const bool keep_pre = gate && p.out2 != nullptr;                 // ldm_gemm_f32_gate_fwd: hidden + both pre-activations
    const int nj = ring_shape(p, groups, gate, false, true, keep_pre);
    if (nj == 0) return 0;
    if (keep_pre) return p.addend ? 0 : ring_launch<0, 2, true, false, false, 3>(p, st);
    if (gate) return p.addend ? 0 : ring_launch<0, 2, true, false>(p, st);
    if (p.addend) return nj == 2 ? ring_launch<0, 2, false, false, true>(p, st) : ring_launch<0, 1, false, false, true>(p, st);
    return nj == 2 ? ring_launch<0, 2, false, false, false>(p, st) : ring_launch<0, 1, false, false, false>(p, st);
}
