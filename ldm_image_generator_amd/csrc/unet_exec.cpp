// Native executor of one UNet forward (unet.py:89-103) -- the denoise loop's per-step body as ONE C call.
//
// The host language of the reference is Python and so is the drop-in layer, but at small batch the
// ~190 kernel launches of a forward are launch-bound when each goes through Python + ctypes (~23 us
// per launch measured, 4.4 ms per denoise step at batch 1).  This file replays exactly the launch
// sequence of ldm_image_generator_amd/unet.py::UNet.forward from C++ over a caller-owned workspace:
// same kernels, same order, bit-identical results (tests/test_gpu_modules.py::test_native_forward_*).
// The Python side keeps what is inherently Python: the draws from Python's global `random`
// (stochastic depth, expert choice), handed over as one int per block.
#include "common.h"
#include <cstring>
#include <cstdlib>

extern "C" {
int ldm_gemm_f32(const ldm_gemm_desc *d, void *stream);
}
// (every other entry point the executor calls is declared in include/ldm_hip.h, which common.h includes)

namespace {

struct Bump {
    char *base;
    size_t size, off;
    bool ok;
    float *take(size_t floats)
    {
        const size_t bytes = (floats * sizeof(float) + 255) & ~(size_t)255;
        if (off + bytes > size) {
            ok = false;
            return nullptr;
        }
        float *p = (float *)(base + off);
        off += bytes;
        return p;
    }
};

size_t pad256(size_t floats) { return (floats * sizeof(float) + 255) & ~(size_t)255; }

struct Level {
    int C, H, W, nblk;          // nblk = encoder + decoder blocks of this level
    long long M;                // B * H * W
    long long Mf;               // nT * H * W (FiLM rows)
};

// everything the executor allocates from the workspace, in one place so that sizing and use cannot diverge
struct Layout {
    float *codes[LDM_MAX_LEVELS], *ench[LDM_MAX_LEVELS], *film[LDM_MAX_LEVELS];
    float *act[LDM_MAX_LEVELS][3];
    float *xf, *hidden, *qkv, *ctx, *pooled;
    float *scratch;             // split-K scratch for the small-M GEMMs (same size / rule as ops.py: 32 MiB, M <= 128, K >= 256)
};

constexpr size_t kScratchBytes = 32u << 20;

bool carve(const ldm_unet_plan *pl, const Level *lv, Bump &b, Layout &L)
{
    size_t mc = 0, m3c = 0, mpool = 0;
    for (int i = 0; i < pl->levels; ++i) {
        L.codes[i] = b.take((size_t)lv[i].Mf * 2 * lv[i].C);
        L.ench[i] = b.take((size_t)lv[i].nblk * lv[i].Mf * 4 * lv[i].C);
        L.film[i] = b.take((size_t)lv[i].nblk * lv[i].Mf * 2 * lv[i].C);
        for (int k = 0; k < 3; ++k) L.act[i][k] = b.take((size_t)lv[i].M * lv[i].C);
        const size_t a = (size_t)lv[i].M * lv[i].C;
        mc = a > mc ? a : mc;
        m3c = 3 * a > m3c ? 3 * a : m3c;
        if (i + 1 < pl->levels) {
            const size_t q = (size_t)lv[i + 1].M * lv[i].C;
            mpool = q > mpool ? q : mpool;
        }
    }
    L.xf = b.take(mc);
    L.hidden = b.take(m3c);
    L.qkv = b.take(m3c);
    L.ctx = b.take(mc);
    L.pooled = b.take(mpool ? mpool : 64);
    L.scratch = b.take(kScratchBytes / sizeof(float));
    return b.ok;
}

bool levels_of(const ldm_unet_plan *pl, int B, int H, int W, int nT, Level *lv)
{
    if (pl->levels < 1 || pl->levels > LDM_MAX_LEVELS) return false;
    for (int i = 0; i < pl->levels; ++i) {
        lv[i].C = pl->channels[i];
        lv[i].H = H >> i;
        lv[i].W = W >> i;
        if (lv[i].H < 1 || lv[i].W < 1 || (i + 1 < pl->levels && ((lv[i].H | lv[i].W) & 1))) return false;
        lv[i].M = (long long)B * lv[i].H * lv[i].W;
        lv[i].Mf = (long long)nT * lv[i].H * lv[i].W;
        lv[i].nblk = pl->enc_blocks[i] + pl->dec_blocks[i];
    }
    return true;
}

// chunking knob of run_stack (bytes of one fp32 activation per chunk; 0 = whole batch).  LDM_UNET_CHUNK_MB overrides (A/B runs).
long long g_chunk_mb[2] = {0, 0};            // [fp32 mode, bf16 mode]
long long unet_chunk_bytes(bool bf16)
{
    static int env = -2;
    if (env == -2) {
        const char *e = getenv("LDM_UNET_CHUNK_MB");
        env = e ? atoi(e) : -1;
    }
    const long long mb = env >= 0 ? env : g_chunk_mb[bf16 ? 1 : 0];
    return mb << 20;
}

// Second stream of the executor.  Inside a SwinBlock the gated GEMM of the MoE reads only the normalised input, like the grouped conv
// (and the attention chain): the two branches meet again at the K-segment GEMM.  With LDM_UNET_STREAMS=2 (or ldm_unet_streams(2)) the gated
// GEMM runs on a side stream between an event fork and join, so the tail of one branch's last tile round overlaps the head of the
// other's.  Same kernels on the same operands: results are bit-identical.  Off by default until measured.
int g_unet_streams = -1;             // -1: read LDM_UNET_STREAMS once; 1: one stream; 2: fork the gated GEMM
struct SideStream {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ok = false, tried = false;
};
SideStream g_side[64];
SideStream *side_stream()
{
    if (g_unet_streams < 0) {
        const char *e = getenv("LDM_UNET_STREAMS");
        g_unet_streams = e ? atoi(e) : 1;
    }
    if (g_unet_streams < 2) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    SideStream &s = g_side[dev & 63];
    if (!s.tried) {
        s.tried = true;
        s.ok = hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
        if (!s.ok) (void)hipGetLastError();
    }
    return s.ok ? &s : nullptr;
}

#define RUN(call)                    \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != LDM_OK) return rc_; \
    } while (0)

ldm_gemm_desc gemm_rows(const float *a, long long M, int N, int K, const float *w, const float *bias, float *out)
{
    ldm_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.a = a; d.lda = K; d.M = (int)M; d.N = N; d.K = K;
    d.nseg = 1; d.w[0] = w; d.bias[0] = bias; d.ldw = K;
    d.out = out; d.ldo = N; d.ldadd = N; d.groups = 1;
    return d;
}

// the small-M rule of ops.gemm: hand the GEMM a scratch so that it may split its reduction over the grid
void allow_splitk(ldm_gemm_desc &d, const Layout &L)
{
    if (d.M <= 128 && d.K >= 256 && d.groups == 1) {
        d.workspace = L.scratch;
        d.workspace_bytes = (long long)kScratchBytes;
    }
}

// bf16 GEMM problem: rows of bf16 in, one bf16 weight segment; everything else as gemm_rows
ldm_gemm_desc gemm_rows16(const void *a, long long M, int N, int K, const void *w, const float *bias, void *out)
{
    ldm_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.a = (const float *)a; d.lda = K; d.M = (int)M; d.N = N; d.K = K;
    d.nseg = 1; d.w[0] = (const float *)w; d.bias[0] = bias; d.ldw = K;
    d.out = (float *)out; d.ldo = N; d.ldadd = N; d.groups = 1;
    return d;
}

// one SwinBlock in the bf16 ("autocast") sampling mode: the residual stream x / y stays fp32; ChannelNorm + FiLM rounds its result
// ONCE to bf16 (the operand of the grouped conv, the in-projection and the gated GEMM); every GEMM accumulates in fp32; the gated
// hidden and the attention context are bf16 rows; the attention core (softmax, float "mask") computes in fp32 on the fp32 QKV.
int run_block_bf16(const ldm_unet_plan *pl, const ldm_unet_block *bk, const ldm_unet_block_bf16 *b16, int decision, const float *film, const int *slot,
                   const float *x, float *y, const Level &lv, int B, const Layout &L, void *st)
{
    const int C = lv.C;
    const long long M = lv.M;
    void *xf16 = L.xf, *hid16 = L.hidden, *ctx16 = L.ctx;
    RUN(ldm_channelnorm_film_bf16(x, film, slot, nullptr, xf16, B, lv.H * lv.W, C, pl->eps, st));
    const int e1 = decision >> 2, e2 = decision & 3;
    const int sel[3] = {0, 1 + e1, 1 + e2};
    auto gated = [&](void *stream) -> int {
        ldm_gemm_desc g = gemm_rows16(xf16, M, 3 * C, C, nullptr, nullptr, hid16);
        g.nseg = 3; g.seg_mode = LDM_SEG_N; g.seg_len = C; g.act = LDM_ACT_GATE;
        for (int s = 0; s < 3; ++s) {
            g.w[s] = (const float *)b16->a_w[sel[s]]; g.bias[s] = bk->a_b[sel[s]];
            g.w2[s] = (const float *)b16->b_w[sel[s]]; g.bias2[s] = bk->b_b[sel[s]];
        }
        return ldm_gemm_bf16_gate_fwd(&g, nullptr, nullptr, stream);
    };
    SideStream *ss = side_stream();
    if (ss) {
        if (hipEventRecord(ss->fork, (hipStream_t)st) != hipSuccess || hipStreamWaitEvent(ss->side, ss->fork, 0) != hipSuccess) ss = nullptr;
    }
    if (ss) {
        RUN(gated(ss->side));
        if (hipEventRecord(ss->join, ss->side) != hipSuccess) { ldm_set_error("ldm_unet_forward: event record failed"); return LDM_ELAUNCH; }
    }
    RUN(ldm_gconv3x3_bf16(xf16, b16->conv_w, bk->conv_b, x, y, B, lv.H, lv.W, C, st));           // y = conv3x3_grouped(xf) + bias + x
    if (bk->attention) {
        ldm_gemm_desc q = gemm_rows16(xf16, M, 3 * C, C, b16->in_w, bk->in_b, L.qkv);           // q, k, v as bf16 rows (half the bytes each way)
        RUN(ldm_gemm_bf16(&q, 1, st));
        RUN(ldm_window_attention_bf16io(L.qkv, 1, bk->in_b, xf16, ctx16, B, lv.H, lv.W, C, pl->window, bk->shift, st));
        ldm_gemm_desc o = gemm_rows16(ctx16, M, C, C, b16->out_w, bk->out_b, y);
        o.addend = y; o.ldadd = C;
        RUN(ldm_gemm_bf16(&o, 0, st));
    }
    {
        if (ss) {
            if (hipStreamWaitEvent((hipStream_t)st, ss->join, 0) != hipSuccess) { ldm_set_error("ldm_unet_forward: stream wait failed"); return LDM_ELAUNCH; }
        } else {
            RUN(gated(st));
        }
        ldm_gemm_desc c = gemm_rows16(hid16, M, C, 3 * C, nullptr, nullptr, y);
        c.nseg = 3; c.seg_mode = LDM_SEG_K; c.seg_len = C; c.ldw = C;
        for (int s = 0; s < 3; ++s) { c.w[s] = (const float *)b16->c_w[sel[s]]; c.bias[s] = bk->c_b[sel[s]]; }
        c.addend = y; c.ldadd = C;
        RUN(ldm_gemm_bf16(&c, 0, st));
    }
    return LDM_OK;
}

// one SwinBlock (unet.py:42-47) on channels-last rows; y may not alias x
int run_block(const ldm_unet_plan *pl, const ldm_unet_block *bk, int decision, const float *film, const int *slot, const float *x,
              float *y, const Level &lv, int B, const Layout &L, void *st)
{
    const int C = lv.C;
    const long long M = lv.M;
    RUN(ldm_channelnorm_film_f32(x, film, slot, L.xf, B, lv.H * lv.W, C, pl->eps, st));
    const int e1 = decision >> 2, e2 = decision & 3;
    const int sel[3] = {0, 1 + e1, 1 + e2};                  // general + the two drawn experts (modules.py:35-36)
    auto gated = [&](void *stream) -> int {
        ldm_gemm_desc g = gemm_rows(L.xf, M, 3 * C, C, nullptr, nullptr, L.hidden);
        g.nseg = 3; g.seg_mode = LDM_SEG_N; g.seg_len = C; g.act = LDM_ACT_GATE;
        for (int s = 0; s < 3; ++s) {
            g.w[s] = bk->a_w[sel[s]]; g.bias[s] = bk->a_b[sel[s]];
            g.w2[s] = bk->b_w[sel[s]]; g.bias2[s] = bk->b_b[sel[s]];
        }
        allow_splitk(g, L);
        return ldm_gemm_f32(&g, stream);
    };
    SideStream *ss = M > 128 ? side_stream() : nullptr;      // small batches share one split-K scratch: one stream
    if (ss) {
        if (hipEventRecord(ss->fork, (hipStream_t)st) != hipSuccess || hipStreamWaitEvent(ss->side, ss->fork, 0) != hipSuccess) ss = nullptr;
    }
    if (ss) {
        RUN(gated(ss->side));
        if (hipEventRecord(ss->join, ss->side) != hipSuccess) { ldm_set_error("ldm_unet_forward: event record failed"); return LDM_ELAUNCH; }
    }
    {   // y = conv3x3_grouped(xf) + bias + x
        ldm_gemm_desc d = gemm_rows(L.xf, M, 32, 288, bk->conv_w, bk->conv_b, y);
        d.lda = C; d.ldw = 288; d.a_mode = LDM_A_CONV3X3; d.H = lv.H; d.W = lv.W; d.Cin = 32;
        d.addend = x; d.ldadd = C; d.ldo = C;
        d.groups = C / 32; d.a_gstride = 32; d.w_gstride = 32 * 288; d.o_gstride = 32; d.b_gstride = 32;
        RUN(ldm_gemm_f32(&d, st));
    }
    if (bk->attention) {
        ldm_gemm_desc q = gemm_rows(L.xf, M, 3 * C, C, bk->in_w, bk->in_b, L.qkv);
        allow_splitk(q, L);
        RUN(ldm_gemm_f32(&q, st));
        RUN(ldm_window_attention_f32(L.qkv, bk->in_b, L.xf, L.ctx, B, lv.H, lv.W, C, pl->window, bk->shift, st));
        ldm_gemm_desc o = gemm_rows(L.ctx, M, C, C, bk->out_w, bk->out_b, y);
        o.addend = y; o.ldadd = C;
        allow_splitk(o, L);
        RUN(ldm_gemm_f32(&o, st));
    }
    {
        if (ss) {
            if (hipStreamWaitEvent((hipStream_t)st, ss->join, 0) != hipSuccess) { ldm_set_error("ldm_unet_forward: stream wait failed"); return LDM_ELAUNCH; }
        } else {
            RUN(gated(st));
        }
        ldm_gemm_desc c = gemm_rows(L.hidden, M, C, 3 * C, nullptr, nullptr, y);
        c.nseg = 3; c.seg_mode = LDM_SEG_K; c.seg_len = C; c.ldw = C;
        for (int s = 0; s < 3; ++s) { c.w[s] = bk->c_w[sel[s]]; c.bias[s] = bk->c_b[sel[s]]; }
        c.addend = y; c.ldadd = C;
        allow_splitk(c, L);
        RUN(ldm_gemm_f32(&c, st));
    }
    return LDM_OK;
}

}  // namespace

extern "C" size_t ldm_unet_workspace_bytes(const ldm_unet_plan *pl, int B, int H, int W, int nT)
{
    Level lv[LDM_MAX_LEVELS];
    if (!pl || B < 1 || nT < 1 || !levels_of(pl, B, H, W, nT, lv)) return 0;
    Bump b{nullptr, ~(size_t)0 >> 1, 0, true};
    Layout L;
    carve(pl, lv, b, L);
    return b.off;
}

static int unet_forward_impl(const ldm_unet_plan *pl, const ldm_unet_plan_bf16 *p16, const float *x, const long long *t_unique, int nT, const int *slot,
                             const int *decisions, int B, int H, int W, void *workspace, size_t ws_bytes, float *out, int films_ready, void *st);

extern "C" int ldm_unet_forward_ex_f32(const ldm_unet_plan *pl, const float *x, const long long *t_unique, int nT, const int *slot,
                                       const int *decisions, int B, int H, int W, void *workspace, size_t ws_bytes, float *out, int films_ready,
                                       void *st)
{
    return unet_forward_impl(pl, nullptr, x, t_unique, nT, slot, decisions, B, H, W, workspace, ws_bytes, out, films_ready, st);
}

// The same forward in the bf16 ("autocast", ddpm.py:52,75) mode: `p16` holds bf16 copies of the GEMM weights (grouped conv packed like
// the fp32 plan's).  FiLM tables, stem, head and the ch_convs stay fp32; so does the residual stream.  Workspace: ldm_unet_workspace_bytes.
extern "C" int ldm_unet_forward_bf16(const ldm_unet_plan *pl, const ldm_unet_plan_bf16 *p16, const float *x, const long long *t_unique, int nT,
                                     const int *slot, const int *decisions, int B, int H, int W, void *workspace, size_t ws_bytes, float *out,
                                     int films_ready, void *st)
{
    LDM_REQUIRE(p16 && p16->blocks && pl && p16->nblocks == pl->nblocks, "ldm_unet_forward_bf16: bf16 plan missing or of another size");
    for (int i = 0; i < pl->levels; ++i) LDM_REQUIRE(pl->channels[i] % 64 == 0, "ldm_unet_forward_bf16: stage width %d is not a multiple of 64", pl->channels[i]);
    LDM_REQUIRE(pl->window * pl->window <= 48, "ldm_unet_forward_bf16: window %d too large for the MFMA attention kernel", pl->window);
    return unet_forward_impl(pl, p16, x, t_unique, nT, slot, decisions, B, H, W, workspace, ws_bytes, out, films_ready, st);
}

extern "C" int ldm_unet_forward_f32(const ldm_unet_plan *pl, const float *x, const long long *t_unique, int nT, const int *slot,
                                    const int *decisions, int B, int H, int W, void *workspace, size_t ws_bytes, float *out, void *st)
{
    return ldm_unet_forward_ex_f32(pl, x, t_unique, nT, slot, decisions, B, H, W, workspace, ws_bytes, out, 0, st);
}

// films_ready != 0: the FiLM tables of (t_unique, nT) are already in this workspace from an earlier call with the same plan, shapes and
// timesteps (the denoise loop computes them for ALL its timesteps in the first step -- they depend on t, never on x -- and every step
// then selects its rows through `slot`); the caller owns that guarantee.
static int unet_forward_impl(const ldm_unet_plan *pl, const ldm_unet_plan_bf16 *p16, const float *x, const long long *t_unique, int nT, const int *slot,
                             const int *decisions, int B, int H, int W, void *workspace, size_t ws_bytes, float *out, int films_ready, void *st)
{
    LDM_REQUIRE(pl && x && t_unique && decisions && workspace && out, "ldm_unet_forward_f32: null pointer");
    LDM_REQUIRE(B > 0 && nT > 0, "ldm_unet_forward_f32: bad batch");
    Level lv[LDM_MAX_LEVELS];
    LDM_REQUIRE(levels_of(pl, B, H, W, nT, lv), "ldm_unet_forward_f32: H=%d W=%d not divisible for %d levels", H, W, pl->levels);
    Bump bump{(char *)workspace, ws_bytes, 0, true};
    Layout L;
    LDM_REQUIRE(carve(pl, lv, bump, L), "ldm_unet_forward_f32: workspace too small (%zu bytes given)", ws_bytes);
    const int n = pl->levels;
    // block index ranges in execution order: encoder level 0..n-1, then decoder level n-1..0
    int enc0[LDM_MAX_LEVELS], dec0[LDM_MAX_LEVELS], total = 0;
    for (int i = 0; i < n; ++i) { enc0[i] = total; total += pl->enc_blocks[i]; }
    for (int i = n - 1; i >= 0; --i) { dec0[i] = total; total += pl->dec_blocks[i]; }
    LDM_REQUIRE(total == pl->nblocks, "ldm_unet_forward_f32: plan has %d blocks, stages sum to %d", pl->nblocks, total);

    // ---- FiLM tables of every block: sin/cos codes per level, then two grouped GEMMs per level (unet.py:18-21)
    for (int i = 0; i < n && !films_ready; ++i) {
        const int C = lv[i].C, G = lv[i].nblk;
        LDM_REQUIRE(G <= LDM_MAX_TABLE, "ldm_unet_forward_f32: %d blocks at level %d exceed the pointer-table size", G, i);
        RUN(ldm_sincos_embed_f32(t_unique, nT, lv[i].H, lv[i].W, C, pl->pos_freq[i], pl->time_freq[i], L.codes[i], st));
        const float *w1[LDM_MAX_TABLE], *b1[LDM_MAX_TABLE], *w2[LDM_MAX_TABLE], *b2[LDM_MAX_TABLE];
        for (int k = 0; k < G; ++k) {
            const int idx = k < pl->enc_blocks[i] ? enc0[i] + k : dec0[i] + (k - pl->enc_blocks[i]);
            const ldm_unet_block *bk = pl->blocks + idx;
            w1[k] = bk->enc_w1; b1[k] = bk->enc_b1; w2[k] = bk->enc_w2; b2[k] = bk->enc_b2;
        }
        ldm_gemm_desc d1 = gemm_rows(L.codes[i], lv[i].Mf, 4 * C, 2 * C, nullptr, nullptr, L.ench[i]);
        d1.w_table = w1; d1.bias_table = b1; d1.act = LDM_ACT_RELU; d1.groups = G; d1.o_gstride = lv[i].Mf * 4 * C;
        RUN(ldm_gemm_f32(&d1, st));
        ldm_gemm_desc d2 = gemm_rows(L.ench[i], lv[i].Mf, 2 * C, 4 * C, nullptr, nullptr, L.film[i]);
        d2.w_table = w2; d2.bias_table = b2; d2.groups = G; d2.a_gstride = lv[i].Mf * 4 * C; d2.o_gstride = lv[i].Mf * 2 * C;
        RUN(ldm_gemm_f32(&d2, st));
    }
    auto film_of = [&](int level, int k) { return L.film[level] + (size_t)k * lv[level].Mf * 2 * lv[level].C; };

    // One stack of SwinBlocks (blocks blk0 .. blk0 + nblk of the plan, FiLM tables film_k0 ...) at `level`, starting from activation buffer
    // `cur`; returns through `cur` the buffer that holds the result.  Samples never interact, so the stack may run over the
    // batch in CHUNKS of samples, chunk by chunk through ALL its blocks: with chunk_bytes > 0 a chunk is sized so that one fp32
    // activation of it takes at most that many bytes, which keeps a chunk's tensors (input, normalised copy, gated hidden, output --
    // and the block-to-block hand-over) inside the 256-MiB Infinity Cache instead of streaming them through HBM between launches.
    // Same kernels on the same rows: results do not depend on the chunking beyond the tile paths a different M selects.
    const long long chunk_bytes = unet_chunk_bytes(p16 != nullptr);
    auto run_stack = [&](int level, int blk0, int nblk, int film_k0, int &cur_buf) -> int {
        const Level &full = lv[level];
        const long long per_sample = (long long)full.H * full.W * full.C * (long long)sizeof(float);
        int bc = B;
        if (chunk_bytes > 0 && per_sample * B > chunk_bytes) {
            bc = (int)(chunk_bytes / per_sample);
            bc = bc < 1 ? 1 : bc;
            const int nchunks = (B + bc - 1) / bc;
            bc = (B + nchunks - 1) / nchunks;                                   // balanced chunks
        }
        int last = cur_buf;
        for (int b0 = 0; b0 < B; b0 += bc) {
            const int bn = B - b0 < bc ? B - b0 : bc;
            Level part = full;
            part.M = (long long)bn * full.H * full.W;
            const size_t off = (size_t)b0 * full.H * full.W * full.C;
            int c = cur_buf;
            for (int k = 0; k < nblk; ++k) {
                const int dcs = decisions[blk0 + k];
                if (dcs < 0) continue;                                          // stochastic depth (unet.py:39-40)
                const int nxt = (c + 1) % 3;
                const float *film = film_of(level, film_k0 + k);
                const int *sl = slot ? slot + b0 : nullptr;
                if (p16) RUN(run_block_bf16(pl, pl->blocks + blk0 + k, p16->blocks + blk0 + k, dcs, film, sl, L.act[level][c] + off, L.act[level][nxt] + off, part, bn, L, st));
                else RUN(run_block(pl, pl->blocks + blk0 + k, dcs, film, sl, L.act[level][c] + off, L.act[level][nxt] + off, part, bn, L, st));
                c = nxt;
            }
            last = c;
        }
        cur_buf = last;
        return LDM_OK;
    };

    // ---- stem -------------------------------------------------------------------------------------------
    int cur[LDM_MAX_LEVELS];                                   // which of the 3 activation buffers holds the live tensor
    for (int i = 0; i < n; ++i) cur[i] = 0;
    RUN(ldm_stem_nchw_f32(x, pl->stem_w, pl->stem_b, L.act[0][0], B, pl->input_channels, H * W, lv[0].C, st));
    int skip_buf[LDM_MAX_LEVELS];
    // ---- encoder ----------------------------------------------------------------------------------------
    for (int i = 0; i < n; ++i) {
        RUN(run_stack(i, enc0[i], pl->enc_blocks[i], 0, cur[i]));
        skip_buf[i] = cur[i];
        if (i + 1 < n) {                                                    // unet.py:83, pool commuted in front of the 1x1 conv
            RUN(ldm_avgpool2_f32(L.act[i][cur[i]], L.pooled, B, lv[i].H, lv[i].W, lv[i].C, st));
            ldm_gemm_desc d = gemm_rows(L.pooled, lv[i + 1].M, lv[i + 1].C, lv[i].C, pl->down_w[i], pl->down_b[i], L.act[i + 1][0]);
            allow_splitk(d, L);
            RUN(ldm_gemm_f32(&d, st));
            cur[i + 1] = 0;
        }
    }
    // ---- decoder ----------------------------------------------------------------------------------------
    for (int i = n - 1; i >= 0; --i) {
        if (i < n - 1) {                                                    // unet.py:85,101: up x2, 1x1 conv, + skip
            const int dst = (skip_buf[i] + 1) % 3;
            ldm_gemm_desc d = gemm_rows(L.act[i + 1][cur[i + 1]], lv[i + 1].M, lv[i].C, lv[i + 1].C, pl->up_w[i], pl->up_b[i], L.act[i][dst]);
            d.addend = L.act[i][skip_buf[i]]; d.ldadd = lv[i].C; d.o_mode = LDM_O_UP2; d.OH = lv[i + 1].H; d.OW = lv[i + 1].W;
            RUN(ldm_gemm_f32(&d, st));
            cur[i] = dst;
        }
        RUN(run_stack(i, dec0[i], pl->dec_blocks[i], pl->enc_blocks[i], cur[i]));
    }
    RUN(ldm_head_nchw_f32(L.act[0][cur[0]], pl->head_w, pl->head_b, out, B, lv[0].C, H * W, pl->input_channels, st));
    return LDM_OK;
}

// 1: the executor uses one stream; 2: the gated GEMM of every SwinBlock runs on a side stream beside the grouped conv / attention
// branch (bit-identical results).  Returns the previous setting; any other v only queries.
extern "C" int ldm_unet_streams(int v)
{
    if (g_unet_streams < 0) (void)side_stream();
    const int old = g_unet_streams;
    if (v == 1 || v == 2) g_unet_streams = v;
    return old;
}
