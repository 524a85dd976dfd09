// Window attention core (attention.py:13-85 + torch multi_head_attention_forward):
// pad -> (roll) -> 6x6 windows -> per-head softmax(q k^T / sqrt(d) + mask) v -> un-roll -> crop,
// all as index arithmetic on the channels-last packed QKV buffer -- no padded / rolled /
// split copies are ever materialised.
//
// One wave per (sample, window, head).  Lane i owns query token i of the window
// (L <= 64 tokens, 36 for the reference's 6x6 windows); K and V of the window/head
// live in LDS ([L][32] floats each) and are read as wave-wide broadcasts
// (ds_read_b128, every lane the same address: conflict-free), scores, softmax and the
// P.V accumulation stay in registers.  head_dim is fixed at 32 (unet.py:26).
#include "common.h"

namespace {

struct AttnP {
    const float *qkv, *bias, *xf;
    float *out;
    int B, H, W, C, ws, shift;
    int Hp, Wp, nwh, nww, heads, L;
    int global;     // H<=ws && W<=ws: one window of H*W tokens, no mask
    long long total_waves;
};

// window token -> source pixel; returns false for padded tokens
__device__ __forceinline__ bool token_src(const AttnP &p, int wr, int wc, int j, int &sy, int &sx, int &py, int &px)
{
    if (p.global) {
        sy = py = j / p.W;
        sx = px = j - sy * p.W;
        return true;
    }
    const int wy = j / p.ws, wx = j - wy * p.ws;
    py = wr * p.ws + wy;
    px = wc * p.ws + wx;
    sy = py - p.shift;                     // x_rolled[py] = x_pad[py - shift]  (attention.py:39)
    sy += sy < 0 ? p.Hp : 0;
    sx = px - p.shift;
    sx += sx < 0 ? p.Wp : 0;
    return sy < p.H && sx < p.W;
}

template <int LMAX>
__global__ __launch_bounds__(256) void window_attention_kernel(const AttnP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = p.L, C = p.C;
    float *Ks = smem + wave * (2 * LMAX * 32 + LMAX);
    float *Vs = Ks + LMAX * 32;
    float *Kb = Vs + LMAX * 32;            // additive key bias (0, -inf, or the float "mask")

    const long long gw = (long long)blockIdx.x * 4 + wave;
    const bool active = gw < p.total_waves;
    const int head = (int)(gw % p.heads);
    const long long t1 = gw / p.heads;
    const int nwin = p.global ? 1 : p.nwh * p.nww;
    const int win = (int)(t1 % nwin);
    const long long b = t1 / nwin;
    const int wr = win / p.nww, wc = win - wr * p.nww;
    const long long img = b * p.H * p.W;

    if (active) {
        // ---- stage K, V of this (window, head) ------------------------------
        for (int idx = lane; idx < L * 8; idx += 64) {
            const int j = idx >> 3, ch = (idx & 7) * 4;
            int sy, sx, py, px;
            const bool ok = token_src(p, wr, wc, j, sy, sx, py, px);
            f32x4 kv, vv;
            if (ok) {
                const float *row = p.qkv + (img + (long long)sy * p.W + sx) * 3 * C + head * 32 + ch;
                kv = *(const f32x4 *)(row + C);
                vv = *(const f32x4 *)(row + 2 * C);
            } else {                        // zero-padded token: projection of 0 is the bias (attention.py:27-28)
                kv = *(const f32x4 *)(p.bias + C + head * 32 + ch);
                vv = *(const f32x4 *)(p.bias + 2 * C + head * 32 + ch);
            }
            *(f32x4 *)(Ks + j * 32 + ch) = kv;
            *(f32x4 *)(Vs + j * 32 + ch) = vv;
        }
        if (lane < L) {
            int sy, sx, py, px;
            const bool ok = token_src(p, wr, wc, lane, sy, sx, py, px);
            float kb = 0.f;
            if (!p.global) {
                if (p.shift == 0) {
                    kb = ok ? 0.f : -INFINITY;              // bool key_padding_mask (attention.py:31-35)
                } else {
                    // attention.py:40: "mask" = roll(roll(x_pad)) channel 0, a float added to the logits
                    int my = (py - 2 * p.shift) % p.Hp, mx = (px - 2 * p.shift) % p.Wp;
                    my += my < 0 ? p.Hp : 0;
                    mx += mx < 0 ? p.Wp : 0;
                    kb = (my < p.H && mx < p.W) ? p.xf[(img + (long long)my * p.W + mx) * C] : 0.f;
                }
            }
            Kb[lane] = kb;
        }
    }
    __syncthreads();
    if (!active || lane >= L) return;

    // ---- this lane's query --------------------------------------------------
    int sy, sx, py, px;
    const bool qok = token_src(p, wr, wc, lane, sy, sx, py, px);
    if (!qok) return;                       // padded queries are cropped (attention.py:59)
    const float *qrow = p.qkv + (img + (long long)sy * p.W + sx) * 3 * C + head * 32;
    const float scale = 0.17677669529663687f;        // sqrt(1/32), applied to q (torch F.multi_head_attention_forward)
    float q[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 v = *(const f32x4 *)(qrow + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[4 * c + e] = __fmul_rn(v[e], scale);
    }
    float s[LMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j < L) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 kv = *(const f32x4 *)(Ks + j * 32 + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) a = fmaf(q[4 * c + e], kv[e], a);
            }
            a += Kb[j];
            s[j] = a;
            mx = fmaxf(mx, a);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j < L) {
            s[j] = expf(s[j] - mx);
            sum += s[j];
        }
    }
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j < L) {
            const float pj = s[j] / sum;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 vv = *(const f32x4 *)(Vs + j * 32 + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * c + e] = fmaf(pj, vv[e], o[4 * c + e]);
            }
        }
    }
    float *orow = p.out + (img + (long long)sy * p.W + sx) * C + head * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) *(f32x4 *)(orow + 4 * c) = f32x4{o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]};
}

}  // namespace

extern "C" int ldm_window_attention_f32(const float *qkv, const float *in_proj_bias, const float *xf, float *out, int B, int H,
                                        int W, int C, int ws, int shift, void *stream)
{
    LDM_REQUIRE(qkv && in_proj_bias && out, "ldm_window_attention_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0, "ldm_window_attention_f32: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    LDM_REQUIRE(ws >= 1 && ws <= 8, "ldm_window_attention_f32: window_size=%d unsupported (1..8)", ws);
    LDM_REQUIRE(shift >= 0 && shift < ws, "ldm_window_attention_f32: shift=%d", shift);
    LDM_REQUIRE(ldm_aligned16(qkv) && ldm_aligned16(in_proj_bias) && ldm_aligned16(out), "ldm_window_attention_f32: unaligned pointer");
    AttnP p{};
    p.qkv = qkv; p.bias = in_proj_bias; p.xf = xf; p.out = out;
    p.B = B; p.H = H; p.W = W; p.C = C; p.ws = ws; p.shift = shift;
    p.heads = C / 32;
    p.global = (H <= ws && W <= ws) ? 1 : 0;
    if (p.global) {
        p.Hp = H; p.Wp = W; p.nwh = p.nww = 1; p.L = H * W; p.shift = 0;
    } else {
        p.Hp = (H + ws - 1) / ws * ws;                       // attention.py:21-25
        p.Wp = (W + ws - 1) / ws * ws;
        p.nwh = p.Hp / ws; p.nww = p.Wp / ws; p.L = ws * ws;
        LDM_REQUIRE(shift == 0 || xf != nullptr, "ldm_window_attention_f32: shift != 0 needs xf (float mask source)");
    }
    p.total_waves = (long long)B * p.nwh * p.nww * p.heads;
    const unsigned blocks = (unsigned)((p.total_waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (p.L <= 36) {
        const size_t smem = 4ull * (2 * 36 * 32 + 36) * sizeof(float);
        hipLaunchKernelGGL(window_attention_kernel<36>, dim3(blocks), dim3(256), smem, st, p);
    } else {
        const size_t smem = 4ull * (2 * 64 * 32 + 64) * sizeof(float);
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void *)window_attention_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            attr_done = true;
        }
        hipLaunchKernelGGL(window_attention_kernel<64>, dim3(blocks), dim3(256), smem, st, p);
    }
    LDM_CHECK_LAUNCH("ldm_window_attention_f32");
    return LDM_OK;
}
