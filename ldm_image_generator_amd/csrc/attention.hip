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

int g_attn_bf16_core = 1;        // ldm_window_attention_bf16io with bf16 QKV: 1 = bf16 matrix cores (default), 0 = the fp32 16x16x4 core (A/B tests)

struct AttnP {
    const float *qkv, *bias, *xf;
    float *out;
    // bf16 sampling (autocast): the context leaves as bf16 rows (the out-projection's GEMM operand) and the float "mask" of shifted
    // windows (attention.py:40) is read from the bf16 copy of the normalised input; the attention arithmetic itself stays fp32
    unsigned short *out16;
    const unsigned short *xf16;
    const unsigned short *qkv16;   // the packed in-projection as bf16 rows (then `qkv` is NULL): q, k, v are widened exactly on load; padded
                                   // tokens take the bias rounded to bf16 -- what the bf16 projection of a zero row would have stored
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

// ---- MFMA version (L <= 48 tokens: every window the reference ever builds) -----------------------------------------
// eight consecutive values of the packed in-projection (element offset `off`) or, for a zero-padded token, of the bias (offset `boff`)
__device__ __forceinline__ void load8(const AttnP &p, bool ok, long long off, int boff, f32x4 &lo, f32x4 &hi)
{
    if (p.qkv16) {
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        if (ok) {
            const u32x4v w = *(const u32x4v *)(p.qkv16 + off);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                lo[2 * e] = __uint_as_float(w[e] << 16);
                lo[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
                hi[2 * e] = __uint_as_float(w[2 + e] << 16);
                hi[2 * e + 1] = __uint_as_float(w[2 + e] & 0xFFFF0000u);
            }
        } else {
            const f32x4 b0 = *(const f32x4 *)(p.bias + boff), b1 = *(const f32x4 *)(p.bias + boff + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = (float)(__bf16)b0[e];
                hi[e] = (float)(__bf16)b1[e];
            }
        }
    } else {
        const float *src = ok ? p.qkv + off : p.bias + boff;
        lo = *(const f32x4 *)src;
        hi = *(const f32x4 *)(src + 4);
    }
}

// One wave per (sample, window, head); v_mfma_f32_16x16x4_f32 (exact fp32) for both products.
//   S^T = K Q^T   tiles [key tile][query tile]: the A operand is K, the B operand is Q; a lane (c = lane & 15,
//                 g = lane >> 4) reads row c of the tile straight from global memory, dims [8g, 8g + 8) (the contraction
//                 order is a free permutation as long as both operands use the same one).
//   softmax       the C/D map leaves a lane with ONE query column (c) and the keys {16 kt + 4 g + e}: the row
//                 reductions are 12 values in registers plus two cross-lane steps (xor 16, xor 32).
//   O^T = V^T P^T P^T is the B operand and is ALREADY in registers in the right place (k = key 16 kt + 4 g + e on lane
//                 group g, column = query c); V^T comes from an LDS image of V (rows = keys, stride 36 floats:
//                 conflict-free); the result leaves as 16-byte stores (query c, dims [16 dt + 4 g, + 4)).
template <int NT>
__global__ __launch_bounds__(256) void window_attention_mfma_kernel(const AttnP p)
{
    constexpr int RS = 36, LT = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int L = p.L, C = p.C;
    float *Vs = smem + wave * (LT * RS + LT);
    float *Kb = Vs + LT * RS;                 // additive key bias (0, -inf, or the float "mask"); -inf past L

    const long long gw = (long long)blockIdx.x * 4 + wave;
    const bool active = gw < p.total_waves;
    const int head = (int)(gw % p.heads);
    const long long t1 = gw / p.heads;
    const int nwin = p.global ? 1 : p.nwh * p.nww;
    const int win = (int)(t1 % nwin);
    const long long b = t1 / nwin;
    const int wr = win / p.nww, wc = win - wr * p.nww;
    const long long img = b * p.H * p.W;
    const float scale = 0.17677669529663687f;        // sqrt(1/32), applied to q (torch F.multi_head_attention_forward)

    f32x4 kf[NT][2], qf[NT][2];
    long long orow[NT];
    bool qok[NT];
    if (active) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + c;
            int sy = 0, sx = 0, py, px;
            const bool ok = j < L && token_src(p, wr, wc, j, sy, sx, py, px);
            // zero-padded token: the projection of 0 is the bias (attention.py:27-28); tokens past L only need finite values
            const long long roff = (img + (long long)sy * p.W + sx) * 3 * C + head * 32;
            qok[t] = ok;
            orow[t] = (img + (long long)sy * p.W + sx) * C + head * 32;
            f32x4 qv[2];
            load8(p, ok, roff + 8 * g, head * 32 + 8 * g, qv[0], qv[1]);
            load8(p, ok, roff + C + 8 * g, C + head * 32 + 8 * g, kf[t][0], kf[t][1]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[t][u][e] = __fmul_rn(qv[u][e], scale);
        }
        for (int idx = lane; idx < LT * 4; idx += 64) {
            const int j = idx >> 2, ch = (idx & 3) * 8;
            int sy = 0, sx = 0, py, px;
            f32x4 v0{0.f, 0.f, 0.f, 0.f}, v1{0.f, 0.f, 0.f, 0.f};
            if (j < L) {
                const bool ok = token_src(p, wr, wc, j, sy, sx, py, px);
                load8(p, ok, (img + (long long)sy * p.W + sx) * 3 * C + 2 * C + head * 32 + ch, 2 * C + head * 32 + ch, v0, v1);
            }
            *(f32x4 *)(Vs + j * RS + ch) = v0;
            *(f32x4 *)(Vs + j * RS + ch + 4) = v1;
        }
        if (lane < LT) {
            float kb = -INFINITY;
            if (lane < L) {
                int sy, sx, py, px;
                const bool ok = token_src(p, wr, wc, lane, sy, sx, py, px);
                kb = 0.f;
                if (!p.global) {
                    if (p.shift == 0) {
                        kb = ok ? 0.f : -INFINITY;              // bool key_padding_mask (attention.py:31-35)
                    } else {
                        // attention.py:40: "mask" = roll(roll(x_pad)) channel 0, a float added to the logits
                        int my = (py - 2 * p.shift) % p.Hp, mx = (px - 2 * p.shift) % p.Wp;
                        my += my < 0 ? p.Hp : 0;
                        mx += mx < 0 ? p.Wp : 0;
                        if (my < p.H && mx < p.W) {
                            const long long mi = (img + (long long)my * p.W + mx) * C;
                            kb = p.xf16 ? __uint_as_float((unsigned)p.xf16[mi] << 16) : p.xf[mi];
                        } else {
                            kb = 0.f;
                        }
                    }
                }
            }
            Kb[lane] = kb;
        }
    }
    __syncthreads();
    if (!active) return;

    // ---- S^T = K Q^T --------------------------------------------------------------------------------------------
    f32x4 s[NT][NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 8; ++st)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt)
                s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][st >> 2][st & 3], qf[qt][st >> 2][st & 3], s[kt][qt], 0, 0, 0);

    // ---- softmax over the keys of each query column ------------------------------------------------------------
    float kbv[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) kbv[kt][e] = Kb[16 * kt + 4 * g + e];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[kt][qt][e] += kbv[kt][e];
                mx = fmaxf(mx, s[kt][qt][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[kt][qt][e] = expf(s[kt][qt][e] - mx);
                sum += s[kt][qt][e];
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kt][qt][e] *= inv;
    }

    // ---- O^T = V^T P^T ------------------------------------------------------------------------------------------
    f32x4 o[NT][2];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) o[qt][0] = o[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float vt = Vs[(16 * kt + 4 * g + e) * RS + 16 * dt + c];
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt, s[kt][qt][e], o[qt][dt], 0, 0, 0);
            }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
        if (qok[qt]) {                          // padded queries are cropped (attention.py:59)
            if (p.out16) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
                unsigned short *dst = p.out16 + orow[qt] + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x2v lo = {o[qt][dt][0], o[qt][dt][1]}, hi = {o[qt][dt][2], o[qt][dt][3]};
                    *(u32x2 *)(dst + 16 * dt) = u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2v)),
                                                      __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2v))};
                }
            } else {
                float *dst = p.out + orow[qt] + 4 * g;
                *(f32x4 *)dst = o[qt][0];
                *(f32x4 *)(dst + 16) = o[qt][1];
            }
        }
}


// ---- bf16 matrix-core version (bf16 QKV rows in, bf16 context out): the autocast sampling mode and the bf16 training step --------------
// Same work split (one wave per sample x window x head) and the same index arithmetic; the two products run on the bf16 matrix cores:
//   S^T = K Q^T   ONE v_mfma_f32_16x16x32_bf16 per 16 x 16 tile: the head dimension (32) is the instruction's K, and a lane's operand
//                 -- row c of the tile, dims [8 g, 8 g + 8) -- is one 16-byte load of the token's bf16 row; scores accumulate in fp32,
//                 scale and key bias are applied in fp32, softmax in fp32 (v_exp_f32);
//   O^T = V^T P^T v_mfma_f32_16x16x16_bf16: P^T (rounded once to bf16, as every 16-bit attention does) is the B operand straight from
//                 the score tile's registers (keys 16 kt + 4 g + e of query column c); V^T comes from the wave's LDS image of V
//                 ([key][32 dims] bf16, 80-byte rows) through the transposing read ds_read_b64_tr_b16 -- lane (d = lane & 15, g) receives
//                 V[16 kt + 4 g + 0..3][16 dt + d], its A operand.
// 27 matrix instructions of 8-16 cycles instead of 144 of 32: the fp32 16x16x4 core above is matrix-pipe-bound (MFMA busy 0.32 of a
// latency-bound kernel); this one is bound by its loads and the 36 exponentials per lane.
typedef short bfrag8 __attribute__((ext_vector_type(8)));
typedef short bfrag4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
typedef unsigned u32x2a __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) bfrag4 *lds_bfrag4_ptr;

__device__ __forceinline__ unsigned pack_bf16_pair(float lo, float hi)
{
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}

// eight consecutive values of the bias, rounded to bf16 (a zero-padded token's q / k / v as the bf16 projection stores them)
__device__ __forceinline__ u32x4a bias8_bf16(const float *b)
{
    const f32x4 b0 = *(const f32x4 *)b, b1 = *(const f32x4 *)(b + 4);
    return u32x4a{pack_bf16_pair(b0[0], b0[1]), pack_bf16_pair(b0[2], b0[3]), pack_bf16_pair(b1[0], b1[1]), pack_bf16_pair(b1[2], b1[3])};
}

template <int NT>
__global__ __launch_bounds__(256) void window_attention_bf16_kernel(const AttnP p)
{
    constexpr int LT = 16 * NT;
    constexpr int VS = 80;                                   // bytes per V row in LDS (64 used)
    constexpr int WB = LT * VS + LT * 4;                      // bytes per wave: V image, then the key bias
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int L = p.L, C = p.C;
    unsigned char *Vs = smem16 + wave * WB;
    float *Kb = (float *)(Vs + LT * VS);

    const long long gw = (long long)blockIdx.x * 4 + wave;
    const bool active = gw < p.total_waves;
    const int head = (int)(gw % p.heads);
    const long long t1 = gw / p.heads;
    const int nwin = p.global ? 1 : p.nwh * p.nww;
    const int win = (int)(t1 % nwin);
    const long long b = t1 / nwin;
    const int wr = win / p.nww, wc = win - wr * p.nww;
    const long long img = b * p.H * p.W;
    const float scale = 0.17677669529663687f;                // sqrt(1/32)

    u32x4a kf[NT], qf[NT];
    long long orow[NT];
    bool qok[NT];
    if (active) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + c;
            int sy = 0, sx = 0, py, px;
            const bool ok = j < L && token_src(p, wr, wc, j, sy, sx, py, px);
            const long long roff = (img + (long long)sy * p.W + sx) * 3 * C + head * 32 + 8 * g;
            qok[t] = ok;
            orow[t] = (img + (long long)sy * p.W + sx) * C + head * 32;
            if (ok) {
                qf[t] = *(const u32x4a *)(p.qkv16 + roff);
                kf[t] = *(const u32x4a *)(p.qkv16 + roff + C);
            } else {                                            // zero-padded token (or a token past L: any finite value, its key bias is -inf)
                qf[t] = bias8_bf16(p.bias + head * 32 + 8 * g);
                kf[t] = bias8_bf16(p.bias + C + head * 32 + 8 * g);
            }
        }
        for (int idx = lane; idx < LT * 4; idx += 64) {
            const int j = idx >> 2, ch = (idx & 3) * 8;
            int sy = 0, sx = 0, py, px;
            u32x4a vv{0u, 0u, 0u, 0u};
            if (j < L) {
                const bool ok = token_src(p, wr, wc, j, sy, sx, py, px);
                vv = ok ? *(const u32x4a *)(p.qkv16 + (img + (long long)sy * p.W + sx) * 3 * C + 2 * C + head * 32 + ch)
                        : bias8_bf16(p.bias + 2 * C + head * 32 + ch);
            }
            *(u32x4a *)(Vs + j * VS + ch * 2) = vv;
        }
        if (lane < LT) {
            float kb = -INFINITY;
            if (lane < L) {
                int sy, sx, py, px;
                const bool ok = token_src(p, wr, wc, lane, sy, sx, py, px);
                kb = 0.f;
                if (!p.global) {
                    if (p.shift == 0) {
                        kb = ok ? 0.f : -INFINITY;
                    } else {
                        int my = (py - 2 * p.shift) % p.Hp, mx = (px - 2 * p.shift) % p.Wp;
                        my += my < 0 ? p.Hp : 0;
                        mx += mx < 0 ? p.Wp : 0;
                        kb = (my < p.H && mx < p.W) ? __uint_as_float((unsigned)p.xf16[(img + (long long)my * p.W + mx) * C] << 16) : 0.f;
                    }
                }
            }
            Kb[lane] = kb;
        }
    }
    __syncthreads();
    if (!active) return;                                     // whole waves only: the transposing reads below need a full EXEC mask

    // ---- S^T = K Q^T, scale, key bias, softmax over the keys of each query column ---------------------------------------------
    f32x4 s[NT][NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
            s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfrag8, kf[kt]), __builtin_bit_cast(bfrag8, qf[qt]),
                                                                f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    float kbv[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) kbv[kt][e] = Kb[16 * kt + 4 * g + e];
    bfrag4 pb[NT][NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[kt][qt][e] = s[kt][qt][e] * scale + kbv[kt][e];
                mx = fmaxf(mx, s[kt][qt][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[kt][qt][e] = __expf(s[kt][qt][e] - mx);
                sum += s[kt][qt][e];
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const u32x2a w = {pack_bf16_pair(s[kt][qt][0] * inv, s[kt][qt][1] * inv), pack_bf16_pair(s[kt][qt][2] * inv, s[kt][qt][3] * inv)};
            pb[kt][qt] = __builtin_bit_cast(bfrag4, w);
        }
    }

    // ---- O^T = V^T P^T ----------------------------------------------------------------------------------------------------
    f32x4 o[NT][2];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) o[qt][0] = o[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = c >> 2, tp = c & 3;                        // this lane SUPPLIES row tq, columns 4 tp .. 4 tp + 3 of its group's 4 x 16 block
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const unsigned char *src = Vs + (16 * kt + 4 * g + tq) * VS + (16 * dt + 4 * tp) * 2;
            const bfrag4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bfrag4_ptr)src);
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pb[kt][qt], o[qt][dt], 0, 0, 0);
        }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
        if (qok[qt]) {                                        // padded queries are cropped (attention.py:59)
            unsigned short *dst = p.out16 + orow[qt] + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *(u32x2a *)(dst + 16 * dt) = u32x2a{pack_bf16_pair(o[qt][dt][0], o[qt][dt][1]), pack_bf16_pair(o[qt][dt][2], o[qt][dt][3])};
        }
}

}  // namespace

static int window_attention_impl(const float *qkv, const float *in_proj_bias, const float *xf, const void *xf16, float *out, void *out16, int B, int H,
                                 int W, int C, int ws, int shift, void *stream, bool qkv16 = false);

extern "C" int ldm_window_attention_f32(const float *qkv, const float *in_proj_bias, const float *xf, float *out, int B, int H,
                                        int W, int C, int ws, int shift, void *stream)
{
    return window_attention_impl(qkv, in_proj_bias, xf, nullptr, out, nullptr, B, H, W, C, ws, shift, stream);
}

extern "C" int ldm_window_attention_bf16_core(int v)
{
    const int old = g_attn_bf16_core;
    if (v == 0 || v == 1) g_attn_bf16_core = v;
    return old;
}

extern "C" int ldm_window_attention_bf16io(const void *qkv, int qkv_is_bf16, const float *in_proj_bias, const void *xf_bf16, void *out_bf16, int B, int H,
                                           int W, int C, int ws, int shift, void *stream)
{
    LDM_REQUIRE(out_bf16 && (((size_t)out_bf16) & 7) == 0, "ldm_window_attention_bf16io: null / unaligned output");
    return window_attention_impl((const float *)qkv, in_proj_bias, nullptr, xf_bf16, nullptr, out_bf16, B, H, W, C, ws, shift, stream, qkv_is_bf16 != 0);
}

static int window_attention_impl(const float *qkv, const float *in_proj_bias, const float *xf, const void *xf16, float *out, void *out16, int B, int H,
                                 int W, int C, int ws, int shift, void *stream, bool qkv16)
{
    LDM_REQUIRE(qkv && in_proj_bias && (out || out16), "ldm_window_attention_f32: null pointer");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 32 && C % 32 == 0, "ldm_window_attention_f32: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    LDM_REQUIRE(ws >= 1 && ws <= 8, "ldm_window_attention_f32: window_size=%d unsupported (1..8)", ws);
    LDM_REQUIRE(shift >= 0 && shift < ws, "ldm_window_attention_f32: shift=%d", shift);
    LDM_REQUIRE(ldm_aligned16(qkv) && ldm_aligned16(in_proj_bias) && (!out || ldm_aligned16(out)), "ldm_window_attention_f32: unaligned pointer");
    AttnP p{};
    p.qkv = qkv16 ? nullptr : qkv; p.qkv16 = qkv16 ? (const unsigned short *)qkv : nullptr;
    p.bias = in_proj_bias; p.xf = xf; p.out = out;
    p.out16 = (unsigned short *)out16; p.xf16 = (const unsigned short *)xf16;
    p.B = B; p.H = H; p.W = W; p.C = C; p.ws = ws; p.shift = shift;
    p.heads = C / 32;
    p.global = (H <= ws && W <= ws) ? 1 : 0;
    if (p.global) {
        p.Hp = H; p.Wp = W; p.nwh = p.nww = 1; p.L = H * W; p.shift = 0;
    } else {
        p.Hp = (H + ws - 1) / ws * ws;                       // attention.py:21-25
        p.Wp = (W + ws - 1) / ws * ws;
        p.nwh = p.Hp / ws; p.nww = p.Wp / ws; p.L = ws * ws;
        LDM_REQUIRE(shift == 0 || xf != nullptr || xf16 != nullptr, "ldm_window_attention_f32: shift != 0 needs xf (float mask source)");
    }
    p.total_waves = (long long)B * p.nwh * p.nww * p.heads;
    const unsigned blocks = (unsigned)((p.total_waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (qkv16 && out16 && p.L <= 48 && (p.global || p.shift == 0 || xf16) && g_attn_bf16_core) {        // bf16 rows in and out: the bf16 matrix cores
        LDM_REQUIRE(ldm_aligned16(qkv) && C % 8 == 0, "ldm_window_attention_bf16io: bf16 QKV rows must be 16-byte addressable");
        const int nt = (p.L + 15) / 16;
        const size_t smem = 4ull * (16 * nt * 80 + 16 * nt * 4);
        if (nt == 1) hipLaunchKernelGGL(window_attention_bf16_kernel<1>, dim3(blocks), dim3(256), smem, st, p);
        else if (nt == 2) hipLaunchKernelGGL(window_attention_bf16_kernel<2>, dim3(blocks), dim3(256), smem, st, p);
        else hipLaunchKernelGGL(window_attention_bf16_kernel<3>, dim3(blocks), dim3(256), smem, st, p);
        LDM_CHECK_LAUNCH("ldm_window_attention_bf16io");
        return LDM_OK;
    }
    if (p.L <= 16) {
        hipLaunchKernelGGL(window_attention_mfma_kernel<1>, dim3(blocks), dim3(256), 4ull * (16 * 36 + 16) * sizeof(float), st, p);
    } else if (p.L <= 32) {
        hipLaunchKernelGGL(window_attention_mfma_kernel<2>, dim3(blocks), dim3(256), 4ull * (32 * 36 + 32) * sizeof(float), st, p);
    } else if (p.L <= 48) {
        hipLaunchKernelGGL(window_attention_mfma_kernel<3>, dim3(blocks), dim3(256), 4ull * (48 * 36 + 48) * sizeof(float), st, p);
    } else {
        LDM_REQUIRE(!out16 && !xf16 && !qkv16, "ldm_window_attention_bf16io: windows of more than 48 tokens have no bf16 I/O kernel");
        const size_t smem = 4ull * (2 * 64 * 32 + 64) * sizeof(float);
        static LdmLdsOptIn opt_in;
        (void)opt_in((const void *)window_attention_kernel<64>, smem);
        hipLaunchKernelGGL(window_attention_kernel<64>, dim3(blocks), dim3(256), smem, st, p);
    }
    LDM_CHECK_LAUNCH("ldm_window_attention_f32");
    return LDM_OK;
}
