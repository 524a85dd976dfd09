/*
 * ldm_hip.h -- C ABI of the MI355X (gfx950) latent-diffusion hot path.
 *
 * The reference (uthree/ldm-image-generator) has no FFI/plugin layer: its hot
 * path is PyTorch ATen ops issued from unet.py / attention.py / modules.py /
 * sinusoidal.py / vae.py / ddpm.py.  This header is the boundary a maintainer
 * binds instead of those ATen calls (ctypes stub in INTEGRATION.md).  Each entry
 * point names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers to fp32
 *     unless stated; the caller (PyTorch) owns every buffer, incl. workspaces;
 *   - activations are channels-last: [B, H, W, C] == [M = B*H*W rows, C cols];
 *     the NCHW tensors of the reference cross the boundary only in
 *     ldm_stem_nchw_f32 / ldm_head_nchw_f32 / ldm_rgb_head_f32 / ldm_nchw_to_nhwc_f32;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) and is re-entrant; safe under hipGraph capture;
 *   - return 0 on success, a negative LDM_E* code otherwise (never throws);
 *     ldm_last_error() returns a thread-local message for the last failure.
 */
#ifndef LDM_HIP_H
#define LDM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LDM_OK          0
#define LDM_EINVAL     -1   /* bad shape / alignment / null pointer            */
#define LDM_ELAUNCH    -2   /* hipLaunch / runtime error (see ldm_last_error)  */
#define LDM_ENODEV     -3   /* no gfx950 device                                */

#define LDM_MAX_SEG     4
#define LDM_MAX_TABLE   32
#define LDM_MAX_LEVELS  8

/* epilogue activation of ldm_gemm_f32 */
#define LDM_ACT_NONE    0
#define LDM_ACT_RELU    1   /* unet.py:13,20 (Encodings.act)                   */
#define LDM_ACT_GATE    2   /* modules.py:15  a(x) * relu(b(x))                */
#define LDM_ACT_LRELU   3   /* vae.py:62,64   F.leaky_relu (slope in desc)     */

/* A-operand addressing */
#define LDM_A_ROWS      0   /* row m at a + m*lda (1x1 conv / Linear)          */
#define LDM_A_CONV3X3   1   /* implicit im2col, zero pad 1 (vae.py:57-58, unet.py:30) */

/* output addressing */
#define LDM_O_ROWS      0   /* out[m*ldo + n]                                  */
#define LDM_O_CONVT2X2  1   /* ConvTranspose2d(k=2,s=2) scatter (vae.py:120)   */
#define LDM_O_UP2       2   /* nearest x2 replicate (+addend at the fine grid) (unet.py:85,101) */

/* which segment dimension the weight pointers split */
#define LDM_SEG_N       0   /* w[s] holds rows [s*seg_len, (s+1)*seg_len) of the N axis */
#define LDM_SEG_K       1   /* w[s] holds K-columns [s*seg_len, ...) for all N rows      */

/*
 * One fp32 GEMM launch on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *
 *     out = act(A . W^T + bias) (+ addend)
 *
 * A is [M, K] (rows or implicit 3x3 im2col), every weight segment is a
 * row-major [rows, K-columns] matrix with row stride ldw (i.e. the reference's
 * Conv2d / Linear weight as stored in its state_dict, K contiguous).
 * Used for: ReGLU/RandomMoE (modules.py:14-15,34-36, weights selected by
 * pointer, never copied), Encodings MLP (unet.py:20), MHA in/out projections
 * (attention.py:82 -> torch multi_head_attention_forward), the 1x1 ch_convs
 * (unet.py:83,85), grouped 3x3 conv (unet.py:30,44; groups on grid.y), the VAE
 * dense 3x3 convs, ConvTranspose2d 2x2 and input_layer (vae.py:57-58,103,120).
 */
typedef struct ldm_gemm_desc {
    const float *a;          /* A operand                                        */
    long long    lda;        /* row stride of A in floats (LDM_A_ROWS) / pixel stride (CONV3X3) */
    int          M, N, K;    /* K = Cin (ROWS) or 9*Cin (CONV3X3); N, K%32 == 0  */
    int          a_mode;     /* LDM_A_*                                          */
    int          H, W, Cin;  /* CONV3X3: spatial dims of A, M == B*H*W           */

    int          nseg;       /* 1..LDM_MAX_SEG weight segments                   */
    int          seg_mode;   /* LDM_SEG_*                                        */
    int          seg_len;    /* extent of one segment along its axis             */
    const float *w[LDM_MAX_SEG];     /* weight segments                          */
    const float *w2[LDM_MAX_SEG];    /* LDM_ACT_GATE: the "b" weights            */
    const float *bias[LDM_MAX_SEG];  /* may be NULL                              */
    const float *bias2[LDM_MAX_SEG]; /* LDM_ACT_GATE: the "b" biases             */
    long long    ldw;        /* weight row stride in floats                      */

    int          act;        /* LDM_ACT_*                                        */
    float        slope;      /* LDM_ACT_LRELU                                    */

    const float *addend;     /* optional, same addressing as out                 */
    long long    ldadd;
    float       *out;
    long long    ldo;
    int          o_mode;     /* LDM_O_*                                          */
    int          OH, OW;     /* CONVT2X2 / UP2: coarse grid, M == B*OH*OW        */
    int          Cout;       /* CONVT2X2: N == 4*Cout, n = (dy*2+dx)*Cout + co   */

    int          groups;     /* >=1; group g runs on grid.y with the offsets below */
    long long    a_gstride;  /* floats added to a per group                       */
    long long    w_gstride;  /* floats added to every w[s] / w2[s] per group      */
    long long    o_gstride;  /* floats added to out and addend per group          */
    long long    b_gstride;  /* floats added to every bias pointer per group      */
    /* pointer-table mode (independent layers batched into one launch, e.g. the
     * Encodings MLPs of all SwinBlocks of one UNet level): HOST arrays of `groups`
     * (<= LDM_MAX_TABLE) DEVICE pointers that replace w[0] / bias[0] for group g
     * (nseg must be 1).  They are copied into the kernel arguments, so the kernel
     * reads them with scalar loads and no device-side table has to exist. */
    const float *const *w_table;
    const float *const *bias_table;
    /* optional DEVICE scratch: lets few-tile / long-K problems (M <= 128) split the reduction over the grid
     * and sum fp32 partial tiles in a fixed order (deterministic); NULL = never split */
    void        *workspace;
    long long    workspace_bytes;
} ldm_gemm_desc;

int         ldm_version(void);
const char *ldm_last_error(void);
int         ldm_device_ok(void);          /* 1 if device 0 is gfx950 */
/* The library keeps one grow-only device scratch per (device, stream) for the fixed-order partial sums of its reductions (loss scalars,
 * bias / column sums, stem / head weight gradients): no float atomics, results are bit-reproducible from run to run.  Frees them all. */
int         ldm_scratch_release(void);

int ldm_gemm_f32(const ldm_gemm_desc *d, void *stream);
/* schedule used by ldm_gemm_f32: 0 = one tile per workgroup, 1 = persistent LDS-DMA stream (default); both use
 * v_mfma_f32_32x32x2_f32 and give bit-identical results.  2 = the stream schedule with the "split" consumer:
 * each fp32 operand value is cut exactly into three bf16 pieces in registers and a product is six
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (fp32-level error, not bit-identical to 0/1; shapes it does
 * not cover run as 1).  Opt-in, process-wide; returns the previous setting (any other v only queries). */
/* ReGLU forward of the fp32 training step in one launch (unet.py:14-15 and what its autograd keeps): d as for ldm_gemm_f32 with
 * act = LDM_ACT_GATE, rows in, rows out, no addend; out = (A Wa^T + ba) * relu(A Wb^T + bb) and a_pre, b_pre = the two pre-activations, all
 * fp32 [M, ldo].  Bit-identical to two plain ldm_gemm_f32 launches + ldm_gate_fwd_f32.  Returns 0 when launched, 1 when no kernel
 * instance takes the shape (nothing was launched: run the three launches instead), negative on error. */
int ldm_gemm_f32_gate_fwd(const ldm_gemm_desc *d, float *a_pre, float *b_pre, void *stream);
/* ReGLU backward of the fp32 training step in the epilogue of dh = dY . Wc (d: that plain GEMM, rows in / out, no activation, no addend):
 * d->out = da = dh * relu(b_pre), db = dh * a_pre * (b_pre > 0), all fp32 [M, ldo]; dh itself is never stored.  Bit-identical to
 * ldm_gemm_f32 + ldm_gate_bwd_f32.  Returns 0 when launched, 1 when no kernel instance takes the shape, negative on error. */
int ldm_gemm_f32_gate_bwd(const ldm_gemm_desc *d, const float *a_pre, const float *b_pre, float *db, void *stream);
int ldm_gemm_variant(int v);
/* epilogue of the stream schedule for plain-rows outputs: 1 (default) = through LDS, 16 bytes per lane per store;
 * 0 = direct from the MFMA layout, 4 bytes per lane.  Bit-identical results; A/B knob.  Returns the previous setting. */
int ldm_gemm_wide_epilogue(int v);
/* schedule of ldm_gemm_f32 / ldm_gemm_bf16 for rows-in / rows-out problems with M a multiple of 256 and N (and N-segments) a
 * multiple of 128: 1 (default) = the one-workgroup-per-CU ring kernel (256 x 256 / 256 x 128 tiles, four-stage LDS ring) when at
 * least 192 tiles exist, 0 = always the 128-row stream kernel, 2 = the ring kernel whenever the shape is legal, 3 = like 2 on at
 * most eight workgroups (tests: many tiles per workgroup).  Bit-identical results; A/B and test knob.  Returns the previous setting (other v only queries). */
int ldm_gemm_ring(int v);

/* hot-kernel timing for bench.py: when enabled every ldm_gemm_f32 launch is
 * bracketed by hipEvents on ITS stream; ldm_prof_read synchronises those events
 * and returns launches, summed kernel milliseconds and summed algorithmic FLOPs. */
int ldm_prof_enable(int on);
int ldm_prof_read(long long *launches, double *ms, double *flops);     /* all classes; clears the records */
/* kernel classes: 0 ldm_gemm_f32, 1 ldm_gemm_tn_f32, 2 ldm_gconv3x3_wgrad_f32, 3 ldm_gemm_bf16, 4 ldm_gemm_tn_bf16,
 * 5 grouped conv with bf16 operands; cls < 0 = all.  Does not clear (call ldm_prof_read last). */
int ldm_prof_read_class(int cls, long long *launches, double *ms, double *flops);
int ldm_prof_read_bytes(int cls, double *bytes);    /* summed algorithmic HBM bytes of the class's launches (operands once) */
/* every record in launch order: out[4 i ..] = (class, kernel ms, algorithmic FLOPs, algorithmic bytes); returns the count (<= max_records) */
long long ldm_prof_dump(double *out, long long max_records);

/* modules.py:23-25 (ChannelNorm, unbiased var, eps inside sqrt) fused with the
 * FiLM of unet.py:22.  film is [nslot, HW, 2C] (mul | bias); slot[b] selects the
 * time slot of sample b (NULL: slot 0 for everyone). */
int ldm_channelnorm_film_f32(const float *x, const float *film, const int *slot, float *out,
                             int B, int HW, int C, float eps, void *stream);
/* unet.py:22 alone: out = x * mul + bias (Encodings.forward without the norm). */
int ldm_film_f32(const float *x, const float *film, const int *slot, float *out, int B, int HW, int C, void *stream);

/* sinusoidal.py:12-19,31-38 + the cat of unet.py:19: writes emb[nT, H*W, 2C] =
 * cat[PositionalEncoding2d(h,w), TimeEncoding2d(t)].  pos_freq[C/4] and
 * time_freq[C/2] are the reference's frequency tables (host-computed). */
int ldm_sincos_embed_f32(const long long *t, int nT, int H, int W, int C,
                         const float *pos_freq, const float *time_freq, float *emb, void *stream);

/* attention.py:13-85 core (pad, roll, window split, per-head softmax(q k^T/sqrt(d) + mask) v,
 * un-roll, crop).  qkv [B,H,W,3C] is the packed in-projection of the UNPADDED
 * tokens; padded tokens are synthesised from in_proj_bias.  shift==0 uses the
 * boolean padding mask; shift!=0 reproduces attention.py:40's float "mask"
 * (channel 0 of the twice-rolled input xf) as an additive key bias.  If
 * H<=ws && W<=ws attention is global and unmasked (attention.py:15-16). */
int ldm_window_attention_f32(const float *qkv, const float *in_proj_bias, const float *xf, float *out,
                             int B, int H, int W, int C, int ws, int shift, void *stream);

/* nn.AvgPool2d(2) on [B,H,W,C] (unet.py:83; commuted in front of the 1x1 conv). */
int ldm_avgpool2_f32(const float *x, float *out, int B, int H, int W, int C, void *stream);

/* unet.py:77,90  stem Conv2d(Cin, C0, 1): NCHW in -> NHWC out.  w [C0, Cin]. */
int ldm_stem_nchw_f32(const float *x, const float *w, const float *bias, float *out,
                      int B, int Cin, int HW, int C0, void *stream);
/* unet.py:78,102 head ConvTranspose2d(C0, Cin, 1): NHWC in -> NCHW out.  w [C0, Cin]. */
int ldm_head_nchw_f32(const float *x, const float *w, const float *bias, float *out,
                      int B, int C0, int HW, int Cin, void *stream);

/* ddpm.py:81-91 one DDIM update, elementwise, in place on x:
 *   x0 = (x - s1*e)/s2 ;  x = last ? x0 : s3*x0 + s4*e + sigma*noise
 * (each product and sum rounded separately, as torch evaluates it). noise may be NULL when sigma == 0. */
int ldm_ddim_update_f32(float *x, const float *e_theta, const float *noise, long long n,
                        float s1, float s2, float s3, float s4, float sigma, int last, void *stream);

/* ddpm.py:46 q-sample: out = sqrt(ab[b])*x + sqrt(1-ab[b])*e, per-sample scalars precomputed by the host. */
int ldm_qsample_f32(const float *x, const float *e, const float *sa, const float *sb, float *out,
                    int B, long long per_sample, void *stream);

/* vae.py:105,113,131: rgb = to_rgb(x) (1x1, C->3) and
 * out = bilinear_x2(prev) + rgb  (prev NULL: out = rgb).  x is NHWC [B,H,W,C];
 * prev [B,3,H/2,W/2] and out [B,3,H,W] are NCHW like the reference's result. */
int ldm_rgb_head_f32(const float *x, const float *w, const float *bias, const float *prev, float *out,
                     int B, int H, int W, int C, void *stream);
/* the same with OC output channels (DecoderStack(channels, num_layers, output_channels), vae.py:100-103; 1 <= OC <= 4): w [OC, C], bias [OC],
 * prev / out [B, OC, ., .]; ldm_rgb_head_f32 is OC = 3 */
int ldm_rgb_head_oc_f32(const float *x, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, int OC,
                        void *stream);

/* layout plumbing at the boundary */
int ldm_nchw_to_nhwc_f32(const float *x, float *out, int B, int C, int HW, void *stream);
int ldm_nhwc_to_nchw_f32(const float *x, float *out, int B, int C, int HW, void *stream);

/* sample_ldm.py:75-77: clamp(-1,1) -> *127.5+127.5 -> uint8 truncation, NCHW -> NHWC bytes. */
int ldm_to_uint8_hwc(const float *img, unsigned char *out, int B, int C, int HW, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Native executor of one UNet forward (unet.py:89-103): the per-step body of the denoise loop as ONE call.
 * The plan is a caller-owned description (device pointers to the reference's parameters in their
 * state_dict layout, plus the re-laid-out grouped-conv weight); nothing is copied or retained.
 * decisions[k] for block k in execution order (encoder stages, then decoder stages): -1 = skipped by
 * stochastic depth (unet.py:39), else e1 * 4 + e2 with the two expert indices drawn by
 * random.sample (modules.py:35).  Results are bit-identical to issuing the same launches one by one.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ldm_unet_block {
    int attention, shift;                    /* window attention present / its shift (unet.py:55-57)       */
    const float *conv_w, *conv_b;            /* grouped 3x3, packed [C][tap][32]; bias [C]                */
    const float *enc_w1, *enc_b1, *enc_w2, *enc_b2;          /* Encodings.proj1 [4C,2C], proj2 [2C,4C]      */
    const float *a_w[5], *a_b[5], *b_w[5], *b_b[5], *c_w[5], *c_b[5];   /* ReGLUs: [0] general, [1..4] experts */
    const float *in_w, *in_b, *out_w, *out_b;                /* MultiheadAttention in_proj / out_proj        */
} ldm_unet_block;

typedef struct ldm_unet_plan {
    int levels, input_channels, window, nblocks;
    float eps;
    int channels[LDM_MAX_LEVELS], enc_blocks[LDM_MAX_LEVELS], dec_blocks[LDM_MAX_LEVELS];
    const float *stem_w, *stem_b, *head_w, *head_b;          /* encoder_first [C0,Cin]; decoder_last [C0,Cin] */
    const float *down_w[LDM_MAX_LEVELS], *down_b[LDM_MAX_LEVELS];   /* encoder ch_conv of level i -> i+1     */
    const float *up_w[LDM_MAX_LEVELS], *up_b[LDM_MAX_LEVELS];       /* decoder ch_conv of level i+1 -> i     */
    const float *pos_freq[LDM_MAX_LEVELS], *time_freq[LDM_MAX_LEVELS];
    const ldm_unet_block *blocks;            /* HOST array, execution order                                   */
} ldm_unet_plan;

size_t ldm_unet_workspace_bytes(const ldm_unet_plan *plan, int B, int H, int W, int nT);
int ldm_unet_forward_f32(const ldm_unet_plan *plan, const float *x_nchw, const long long *t_unique, int nT, const int *slot,
                         const int *decisions /* host */, int B, int H, int W, void *workspace, size_t ws_bytes,
                         float *out_nchw, void *stream);
/* the same forward; films_ready != 0 skips the FiLM tables because an earlier call on THIS workspace with the same plan, B, H, W and
 * (t_unique, nT) left them there (DDPM.sample computes the tables of all its timesteps in the first denoise step: they depend on t,
 * never on x; each step then passes slot[b] = its step index).  The caller owns that guarantee. */
int ldm_unet_forward_ex_f32(const ldm_unet_plan *plan, const float *x, const long long *t_unique, int nT, const int *slot,
                            const int *decisions, int B, int H, int W, void *workspace, size_t workspace_bytes, float *out,
                            int films_ready, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training step (ddpm.py:39-48 + autograd of unet.py / modules.py / attention.py).  GEMM-shaped
 * gradients go through ldm_gemm_f32 (data grads with transposed weights, weight grads as split-K
 * groups over transposed activations); the entry points below are the non-GEMM pieces.
 * ------------------------------------------------------------------------------------------------ */
int ldm_gate_fwd_f32(const float *a, const float *b, float *out, long long n, void *stream);          /* a * relu(b) */
int ldm_gate_bwd_f32(const float *dh, const float *a, const float *b, float *da, float *db, long long n, void *stream);
int ldm_relu_bwd_f32(const float *dy, const float *y, float *dx, long long n, void *stream);
int ldm_add_f32(float *y, const float *x, long long n, void *stream);                                 /* y += x */
int ldm_colsum_f32(const float *x, float *out, long long M, int N, int accumulate, void *stream);     /* bias grads */
/* out[Cc, R] = x[R, Cc]^T and csum[c] = sum_r x[r][c] in one pass (activation transpose + bias gradient) */
int ldm_transpose_colsum_f32(const float *x, float *out, float *csum, long long R, int Cc, void *stream);
int ldm_reduce_partials_f32(const float *parts, float *out, int S, long long n, void *stream);        /* split-K sum */
/* two such sums with the same S in one launch (a weight gradient's partial planes and its bias gradient's); n_a, n_b multiples of 4;
 * each result equals ldm_reduce_partials_f32's bit for bit.  seg_len_a > 0: sum a is a [n_a / row_len_a, row_len_a] matrix whose column
 * blocks of seg_len_a belong to different parameters (the c-weights of a block's three ReGLUs, unet.py:20-27): it is stored as
 * [row_len_a / seg_len_a][rows][seg_len_a], every block a contiguous tensor; 0: plain */
int ldm_reduce_partials_pair_f32(const float *parts_a, float *out_a, long long n_a, const float *parts_b, float *out_b, long long n_b, int S,
                                 long long row_len_a, long long seg_len_a, void *stream);
/* Weight gradient of a 1x1 conv / Linear WITHOUT transposed copies (autograd of modules.py:10-12, unet.py:20-21,
 * attention in/out projections): out[s][n][k] = sum over rows m of split s of a[m*lda + n] * b[m*ldb + k], i.e.
 * dW = dY^T X with the pixel rows as the contraction.  N, K multiples of 128; M a multiple of 32; split s takes rows [s ms, (s + 1) ms)
 * with ms = M / splits rounded up to a multiple of 32 (the last split takes what is left, at least 32 rows); the caller
 * sums the `splits` partial planes (ldm_reduce_partials_f32) -- fixed order, deterministic.  colsum_a (optional,
 * [splits][N]) receives the column sums of `a` per split: the bias gradient that goes with dW. */
int ldm_gemm_tn_f32(const float *a, long long lda, const float *b, long long ldb, float *out, float *colsum_a, int M, int N, int K,
                    int splits, void *stream);
/* Weight gradient of a dense 3x3 conv (zero pad 1; autograd of vae.py:57-58) without the im2col matrix -- the same kernel with an implicit
 * B operand: out[s][n][tap * Cin + ci] = sum over the pixels m of split s of dy[m * lda + n] * x[(pixel m shifted by tap) * Cin + ci].
 * out is [splits][Npad][Kpad], Npad = ldm_conv3x3_wgrad_npad(Cout) (Cout rounded up to the tile height: 32 for Cout <= 32, 64 for Cout <= 64,
 * else multiples of 128) and Kpad = 9 * Cin rounded up to a multiple of 128 (the padding comes out as zeros); the caller
 * sums the planes and takes the [Cout][9 * Cin] corner.  colsum_dy: optional [splits][Npad].  B*H*W a multiple of 32; splits as in ldm_gemm_tn_f32
 * (rows per split rounded up to 32, the last split shorter); H*W <= 2^24. */
int ldm_conv3x3_wgrad_npad(int Cout);
int ldm_conv3x3_wgrad_f32(const float *dy, long long lda, const float *x, float *out, float *colsum_dy, int B, int H, int W, int Cin, int Cout,
                          int splits, void *stream);
/* backward of ldm_channelnorm_film_f32: dx = dres + dnorm(dxf * mul); dfilm (mul | bias) per (slot, pixel): accumulated
 * atomically into a zeroed buffer when samples may share a slot, or -- unique_slots != 0: every sample has its own
 * slot -- written once with plain stores (no zeroing needed) */
int ldm_channelnorm_film_bwd_f32(const float *x, const float *film, const int *slot, const float *dxf, const float *dres,
                                 float *dx, float *dfilm, int B, int HW, int C, float eps, int unique_slots, void *stream);
int ldm_avgpool2_bwd_f32(const float *dlo, float *dx, int B, int H, int W, int C, int accumulate, void *stream);
int ldm_sumpool2_f32(const float *dhi, float *dlo, int B, int H, int W, int C, void *stream);         /* backward of nearest x2 */
int ldm_stem_bwd_f32(const float *x, const float *dy, float *dw, int B, int Cin, int HW, int C0, void *stream);
int ldm_head_bwd_f32(const float *x, const float *w, const float *dout, float *dx, float *dw, float *db,
                     int B, int C0, int HW, int Cin, void *stream);
/* nn.L1Loss (ddpm.py:16,47): loss[0] = mean |pred - target|; grad = sign(pred - target) * gscale[0] / n */
int ldm_l1_loss_f32(const float *pred, const float *target, long long n, float *loss, void *stream);
int ldm_l1_loss_bwd_f32(const float *pred, const float *target, const float *gscale, float *grad, long long n, void *stream);
/* transposed im2col of the grouped 3x3 conv input: out[g][tap*32+ci][m] (weight gradient of unet.py:30) */
int ldm_im2col3x3_t_f32(const float *x, float *out, int B, int H, int W, int C, void *stream);
/* weight gradient of the grouped 3x3 conv (32 in / 32 out per group; autograd of unet.py:30,44) straight from the
 * row-major activations: out_planes[(s * 4 + w)][C][288] (288 = tap * 32 + ci, the forward's weight layout), s < splits,
 * w < 4 -- the caller sums the 4 * splits planes (ldm_reduce_partials_f32).  B*H*W must split into `splits` runs of a
 * multiple of 128 pixels; 2 <= W <= 96. */
int ldm_gconv3x3_wgrad_f32(const float *x, const float *dy, float *out_planes, int B, int H, int W, int C, int splits, void *stream);
/* backward of ldm_window_attention_f32: dqkv [B,H,W,3C]; gradients of zero-padded tokens' k, v go to dbias_pad [3C] */
int ldm_window_attention_bwd_f32(const float *qkv, const float *in_proj_bias, const float *xf, const float *dctx, float *dqkv,
                                 float *dbias_pad, int B, int H, int W, int C, int ws, int shift, void *stream);
/* kernel behind ldm_window_attention_bwd_f32: 1 (default) v_mfma_f32_16x16x4_f32 products, 0 the scalar kernel (A/B tests).
 * Returns the old value; any other argument only queries. */
int ldm_window_attention_bwd_mfma(int v);

/* ------------------------------------------------------------------------------------------------
 * bf16 training step (BASELINE.json configs[4]: "train_ldm.py ... bf16 ... fwd+bwd HIP kernels"; the reference
 * trains under reduced-precision autocast with fp32 master weights, train_ldm.py:68,80).  GEMM operands are rounded
 * ONCE to bf16 (round-to-nearest-even) by their producers; products accumulate in fp32 (v_mfma_f32_32x32x16_bf16);
 * the residual stream, FiLM rows, every parameter gradient and the optimizer stay fp32.  bf16 buffers travel as
 * `void *` (2 bytes per element, the high half of the fp32 encoding).
 * ------------------------------------------------------------------------------------------------ */
/* ldm_gemm_f32's descriptor with bf16 operands: d->a, d->w[s] point to bf16; d->lda, d->ldw, d->K, K-segment lengths and
 * the a/w group strides count bf16 elements (K, N multiples of 64; rows 16-byte addressable).  bias / addend are fp32.
 * out_bf16 == 0: d->out (and d->addend, if any) is fp32 [M, ldo];  != 0: d->out is bf16 [M, ldo] and d->addend, if any, is a bf16
 * [M, ldadd] matrix added in fp32 before the one rounding (the VAE ResBlock skip of the bf16 decode mode, vae.py:65).
 * d->a_mode == LDM_A_CONV3X3 (bf16 output only): dense 3x3, zero pad 1, over bf16 rows [M = B*H*W, Cin], Cin a multiple of 64,
 * K = 9*Cin, weights packed [N][tap][Cin] bf16 (vae.py:57-58 under autocast).  Rows out only, no gate. */
int ldm_gemm_bf16(const ldm_gemm_desc *d, int out_bf16, void *stream);
/* ldm_window_attention_bwd_f32 with bf16 rows around the fp32 core: qkv [B,H,W,3C], xf (float "mask" source, NULL if shift == 0) and
 * dctx [B,H,W,C] are bf16, dqkv [B,H,W,3C] leaves as bf16; zero-padded tokens take the bias rounded to bf16 (what the bf16
 * in-projection of a zero row stores); dbias_pad [3C] stays fp32. */
int ldm_window_attention_bwd_bf16(const void *qkv, const float *in_proj_bias, const void *xf, const void *dctx, void *dqkv, float *dbias_pad,
                                  int B, int H, int W, int C, int ws, int shift, void *stream);
/* core of ldm_window_attention_bwd_bf16: 1 (default) = all seven products on the bf16 matrix cores (P and dS rounded once to bf16 as the
 * operands of the token contractions; scores, softmax, dS in fp32), 0 = the fp32 16x16x4 core on the widened values.  Returns the old value. */
int ldm_window_attention_bwd_bf16_core(int v);
/* ReGLU forward in one launch (modules.py:14-15): d->act == LDM_ACT_GATE, d->w / d->w2 (+ bias / bias2) the "a" / "b" branches,
 * d->out = a * relu(b) as bf16 [M, ldo]; a_pre / b_pre (both or neither): bf16 [M, ldo] copies of the pre-activations a, b that
 * the backward needs.  ldm_gemm_bf16_gate_bwd is the data-gradient GEMM dh = dy . Wc (d as for ldm_gemm_bf16) with the gate's
 * backward in its epilogue: d->out = da = dh * relu(b), db = dh * a * (b > 0) (all bf16 [M, ldo]); dh itself is never stored. */
int ldm_gemm_bf16_gate_fwd(const ldm_gemm_desc *d, void *a_pre, void *b_pre, void *stream);
int ldm_gemm_bf16_gate_bwd(const ldm_gemm_desc *d, const void *a_pre, const void *b_pre, void *db, void *stream);
/* weight gradient from bf16 operands as they lie in memory (ldm_gemm_tn_f32's contract; M / splits a multiple of 64):
 * out[s][n][k] fp32 = sum over the rows m of split s of a[m*lda + n] * b[m*ldb + k]; colsum_a optional [splits][N] fp32. */
int ldm_gemm_tn_bf16(const void *a, long long lda, const void *b, long long ldb, float *out, float *colsum_a, int M, int N, int K, int splits,
                     void *stream);
/* kernel behind ldm_gemm_tn_bf16: 1 (default) = 256 x 256 tiles with one workgroup of eight waves per CU and a four-stage LDS ring where N and K are
 * multiples of 256 and tiles x splits fill the chip, 0 = the 128-row kernel everywhere.  Bit-identical results.  Returns the previous setting. */
int ldm_gemm_tn_ring(int v);
int ldm_cast_bf16(const float *x, void *out, long long n, void *stream);                       /* out = bf16(x), n % 4 == 0          */
int ldm_uncast_bf16(const void *x, float *out, long long n, void *stream);                      /* out = fp32(x), exact               */
int ldm_transpose_cast_bf16(const float *x, void *out, long long R, int C, void *stream);      /* out[C][R] = bf16(x[R][C]^T)        */
int ldm_gate_fwd_bf16(const void *a, const void *b, void *out, long long n, void *stream);     /* a * relu(b), all bf16, n % 8 == 0  */
int ldm_gate_bwd_bf16(const void *dh, const void *a, const void *b, void *da, void *db, long long n, void *stream);
int ldm_relu_bwd_bf16(const void *dy, const void *y, void *dx, long long n, void *stream);
/* ldm_channelnorm_film_f32 with the result written as fp32 and / or bf16 (either output may be NULL) */
int ldm_channelnorm_film_bf16(const float *x, const float *film, const int *slot, float *out_f32, void *out_bf16, int B, int HW, int C, float eps,
                              void *stream);
/* ldm_channelnorm_film_bwd_f32 for one FiLM slot per sample: dx fp32 (+ optional bf16 copy for the GEMMs that consume it),
 * dfilm (mul | bias) [B*HW, 2C] written once as bf16 */
int ldm_channelnorm_film_bwd_bf16(const float *x, const float *film, const int *slot, const float *dxf, const float *dres, float *dx, void *dx_bf16,
                                  void *dfilm_bf16, int B, int HW, int C, float eps, void *stream);
/* ldm_channelnorm_film_bf16 / ldm_channelnorm_film_bwd_bf16 with the FiLM rows themselves in bf16 ([nslot, HW, 2C] bf16, widened exactly on
 * load): the bf16 training step keeps its per-(sample, pixel) FiLM rows as the bf16 output of the proj2 GEMM */
int ldm_channelnorm_film16_bf16(const float *x, const void *film_bf16, const int *slot, float *out_f32, void *out_bf16, int B, int HW, int C, float eps,
                                void *stream);
int ldm_channelnorm_film16_bwd_bf16(const float *x, const void *film_bf16, const int *slot, const float *dxf, const float *dres, float *dx, void *dx_bf16,
                                    void *dfilm_bf16, int B, int HW, int C, float eps, void *stream);

/* Encodings.proj1 in separable form for the training step (unet.py:18-20 with one timestep per sample): the input of proj1 is
 * cat[pe(pixel), te(t_b)], so proj1(cat) = P[pixel] + T[b] with P = W1[:, :C] pe [HW, N] and T = W1[:, C:] te + b1 [B, N]
 * (two small GEMMs, N = 4C).  ldm_film_hidden: out[b, pixel, :] = relu(P[pixel] + T[b]) as fp32 or bf16 [B*HW, N].
 * ldm_film_hidden_bwd: dhm = dh * (hid > 0); partial planes dP_part[z][HW][N] (z < zchunks: sample chunks, summed by the
 * caller) and dT_part[ceil(HW/32)][B][N] (pixel tiles, summed by the caller); dh / hid fp32 or bf16 (is_bf16); N % 64 == 0.
 * ldm_film_hidden_bwd_chunks returns the zchunks that fills the chip. */
int ldm_film_hidden(const float *P, const float *T, void *out, int out_bf16, int B, int HW, int N, void *stream);
int ldm_film_hidden_bwd_chunks(int B, int HW, int N);
int ldm_film_hidden_bwd(const void *dh, const void *hid, int is_bf16, float *dP_part, float *dT_part, int B, int HW, int N, int zchunks,
                        void *stream);

/* ------------------------------------------------------------------------------------------------
 * VectorQuantizer of the VAE training path (reference vae.py:7-26).
 * ------------------------------------------------------------------------------------------------ */
/* vae.py:18-22: idx[m] = argmax_n -cdist(x, emb)[m, n], FIRST maximum (torch.argmax), with torch.cdist's matrix-multiplication
 * rounding (csrc/vq.hip).  x [M, D] fp32 rows, emb [N, D], idx [M] int64; D in {4, 8, 16}. */
int ldm_vq_quantize_f32(const float *x, const float *emb, long long *idx, long long M, int N, int D, void *stream);
int ldm_vq_embed_f32(const long long *idx, const float *emb, float *out, long long M, int D, void *stream);      /* vae.py:24-26 */
/* vae.py:12-16: loss[0] = l1(x, e) + l1(e, x) over n = M*D elements; backward: dx = g sign(x - e) / n, demb[idx] += g sign(e - x) / n
 * (demb [N, D] is zeroed by the call; gscale = 1-element device tensor with the incoming gradient) */
int ldm_vq_loss_f32(const float *x, const float *e, long long n, float *loss, void *stream);
int ldm_vq_loss_bwd_f32(const float *x, const float *e, const long long *idx, const float *gscale, float *dx, float *demb, long long M, int N, int D,
                        void *stream);

/* ---- VAE Decoder backward (vae.py:54-66,99-132), the pieces that are not GEMMs --------------------------------------
 * The dense 3x3 data gradient is ldm_gemm_f32 (LDM_A_CONV3X3) on the flipped, in/out-swapped filter, its weight gradient
 * ldm_gemm_tn_f32 of dy against ldm_im2col3x3_f32(x); the ConvTranspose 2x2 gradients are plain GEMMs on the
 * ldm_space_to_depth2_f32 image of the fine gradient. */
/* dx = dy * (y > 0 ? 1 : slope) with y the activated output of F.leaky_relu (vae.py:62,64); n % 4 == 0 */
int ldm_lrelu_bwd_f32(const float *dy, const float *y, float *dx, long long n, float slope, void *stream);
/* out[p][tap * C + c] = x[p + tap][c], zero outside the image; x [B*H*W, C] channels-last rows, out [B*H*W, 9*C] */
int ldm_im2col3x3_f32(const float *x, float *out, int B, int H, int W, int C, void *stream);
/* out[(b, y, x)][(dy*2 + dx)*C + c] = fine[(b, 2y + dy, 2x + dx)][c]; H, W = the COARSE size, fine [B*2H*2W, C], out [B*H*W, 4*C] */
int ldm_space_to_depth2_f32(const float *fine, float *out, int B, int H, int W, int C, void *stream);
/* backward of ldm_rgb_head_f32 (to_rgb 1x1 conv + bilinear x2 accumulation of the previous stage's RGB, vae.py:106-107,131):
 * drgb [B, 3, H, W] (NCHW); drows [B*H*W, C] = (accumulate ? drows : 0) + drgb . w; dprev [B, 3, H/2, W/2] (zeroed by the caller,
 * or NULL) += adjoint of the bilinear x2; dw [3, C] and db [3] (zeroed by the caller) += the to_rgb weight / bias gradient */
int ldm_rgb_head_bwd_f32(const float *drgb, const float *w, const float *rows, float *drows, int accumulate, float *dprev, float *dw,
                         float *db, int B, int H, int W, int C, void *stream);
int ldm_rgb_head_bwd_oc_f32(const float *drgb, const float *w, const float *rows, float *drows, int accumulate, float *dprev, float *dw,
                            float *db, int B, int H, int W, int C, int OC, void *stream);      /* OC output channels (1..4) */

/* Grouped 3x3 conv of unet.py:30,44 (32 in / 32 out per group, zero pad 1) with bf16 operands, fp32 accumulate:
 * out[m, :] = conv(x)[m, :] (+ bias) (+ addend[m, :]);  x [B*H*W, C] bf16, w [C][9][32] bf16 (ldm_gemm_f32's packed grouped
 * layout: output channel, tap, input channel), bias [C] / addend [B*H*W, C] fp32 or NULL, out fp32 (may alias addend).
 * The data gradient is the same call on dy with the flipped, in/out-swapped filter. */
int ldm_gconv3x3_bf16(const void *x, const void *w, const float *bias, const float *addend, float *out, int B, int H, int W, int C, void *stream);
/* kernel behind ldm_gconv3x3_bf16: 1 (default) = the LDS-tiled kernel where the shape allows (W a power of two in 8 .. 64, H a multiple of the
 * tile's rows, an even number of groups), 0 = the direct-from-global kernel everywhere.  Bit-identical results; A/B knob.  Returns the old value. */
int ldm_gconv3x3_bf16_tiled(int v);

/* Weight gradient of the same layer from bf16 x and dy [B*H*W, C]: out_planes[(s * 4 + w)][C][288] fp32 (288 = tap * 32 + ci),
 * s < splits, w < 4; the caller sums the 4 * splits planes (ldm_reduce_partials_f32).  ldm_gconv3x3_wgrad_bf16_splits suggests splits. */
int ldm_gconv3x3_wgrad_bf16_splits(int B, int H, int W, int C);
int ldm_gconv3x3_wgrad_bf16(const void *x, const void *dy, float *out_planes, int B, int H, int W, int C, int splits, void *stream);

/* All bf16 weight copies of a training step in one launch: job j = fp32 row-major [rows, cols] -> bf16 copy `dst` [rows, cols]
 * and / or transposed bf16 copy `dst_t` [cols, rows] (NULL = not wanted).  `items` is a HOST array; `table_dev` a DEVICE scratch
 * of ldm_multi_cast_table_bytes(njobs) that holds the uploaded job table between calls: pass rebuild != 0 on the first call and
 * whenever a pointer or shape changed, 0 otherwise (*tiles_io carries the grid size between calls). */
typedef struct ldm_cast_job {
    const float *src;
    void        *dst, *dst_t;
    long long    rows;
    int          cols;
} ldm_cast_job;
/* Both bf16 filter tables of the grouped 3x3 conv from conv.weight [C, 32, 3, 3] fp32 in one launch: fwd [C][tap][ci] (the forward's
 * operand) and rot [g*32 + ci][mirrored tap][co] (the data gradient's: the flipped, in/out-swapped filter). */
int ldm_gconv_pack_bf16(const float *w, void *fwd_bf16, void *rot_bf16, int C, void *stream);
/* out[r][0..n) = src[0..n) for r < reps (one gradient shared by several biases, as separate rows) */
int ldm_replicate_f32(const float *src, float *out, int n, int reps, void *stream);
/* dense 3x3 conv.weight [Cout, Cin, 3, 3] -> the forward matrix fwd [Cout][tap * Cin + ci] (LDM_A_CONV3X3's column order) and the
 * data-gradient matrix dgrad [Cin][tap' * Cout + co] with tap' = 8 - tap (the mirrored filter, in / out swapped: vae.py:57-58's autograd)
 * in one launch; either output may be NULL */
int ldm_pack3x3_f32(const float *w, float *fwd, float *dgrad, int Cout, int Cin, void *stream);
size_t ldm_multi_cast_table_bytes(int njobs);
int ldm_multi_cast_bf16(const ldm_cast_job *items, int njobs, void *table_dev, int rebuild, long long *tiles_io, void *stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 sampling / decode ("autocast": ddpm.py:52,75 -- on a GPU the reference runs DDPM.sample under 16-bit autocast,
 * sample_ldm.py:17,72; fp16 overflows on these weights, so the 16-bit type here is bf16).  Opt-in; the default path is
 * exact fp32.  GEMM operands are bf16 with fp32 accumulation; the UNet's residual stream, FiLM tables, attention core
 * and the DDIM update stay fp32; the VAE decoder keeps its activations as bf16 rows and emits fp32 RGB planes.
 * ------------------------------------------------------------------------------------------------ */
/* ldm_window_attention_f32 behind a bf16 in-projection: qkv [B,H,W,3C] is fp32 (qkv_is_bf16 == 0) or bf16 (!= 0: widened exactly
 * on load, zero-padded tokens take the bias rounded to bf16); the float "mask" source of shifted windows is the bf16 normalised
 * input xf_bf16 [B,H,W,C] (NULL allowed when shift == 0); the context leaves as bf16 rows [B,H,W,C].  The attention arithmetic
 * (scores, softmax, P.V) is fp32.  ws*ws <= 48. */
int ldm_window_attention_bf16io(const void *qkv, int qkv_is_bf16, const float *in_proj_bias, const void *xf_bf16, void *out_bf16,
                                int B, int H, int W, int C, int ws, int shift, void *stream);
/* core of ldm_window_attention_bf16io when qkv is bf16: 1 (default) = both products on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16 for
 * K Q^T over the 32 head dims, v_mfma_f32_16x16x16_bf16 for P V with P rounded once to bf16; scores, softmax in fp32), 0 = the fp32
 * 16x16x4 core on the widened values (A/B tests).  Returns the previous setting; any other v only queries. */
int ldm_window_attention_bf16_core(int v);
/* ldm_stem_nchw_f32 with bf16 rows out (VAE Decoder.input_layer, vae.py:112,123) */
int ldm_stem_nchw_bf16(const float *x, const float *w, const float *bias, void *out_bf16, int B, int Cin, int HW, int C0, void *stream);
/* the scatter of ConvTranspose2d(k=2, s=2) (vae.py:120) as its own pass: in [B*H*W, 4*C] bf16 with columns (dy, dx, c)
 * -> out [B*2H*2W, C] bf16 */
int ldm_depth_to_space2_bf16(const void *in_bf16, void *out_bf16, int B, int H, int W, int C, void *stream);
/* ldm_rgb_head_f32 on bf16 rows (vae.py:104,131): out / prev stay fp32 NCHW planes */
int ldm_rgb_head_bf16(const void *x_bf16, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, void *stream);
int ldm_rgb_head_oc_bf16(const void *x_bf16, const float *w, const float *bias, const float *prev, float *out, int B, int H, int W, int C, int OC,
                         void *stream);                                                        /* OC output channels (1..4) */
/* out[(b, y, x), :] = coarse[(b, y/2, x/2), :] (+ skip[(b, y, x), :]) on fp32 rows; H, W are the COARSE sizes (unet.py:85,101 split
 * from its GEMM) */
int ldm_up2_add_f32(const float *coarse, const float *skip, float *out, int B, int H, int W, int C, void *stream);
/* ldm_avgpool2_f32 rounding its result once to bf16 rows (the operand of the 1x1 down conv, unet.py:83) */
int ldm_avgpool2_bf16(const float *x, void *out_bf16, int B, int H, int W, int C, void *stream);

/* bf16 copies of the GEMM weights of a ldm_unet_plan, block by block in the plan's order: grouped conv packed [C][tap][32],
 * ReGLU a / b / c [C, C], in_proj [3C, C], out_proj [C, C] (NULL where the block has no attention). */
typedef struct ldm_unet_block_bf16 {
    const void *conv_w;
    const void *a_w[5], *b_w[5], *c_w[5];
    const void *in_w, *out_w;
} ldm_unet_block_bf16;
typedef struct ldm_unet_plan_bf16 {
    int nblocks;
    const ldm_unet_block_bf16 *blocks;       /* HOST array, execution order */
} ldm_unet_plan_bf16;
/* ldm_unet_forward_ex_f32 in the bf16 mode (same workspace size, same arguments); every stage width must be a multiple of 64 */
/* streams of the native executor: 1 (default) = everything on the caller's stream; 2 = the gated GEMM of every SwinBlock on a library-owned
 * side stream beside the grouped conv / attention branch (event fork / join on the caller's stream; bit-identical results).  LDM_UNET_STREAMS
 * in the environment sets the initial value.  Returns the previous setting; any other v only queries. */
int ldm_unet_streams(int v);
int ldm_unet_forward_bf16(const ldm_unet_plan *plan, const ldm_unet_plan_bf16 *plan16, const float *x, const long long *t_unique, int nT,
                          const int *slot, const int *decisions, int B, int H, int W, void *workspace, size_t workspace_bytes, float *out,
                          int films_ready, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LDM_HIP_H */
