"""Thin tensor-level wrappers over the C ABI (include/ldm_hip.h).

PyTorch is plumbing here: it owns device memory and the stream.  Every function
hands raw ``data_ptr()`` addresses and the current HIP stream to the library and
returns immediately (kernels are only enqueued).  Nothing in this file computes
on the CPU and nothing falls back to ATen ops.
"""
import ctypes

import torch

from . import _lib
from ._lib import (A_CONV3X3, A_ROWS, ACT_GATE, ACT_LRELU, ACT_NONE, ACT_RELU, O_CONVT2X2, O_ROWS, O_UP2, SEG_K, SEG_N,
                   GemmDesc)

__all__ = ["gemm", "pointer_table", "channelnorm_film", "film", "sincos_embed", "window_attention", "avgpool2", "stem_nchw", "head_nchw",
           "ddim_update", "qsample", "rgb_head", "nchw_to_nhwc", "nhwc_to_nchw", "to_uint8_hwc", "prof_enable", "prof_read", "prof_read_class", "prof_read_bytes", "gate_fwd", "gate_bwd", "relu_bwd", "add_", "colsum", "transpose_colsum", "reduce_partials", "channelnorm_film_bwd",
           "avgpool2_bwd", "sumpool2", "stem_bwd", "head_bwd", "l1_loss", "l1_loss_bwd", "im2col3x3_t", "window_attention_bwd", "window_attention_bwd_mfma", "reduce_partials_pair", "gemm_variant", "gemm_wide_epilogue", "gemm_ring", "lrelu_bwd", "im2col3x3", "space_to_depth2", "rgb_head_bwd", "gemm_tn", "gconv3x3_wgrad",
           "ACT_NONE", "ACT_RELU", "ACT_GATE", "ACT_LRELU", "A_ROWS", "A_CONV3X3", "O_ROWS", "O_CONVT2X2", "O_UP2",
           "SEG_N", "SEG_K"]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.LdmHipUnavailable(
            "%s must be a GPU tensor: the HIP path is the only implementation (no CPU fallback)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def _opt(t, name, dtype=torch.float32):
    return None if t is None else _dev(t, name, dtype)


_SCRATCH = {}


def _scratch(device, nbytes=32 << 20):
    """Per-(device, stream) split-K scratch for small-M GEMMs (ldm_gemm_desc.workspace)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None:
        buf = _SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


def gemm(a, M, N, K, weights, out, *, lda=None, ldo=None, weights2=None, biases=None, biases2=None, ldw=None,
         seg_mode=SEG_N, act=ACT_NONE, slope=0.0, addend=None, ldadd=None, a_mode=A_ROWS, conv_hw=None, cin=0,
         o_mode=O_ROWS, out_hw=None, cout=0, groups=1, a_gstride=0, w_gstride=0, o_gstride=0, b_gstride=0,
         w_table=None, bias_table=None):
    """out = act(A . W^T + bias) (+ addend); see struct ldm_gemm_desc."""
    lib = _lib.load()
    d = GemmDesc()
    d.a = _dev(a, "a")
    d.lda = K if lda is None else lda
    d.M, d.N, d.K = M, N, K
    d.a_mode = a_mode
    if conv_hw is not None:
        d.H, d.W = conv_hw
    d.Cin = cin
    nseg = 1 if w_table is not None else len(weights)
    d.nseg = nseg
    d.seg_mode = seg_mode
    d.seg_len = (N if seg_mode == SEG_N else K) // nseg
    if w_table is not None:                 # host ctypes arrays of device addresses (see pointer_table)
        d.w_table = ctypes.cast(w_table, ctypes.c_void_p)
        d.bias_table = None if bias_table is None else ctypes.cast(bias_table, ctypes.c_void_p)
    for s in range(0 if w_table is not None else nseg):
        d.w[s] = _dev(weights[s], "weight")
        if weights2 is not None:
            d.w2[s] = _dev(weights2[s], "weight2")
        if biases is not None and biases[s] is not None:
            d.bias[s] = _dev(biases[s], "bias")
        if biases2 is not None and biases2[s] is not None:
            d.bias2[s] = _dev(biases2[s], "bias2")
    if ldw is None:
        ldw = K if seg_mode == SEG_N else K // nseg
    d.ldw = ldw
    d.act = act
    d.slope = slope
    d.addend = _opt(addend, "addend")
    d.ldadd = (N if ldadd is None else ldadd)
    d.out = _dev(out, "out")
    d.ldo = N if ldo is None else ldo
    d.o_mode = o_mode
    if out_hw is not None:
        d.OH, d.OW = out_hw
    d.Cout = cout
    d.groups = groups
    d.a_gstride, d.w_gstride, d.o_gstride, d.b_gstride = a_gstride, w_gstride, o_gstride, b_gstride
    if M <= 128 and K >= 256 and groups == 1:
        ws = _scratch(a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.ldm_gemm_f32(ctypes.byref(d), _stream()), "ldm_gemm_f32")
    return out


def gemm_gate_fwd(a, M, N, K, weights_a, weights_b, hid, a_pre, b_pre, *, biases_a=None, biases_b=None):
    """ReGLU forward of the fp32 training step: hid = (A Wa^T + ba) * relu(A Wb^T + bb) plus both pre-activations, all fp32 [M, N], in ONE
    launch where the ring kernel takes the shape; otherwise the three launches it is bit-identical to (two plain GEMMs, gate_fwd)."""
    lib = _lib.load()
    d = GemmDesc()
    d.a, d.lda = _dev(a, "a"), K
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.o_mode = A_ROWS, O_ROWS
    nseg = len(weights_a)
    d.nseg, d.seg_mode, d.seg_len = nseg, SEG_N, N // nseg
    for s in range(nseg):
        d.w[s], d.w2[s] = _dev(weights_a[s], "weight a"), _dev(weights_b[s], "weight b")
        if biases_a is not None and biases_a[s] is not None:
            d.bias[s] = _dev(biases_a[s], "bias a")
        if biases_b is not None and biases_b[s] is not None:
            d.bias2[s] = _dev(biases_b[s], "bias b")
    d.ldw, d.act = K, ACT_GATE
    d.out, d.ldo, d.ldadd, d.groups = _dev(hid, "hid"), N, N, 1
    rc = lib.ldm_gemm_f32_gate_fwd(ctypes.byref(d), _dev(a_pre, "a_pre"), _dev(b_pre, "b_pre"), _stream())
    if rc == 1:                               # no single-launch instance for this shape
        gemm(a, M, N, K, weights_a, a_pre, biases=biases_a)
        gemm(a, M, N, K, weights_b, b_pre, biases=biases_b)
        return gate_fwd(a_pre, b_pre, hid)
    _lib.check(rc, "ldm_gemm_f32_gate_fwd")
    return hid


def gemm_gate_bwd(dy, M, N, K, weights_t, a_pre, b_pre, da, db):
    """ReGLU backward of the fp32 training step: dh = dy . Wc (weights_t: the transposed c-weights as N-segments [N / nseg, K]) with
    da = dh * relu(b_pre), db = dh * a_pre * (b_pre > 0) formed in the GEMM's epilogue (dh never reaches HBM) where the ring kernel takes
    the shape; otherwise the two launches it is bit-identical to (plain GEMM into a temporary, gate_bwd)."""
    lib = _lib.load()
    d = GemmDesc()
    d.a, d.lda = _dev(dy, "dy"), K
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.o_mode = A_ROWS, O_ROWS
    nseg = len(weights_t)
    d.nseg, d.seg_mode, d.seg_len = nseg, SEG_N, N // nseg
    for s in range(nseg):
        d.w[s] = _dev(weights_t[s], "weight")
    d.ldw, d.act = K, ACT_NONE
    d.out, d.ldo, d.ldadd, d.groups = _dev(da, "da"), N, N, 1
    rc = lib.ldm_gemm_f32_gate_bwd(ctypes.byref(d), _dev(a_pre, "a_pre"), _dev(b_pre, "b_pre"), _dev(db, "db"), _stream())
    if rc == 1:
        dh = torch.empty(M, N, device=dy.device, dtype=torch.float32)
        gemm(dy, M, N, K, weights_t, dh)
        gate_bwd(dh, a_pre, b_pre, da, db)
        return da, db
    _lib.check(rc, "ldm_gemm_f32_gate_bwd")
    return da, db


def pointer_table(tensors):
    """Host array of device addresses for gemm(w_table=/bias_table=); keep it alive during the call."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = _dev(t, "table entry")
    return arr


def channelnorm_film(x, film, slot, out, B, HW, C, eps=1e-4):
    lib = _lib.load()
    _lib.check(lib.ldm_channelnorm_film_f32(_dev(x, "x"), _dev(film, "film"), _opt(slot, "slot", torch.int32),
                                            _dev(out, "out"), B, HW, C, eps, _stream()), "ldm_channelnorm_film_f32")
    return out


def film(x, film_rows, slot, out, B, HW, C):
    lib = _lib.load()
    _lib.check(lib.ldm_film_f32(_dev(x, "x"), _dev(film_rows, "film"), _opt(slot, "slot", torch.int32), _dev(out, "out"),
                                B, HW, C, _stream()), "ldm_film_f32")
    return out


def sincos_embed(t, H, W, C, pos_freq, time_freq, out):
    lib = _lib.load()
    _lib.check(lib.ldm_sincos_embed_f32(_dev(t, "t", torch.int64), t.numel(), H, W, C, _dev(pos_freq, "pos_freq"),
                                        _dev(time_freq, "time_freq"), _dev(out, "emb"), _stream()), "ldm_sincos_embed_f32")
    return out


def window_attention(qkv, in_proj_bias, xf, out, B, H, W, C, ws, shift):
    lib = _lib.load()
    _lib.check(lib.ldm_window_attention_f32(_dev(qkv, "qkv"), _dev(in_proj_bias, "in_proj_bias"), _opt(xf, "xf"),
                                            _dev(out, "out"), B, H, W, C, ws, shift, _stream()), "ldm_window_attention_f32")
    return out


def avgpool2(x, out, B, H, W, C):
    lib = _lib.load()
    _lib.check(lib.ldm_avgpool2_f32(_dev(x, "x"), _dev(out, "out"), B, H, W, C, _stream()), "ldm_avgpool2_f32")
    return out


def stem_nchw(x, w, bias, out, B, Cin, HW, C0):
    lib = _lib.load()
    _lib.check(lib.ldm_stem_nchw_f32(_dev(x, "x"), _dev(w, "w"), _opt(bias, "bias"), _dev(out, "out"), B, Cin, HW, C0,
                                     _stream()), "ldm_stem_nchw_f32")
    return out


def head_nchw(x, w, bias, out, B, C0, HW, Cin):
    lib = _lib.load()
    _lib.check(lib.ldm_head_nchw_f32(_dev(x, "x"), _dev(w, "w"), _opt(bias, "bias"), _dev(out, "out"), B, C0, HW, Cin,
                                     _stream()), "ldm_head_nchw_f32")
    return out


def ddim_update(x, e_theta, noise, s1, s2, s3, s4, sigma, last):
    lib = _lib.load()
    _lib.check(lib.ldm_ddim_update_f32(_dev(x, "x"), _dev(e_theta, "e_theta"), _opt(noise, "noise"), x.numel(),
                                       s1, s2, s3, s4, sigma, int(last), _stream()), "ldm_ddim_update_f32")
    return x


def qsample(x, e, sa, sb, out):
    lib = _lib.load()
    B = x.shape[0]
    _lib.check(lib.ldm_qsample_f32(_dev(x, "x"), _dev(e, "e"), _dev(sa, "sa"), _dev(sb, "sb"), _dev(out, "out"), B,
                                   x.numel() // B, _stream()), "ldm_qsample_f32")
    return out


def rgb_head(x, w, bias, prev, out, B, H, W, C):
    """to_rgb [OC, C] + bilinear x2 accumulation of ``prev`` [B, OC, H/2, W/2] into ``out`` [B, OC, H, W]; OC = w.shape[0] (1..4)."""
    lib = _lib.load()
    _lib.check(lib.ldm_rgb_head_oc_f32(_dev(x, "x"), _dev(w, "w"), _dev(bias, "bias"), _opt(prev, "prev"), _dev(out, "out"),
                                       B, H, W, C, w.shape[0], _stream()), "ldm_rgb_head_oc_f32")
    return out


def nchw_to_nhwc(x, out, B, C, HW):
    lib = _lib.load()
    _lib.check(lib.ldm_nchw_to_nhwc_f32(_dev(x, "x"), _dev(out, "out"), B, C, HW, _stream()), "ldm_nchw_to_nhwc_f32")
    return out


def nhwc_to_nchw(x, out, B, C, HW):
    lib = _lib.load()
    _lib.check(lib.ldm_nhwc_to_nchw_f32(_dev(x, "x"), _dev(out, "out"), B, C, HW, _stream()), "ldm_nhwc_to_nchw_f32")
    return out


def to_uint8_hwc(img, out, B, C, HW):
    lib = _lib.load()
    _lib.check(lib.ldm_to_uint8_hwc(_dev(img, "img"), _dev(out, "out", torch.uint8), B, C, HW, _stream()), "ldm_to_uint8_hwc")
    return out


def gemm_wide_epilogue(v):
    """1 (default): stream-schedule epilogue through LDS with 16-byte stores; 0: direct 4-byte stores.  Returns the old value."""
    return _lib.load().ldm_gemm_wide_epilogue(v)


def gemm_ring(v):
    """Schedule of gemm / gemm_bf16 for 256-row-aligned rows problems: 1 auto (default), 0 stream kernel only, 2 ring kernel whenever legal.  Returns the old value."""
    return _lib.load().ldm_gemm_ring(v)


def window_attention_bwd_mfma(v):
    """Kernel behind window_attention_bwd: 1 (default) MFMA products, 0 the scalar kernel.  Returns the old value."""
    return _lib.load().ldm_window_attention_bwd_mfma(v)


def gemm_variant(v):
    """0 = tile-per-block schedule, 1 = persistent LDS-DMA stream; returns the previous value."""
    return _lib.load().ldm_gemm_variant(v)


def prof_enable(on):
    _lib.check(_lib.load().ldm_prof_enable(1 if on else 0), "ldm_prof_enable")


def prof_read():
    n, ms, fl = ctypes.c_longlong(0), ctypes.c_double(0), ctypes.c_double(0)
    _lib.check(_lib.load().ldm_prof_read(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), "ldm_prof_read")
    return n.value, ms.value, fl.value


def prof_read_class(cls):
    """(launches, kernel ms, algorithmic FLOPs) of one kernel class (see ldm_prof_read_class); does not clear."""
    n, ms, fl = ctypes.c_longlong(0), ctypes.c_double(0), ctypes.c_double(0)
    _lib.check(_lib.load().ldm_prof_read_class(cls, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), "ldm_prof_read_class")
    return n.value, ms.value, fl.value


def prof_read_bytes(cls=-1):
    b = ctypes.c_double(0)
    _lib.check(_lib.load().ldm_prof_read_bytes(cls, ctypes.byref(b)), "ldm_prof_read_bytes")
    return b.value


# ---- training-step entry points (include/ldm_hip.h, "Training step") ---------------------------------
def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, _stream()), name)


def gate_fwd(a, b, out):
    _call("ldm_gate_fwd_f32", _dev(a, "a"), _dev(b, "b"), _dev(out, "out"), a.numel())
    return out


def gate_bwd(dh, a, b, da, db):
    _call("ldm_gate_bwd_f32", _dev(dh, "dh"), _dev(a, "a"), _dev(b, "b"), _dev(da, "da"), _dev(db, "db"), a.numel())


def relu_bwd(dy, y, dx):
    _call("ldm_relu_bwd_f32", _dev(dy, "dy"), _dev(y, "y"), _dev(dx, "dx"), y.numel())
    return dx


def add_(y, x):
    _call("ldm_add_f32", _dev(y, "y"), _dev(x, "x"), y.numel())
    return y


def colsum(x, M, N, out=None, accumulate=False):
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    _call("ldm_colsum_f32", _dev(x, "x"), _dev(out, "out"), M, N, int(accumulate))
    return out


def transpose_colsum(x, out, csum):
    r, c = x.shape
    _call("ldm_transpose_colsum_f32", _dev(x, "x"), _dev(out, "out"), _dev(csum, "csum"), r, c)


def gemm_tn(a, b, out, M, N, K, splits=1, lda=None, ldb=None, colsum=None):
    """out[s] [N, K] = sum over the rows of split s of a[m, :N]^T b[m, :K] (weight gradients, no transposed copies);
    ``colsum`` [splits, N] optionally receives the per-split column sums of ``a`` (the bias gradient)."""
    _call("ldm_gemm_tn_f32", _dev(a, "a"), N if lda is None else lda, _dev(b, "b"), K if ldb is None else ldb, _dev(out, "out"),
          _opt(colsum, "colsum"), M, N, K, splits)
    return out


def gconv3x3_wgrad(x, dy, planes, B, H, W, C, splits):
    """planes [4 * splits, C, 288]: partial weight gradients of the grouped 3x3 conv (sum them with reduce_partials)."""
    _call("ldm_gconv3x3_wgrad_f32", _dev(x, "x"), _dev(dy, "dy"), _dev(planes, "planes"), B, H, W, C, splits)
    return planes


def reduce_partials(parts, S, n, out):
    _call("ldm_reduce_partials_f32", _dev(parts, "parts"), _dev(out, "out"), S, n)
    return out


def reduce_partials_pair(parts_a, n_a, out_a, parts_b, n_b, out_b, S, row_len_a=0, seg_len_a=0):
    """reduce_partials twice (same S) in one launch; falls back to two launches when a count is not a multiple of 4.
    ``seg_len_a`` > 0: sum a, a [n_a / row_len_a, row_len_a] matrix, is stored as [row_len_a / seg_len_a][rows][seg_len_a]
    (contiguous column blocks: gradients of several parameters out of one weight-gradient GEMM)."""
    if (n_a % 4 or n_b % 4) and not seg_len_a:
        reduce_partials(parts_a, S, n_a, out_a)
        reduce_partials(parts_b, S, n_b, out_b)
    else:
        _call("ldm_reduce_partials_pair_f32", _dev(parts_a, "parts_a"), _dev(out_a, "out_a"), n_a, _dev(parts_b, "parts_b"), _dev(out_b, "out_b"), n_b, S,
              row_len_a, seg_len_a)
    return out_a, out_b


def channelnorm_film_bwd(x, film, slot, dxf, dres, dx, dfilm, B, HW, C, eps=1e-4, unique_slots=False):
    _call("ldm_channelnorm_film_bwd_f32", _dev(x, "x"), _dev(film, "film"), _opt(slot, "slot", torch.int32), _dev(dxf, "dxf"),
          _opt(dres, "dres"), _dev(dx, "dx"), _dev(dfilm, "dfilm"), B, HW, C, eps, int(unique_slots))
    return dx


def avgpool2_bwd(dlo, dx, B, H, W, C, accumulate):
    _call("ldm_avgpool2_bwd_f32", _dev(dlo, "dlo"), _dev(dx, "dx"), B, H, W, C, int(accumulate))
    return dx


def sumpool2(dhi, dlo, B, H, W, C):
    _call("ldm_sumpool2_f32", _dev(dhi, "dhi"), _dev(dlo, "dlo"), B, H, W, C)
    return dlo


def stem_bwd(x, dy, dw, B, Cin, HW, C0):
    _call("ldm_stem_bwd_f32", _dev(x, "x"), _dev(dy, "dy"), _dev(dw, "dw"), B, Cin, HW, C0)
    return dw


def head_bwd(x, w, dout, dx, dw, db, B, C0, HW, Cin):
    _call("ldm_head_bwd_f32", _dev(x, "x"), _dev(w, "w"), _dev(dout, "dout"), _dev(dx, "dx"), _dev(dw, "dw"), _dev(db, "db"), B, C0, HW, Cin)


def l1_loss(pred, target, loss):
    _call("ldm_l1_loss_f32", _dev(pred, "pred"), _dev(target, "target"), pred.numel(), _dev(loss, "loss"))
    return loss


def l1_loss_bwd(pred, target, gscale, grad):
    _call("ldm_l1_loss_bwd_f32", _dev(pred, "pred"), _dev(target, "target"), _dev(gscale, "gscale"), _dev(grad, "grad"), pred.numel())
    return grad


def im2col3x3_t(x, out, B, H, W, C):
    _call("ldm_im2col3x3_t_f32", _dev(x, "x"), _dev(out, "out"), B, H, W, C)
    return out


def window_attention_bwd(qkv, in_proj_bias, xf, dctx, dqkv, dbias_pad, B, H, W, C, ws, shift):
    _call("ldm_window_attention_bwd_f32", _dev(qkv, "qkv"), _dev(in_proj_bias, "bias"), _opt(xf, "xf"), _dev(dctx, "dctx"),
          _dev(dqkv, "dqkv"), _dev(dbias_pad, "dbias_pad"), B, H, W, C, ws, shift)


# ---- bf16 training step (include/ldm_hip.h, "bf16 training step") --------------------------------------
BF16 = torch.bfloat16


def gemm_bf16(a, M, N, K, weights, out, *, lda=None, ldo=None, biases=None, ldw=None, seg_mode=SEG_N, act=ACT_NONE, slope=0.0,
              addend=None, ldadd=None, a_mode=A_ROWS, conv_hw=None, cin=0):
    """out = act(A . W^T + bias) (+ addend) with bf16 A and W (fp32 accumulate); ``out`` is fp32 or bf16 by its dtype and an
    addend has the output's dtype.  ``a_mode=A_CONV3X3`` (bf16 out): dense 3x3 over bf16 rows [M, cin], K = 9 * cin."""
    lib = _lib.load()
    d = GemmDesc()
    d.a = _dev(a, "a", BF16)
    d.lda = (cin if a_mode == A_CONV3X3 else K) if lda is None else lda
    d.M, d.N, d.K = M, N, K
    d.a_mode = a_mode
    if conv_hw is not None:
        d.H, d.W = conv_hw
    d.Cin = cin
    nseg = len(weights)
    d.nseg, d.seg_mode = nseg, seg_mode
    d.seg_len = (N if seg_mode == SEG_N else K) // nseg
    for s in range(nseg):
        d.w[s] = _dev(weights[s], "weight", BF16)
        if biases is not None and biases[s] is not None:
            d.bias[s] = _dev(biases[s], "bias")
    d.ldw = (K if seg_mode == SEG_N else K // nseg) if ldw is None else ldw
    d.act, d.slope = act, slope
    out_bf16 = out.dtype == BF16
    d.addend = _opt(addend, "addend", BF16 if out_bf16 else torch.float32)
    d.ldadd = N if ldadd is None else ldadd
    d.out = _dev(out, "out", BF16 if out_bf16 else torch.float32)
    d.ldo = N if ldo is None else ldo
    d.o_mode = O_ROWS
    d.groups = 1
    _lib.check(lib.ldm_gemm_bf16(ctypes.byref(d), int(out_bf16), _stream()), "ldm_gemm_bf16")
    return out


def _desc_bf16(a, M, N, K, weights, out, biases, seg_mode, act=ACT_NONE, weights2=None, biases2=None):
    d = GemmDesc()
    d.a = _dev(a, "a", BF16)
    d.lda = K
    d.M, d.N, d.K = M, N, K
    d.a_mode = A_ROWS
    nseg = len(weights)
    d.nseg, d.seg_mode = nseg, seg_mode
    d.seg_len = (N if seg_mode == SEG_N else K) // nseg
    for s in range(nseg):
        d.w[s] = _dev(weights[s], "weight", BF16)
        if biases is not None and biases[s] is not None:
            d.bias[s] = _dev(biases[s], "bias")
        if weights2 is not None:
            d.w2[s] = _dev(weights2[s], "weight2", BF16)
            if biases2 is not None and biases2[s] is not None:
                d.bias2[s] = _dev(biases2[s], "bias2")
    d.ldw = K if seg_mode == SEG_N else K // nseg
    d.act = act
    d.out = _dev(out, "out", BF16)
    d.ldo = d.ldadd = N
    d.o_mode = O_ROWS
    d.groups = 1
    return d


def gemm_bf16_gate_fwd(a, M, N, K, weights_a, weights_b, hid, *, biases_a=None, biases_b=None, a_pre=None, b_pre=None):
    """ReGLU forward in one launch: hid = (A Wa^T + ba) * relu(A Wb^T + bb), all bf16; a_pre / b_pre (optional, both) receive the
    pre-activations.  Weight segments along N select experts by pointer."""
    d = _desc_bf16(a, M, N, K, weights_a, hid, biases_a, SEG_N, ACT_GATE, weights_b, biases_b)
    _lib.check(_lib.load().ldm_gemm_bf16_gate_fwd(ctypes.byref(d), _opt(a_pre, "a_pre", BF16), _opt(b_pre, "b_pre", BF16), _stream()),
               "ldm_gemm_bf16_gate_fwd")
    return hid


def gemm_bf16_gate_bwd(dy, M, N, K, weights_t, a_pre, b_pre, da, db):
    """dh = dy . Wc (weights_t: the transposed c-weights, N-segments) with the gate's backward in the epilogue:
    da = dh * relu(b), db = dh * a * (b > 0); everything bf16 [M, N]."""
    d = _desc_bf16(dy, M, N, K, weights_t, da, None, SEG_N)
    _lib.check(_lib.load().ldm_gemm_bf16_gate_bwd(ctypes.byref(d), _dev(a_pre, "a_pre", BF16), _dev(b_pre, "b_pre", BF16), _dev(db, "db", BF16),
                                                  _stream()), "ldm_gemm_bf16_gate_bwd")
    return da, db


def gemm_tn_bf16(a, b, out, M, N, K, splits=1, lda=None, ldb=None, colsum=None):
    """out[s] [N, K] fp32 = sum over the rows of split s of a[m, :N]^T b[m, :K], bf16 operands as they lie in memory."""
    _call("ldm_gemm_tn_bf16", _dev(a, "a", BF16), N if lda is None else lda, _dev(b, "b", BF16), K if ldb is None else ldb, _dev(out, "out"),
          _opt(colsum, "colsum"), M, N, K, splits)
    return out


def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=BF16)
    _call("ldm_cast_bf16", _dev(x, "x"), _dev(out, "out", BF16), x.numel())
    return out


def transpose_cast_bf16(x, out=None):
    r, c = x.shape
    if out is None:
        out = torch.empty(c, r, device=x.device, dtype=BF16)
    _call("ldm_transpose_cast_bf16", _dev(x, "x"), _dev(out, "out", BF16), r, c)
    return out


def gate_fwd_bf16(a, b, out):
    _call("ldm_gate_fwd_bf16", _dev(a, "a", BF16), _dev(b, "b", BF16), _dev(out, "out", BF16), a.numel())
    return out


def gate_bwd_bf16(dh, a, b, da, db):
    _call("ldm_gate_bwd_bf16", _dev(dh, "dh", BF16), _dev(a, "a", BF16), _dev(b, "b", BF16), _dev(da, "da", BF16), _dev(db, "db", BF16), a.numel())


def relu_bwd_bf16(dy, y, dx):
    _call("ldm_relu_bwd_bf16", _dev(dy, "dy", BF16), _dev(y, "y", BF16), _dev(dx, "dx", BF16), y.numel())
    return dx


def channelnorm_film_bf16(x, film, slot, out_f32, out_bf16, B, HW, C, eps=1e-4):
    """ChannelNorm + FiLM with bf16 (and / or fp32) output; ``film`` rows fp32 or bf16 by their dtype."""
    f16 = film.dtype == BF16
    _call("ldm_channelnorm_film16_bf16" if f16 else "ldm_channelnorm_film_bf16", _dev(x, "x"), _dev(film, "film", BF16 if f16 else torch.float32),
          _opt(slot, "slot", torch.int32), _opt(out_f32, "out_f32"), _opt(out_bf16, "out_bf16", BF16), B, HW, C, eps)


def channelnorm_film_bwd_bf16(x, film, slot, dxf, dres, dx, dx_bf16, dfilm_bf16, B, HW, C, eps=1e-4):
    f16 = film.dtype == BF16
    _call("ldm_channelnorm_film16_bwd_bf16" if f16 else "ldm_channelnorm_film_bwd_bf16", _dev(x, "x"), _dev(film, "film", BF16 if f16 else torch.float32),
          _opt(slot, "slot", torch.int32), _dev(dxf, "dxf"), _opt(dres, "dres"),
          _dev(dx, "dx"), _opt(dx_bf16, "dx_bf16", BF16), _dev(dfilm_bf16, "dfilm_bf16", BF16), B, HW, C, eps)
    return dx


def uncast_bf16(x, out):
    _call("ldm_uncast_bf16", _dev(x, "x", BF16), _dev(out, "out"), x.numel())
    return out


def vq_quantize(x_rows, emb):
    """vae.py:18-22 on rows: x [M, D], emb [N, D] -> int64 indices [M] (first nearest codebook row, torch.cdist rounding)."""
    m, d = x_rows.shape
    idx = torch.empty(m, dtype=torch.int64, device=x_rows.device)
    _call("ldm_vq_quantize_f32", _dev(x_rows, "x"), _dev(emb, "emb"), _dev(idx, "idx", torch.int64), m, emb.shape[0], d)
    return idx


def vq_embed(idx, emb):
    out = torch.empty(idx.numel(), emb.shape[1], device=emb.device, dtype=torch.float32)
    _call("ldm_vq_embed_f32", _dev(idx, "idx", torch.int64), _dev(emb, "emb"), _dev(out, "out"), idx.numel(), emb.shape[1])
    return out


def vq_loss(x_rows, e_rows):
    loss = torch.empty(1, device=x_rows.device, dtype=torch.float32)
    _call("ldm_vq_loss_f32", _dev(x_rows, "x"), _dev(e_rows, "e"), x_rows.numel(), _dev(loss, "loss"))
    return loss


def lrelu_bwd(dy, y, slope=0.01):
    """dy * (y > 0 ? 1 : slope), y = the activated output."""
    dx = torch.empty_like(dy)
    _call("ldm_lrelu_bwd_f32", _dev(dy, "dy"), _dev(y, "y"), _dev(dx, "dx"), dy.numel(), float(slope))
    return dx


def im2col3x3(rows, B, H, W, C):
    out = torch.empty(B * H * W, 9 * C, device=rows.device, dtype=torch.float32)
    _call("ldm_im2col3x3_f32", _dev(rows, "rows"), _dev(out, "out"), B, H, W, C)
    return out


def space_to_depth2(fine, B, H, W, C):
    """fine [B*2H*2W, C] -> [B*H*W, 4C] ((dy, dx, c) minor)."""
    out = torch.empty(B * H * W, 4 * C, device=fine.device, dtype=torch.float32)
    _call("ldm_space_to_depth2_f32", _dev(fine, "fine"), _dev(out, "out"), B, H, W, C)
    return out


def rgb_head_bwd(drgb, w, rows, drows, accumulate, dprev, dw, db, B, H, W, C):
    _call("ldm_rgb_head_bwd_oc_f32", _dev(drgb, "drgb"), _dev(w, "w"), _dev(rows, "rows"), _dev(drows, "drows"), int(bool(accumulate)),
          _opt(dprev, "dprev"), _dev(dw, "dw"), _dev(db, "db"), B, H, W, C, w.shape[0])


def vq_loss_bwd(x_rows, e_rows, idx, gscale, n_emb):
    m, d = x_rows.shape
    dx = torch.empty_like(x_rows)
    demb = torch.empty(n_emb, d, device=x_rows.device, dtype=torch.float32)
    _call("ldm_vq_loss_bwd_f32", _dev(x_rows, "x"), _dev(e_rows, "e"), _dev(idx, "idx", torch.int64), _dev(gscale, "gscale"), _dev(dx, "dx"),
          _dev(demb, "demb"), m, n_emb, d)
    return dx, demb


def gconv3x3_bf16(x16, w16, bias, addend, out, B, H, W, C):
    """Grouped 3x3 conv (32 per group) with bf16 operands: out = conv(x16) (+ bias) (+ addend), fp32 out (may alias addend)."""
    _call("ldm_gconv3x3_bf16", _dev(x16, "x", BF16), _dev(w16, "w", BF16), _opt(bias, "bias"), _opt(addend, "addend"), _dev(out, "out"), B, H, W, C)
    return out


def gconv3x3_wgrad_bf16(x16, dy16, B, H, W, C):
    """dW [C, 288] fp32 (row = output channel, 288 = tap * 32 + input channel) of the grouped 3x3 conv from bf16 x and dy rows."""
    lib = _lib.load()
    sp = lib.ldm_gconv3x3_wgrad_bf16_splits(B, H, W, C)
    planes = torch.empty(4 * sp, C, 288, device=x16.device, dtype=torch.float32)
    _call("ldm_gconv3x3_wgrad_bf16", _dev(x16, "x", BF16), _dev(dy16, "dy", BF16), _dev(planes, "planes"), B, H, W, C, sp)
    return reduce_partials(planes, 4 * sp, C * 288, torch.empty(C, 288, device=x16.device, dtype=torch.float32))


def film_hidden(p_rows, t_rows, out, B, HW, N):
    """out[b, pixel, :] = relu(P[pixel] + T[b]) -- Encodings.proj1 in separable form; ``out`` fp32 or bf16 [B*HW, N]."""
    bf = out.dtype == BF16
    _call("ldm_film_hidden", _dev(p_rows, "P"), _dev(t_rows, "T"), _dev(out, "out", BF16 if bf else torch.float32), int(bf), B, HW, N)
    return out


def film_hidden_bwd(dh, hid, B, HW, N):
    """-> (dP [HW, N], dT [B, N]) fp32: the sums over samples / over pixels of dh * (hid > 0); dh, hid fp32 or bf16."""
    bf = dh.dtype == BF16
    dt = BF16 if bf else torch.float32
    z = _lib.load().ldm_film_hidden_bwd_chunks(B, HW, N)
    ptiles = (HW + 31) // 32
    dev = dh.device
    dp_part = torch.empty(z, HW, N, device=dev, dtype=torch.float32)
    dt_part = torch.empty(ptiles, B, N, device=dev, dtype=torch.float32)
    _call("ldm_film_hidden_bwd", _dev(dh, "dh", dt), _dev(hid, "hid", dt), int(bf), _dev(dp_part, "dP_part"), _dev(dt_part, "dT_part"), B, HW, N, z)
    dp = dp_part[0] if z == 1 else reduce_partials(dp_part, z, HW * N, torch.empty(HW, N, device=dev, dtype=torch.float32))
    dtt = dt_part[0] if ptiles == 1 else reduce_partials(dt_part, ptiles, B * N, torch.empty(B, N, device=dev, dtype=torch.float32))
    return dp, dtt


# ---- bf16 sampling / decode (include/ldm_hip.h, "bf16 sampling / decode") ------------------------------------
def window_attention_bf16io(qkv, in_proj_bias, xf16, out16, B, H, W, C, ws, shift):
    """window_attention behind a bf16 in-projection (qkv fp32 or bf16 by its dtype): float "mask" read from the bf16 normalised
    input, bf16 context out."""
    q16 = qkv.dtype == BF16
    _call("ldm_window_attention_bf16io", _dev(qkv, "qkv", BF16 if q16 else torch.float32), int(q16), _dev(in_proj_bias, "in_proj_bias"),
          _opt(xf16, "xf16", BF16), _dev(out16, "out", BF16), B, H, W, C, ws, shift)
    return out16


def stem_nchw_bf16(x, w, bias, out16, B, Cin, HW, C0):
    _call("ldm_stem_nchw_bf16", _dev(x, "x"), _dev(w, "w"), _opt(bias, "bias"), _dev(out16, "out", BF16), B, Cin, HW, C0)
    return out16


def depth_to_space2_bf16(x16, B, H, W, C):
    """[B*H*W, 4C] bf16 with columns (dy, dx, c) -> [B*2H*2W, C] bf16."""
    out = torch.empty(B * 4 * H * W, C, device=x16.device, dtype=BF16)
    _call("ldm_depth_to_space2_bf16", _dev(x16, "x", BF16), _dev(out, "out", BF16), B, H, W, C)
    return out


def rgb_head_bf16(x16, w, bias, prev, out, B, H, W, C):
    _call("ldm_rgb_head_oc_bf16", _dev(x16, "x", BF16), _dev(w, "w"), _dev(bias, "bias"), _opt(prev, "prev"), _dev(out, "out"), B, H, W, C, w.shape[0])
    return out


def up2_add(coarse, skip, out, B, H, W, C):
    _call("ldm_up2_add_f32", _dev(coarse, "coarse"), _opt(skip, "skip"), _dev(out, "out"), B, H, W, C)
    return out


def avgpool2_bf16(x, out16, B, H, W, C):
    _call("ldm_avgpool2_bf16", _dev(x, "x"), _dev(out16, "out", BF16), B, H, W, C)
    return out16


def window_attention_bwd_bf16(qkv16, in_proj_bias, xf16, dctx16, dqkv16, dbias_pad, B, H, W, C, ws, shift):
    """window_attention_bwd with bf16 rows in (qkv, float-mask source, dctx) and bf16 dqkv out; dbias_pad fp32 [3C]."""
    _call("ldm_window_attention_bwd_bf16", _dev(qkv16, "qkv", BF16), _dev(in_proj_bias, "bias"), _opt(xf16, "xf", BF16), _dev(dctx16, "dctx", BF16),
          _dev(dqkv16, "dqkv", BF16), _dev(dbias_pad, "dbias_pad"), B, H, W, C, ws, shift)


def gconv_pack_bf16(w, fwd16, rot16):
    """conv.weight [C, 32, 3, 3] fp32 -> the grouped conv's bf16 filter tables: forward [C, 288] and data-gradient (flipped, in/out swapped) [C, 288]."""
    _call("ldm_gconv_pack_bf16", _dev(w, "w"), _dev(fwd16, "fwd", BF16), _dev(rot16, "rot", BF16), w.shape[0])


def pack3x3(w, want_dgrad=True):
    """(forward matrix [Cout, 9 Cin], data-gradient matrix [Cin, 9 Cout] or None) of a dense 3x3 conv weight [Cout, Cin, 3, 3]: one launch."""
    w = w.detach().contiguous()
    cout, cin = w.shape[0], w.shape[1]
    fwd = torch.empty(cout, 9 * cin, device=w.device, dtype=torch.float32)
    dgrad = torch.empty(cin, 9 * cout, device=w.device, dtype=torch.float32) if want_dgrad else None
    _call("ldm_pack3x3_f32", _dev(w, "w"), _dev(fwd, "fwd"), _opt(dgrad, "dgrad"), cout, cin)
    return fwd, dgrad


def replicate(src, reps):
    """-> [reps, n] fp32: ``reps`` separate copies of the vector ``src`` in one launch."""
    n = src.numel()
    out = torch.empty(reps, n, device=src.device, dtype=torch.float32)
    _call("ldm_replicate_f32", _dev(src, "src"), _dev(out, "out"), n, reps)
    return out


def window_attention_bf16_core(v):
    """1 (default): bf16-QKV attention on the bf16 matrix cores; 0: the fp32 16x16x4 core on the widened values.  Returns the old value."""
    return _lib.load().ldm_window_attention_bf16_core(v)


def window_attention_bwd_bf16_core(v):
    """1 (default): window_attention_bwd_bf16 on the bf16 matrix cores; 0: the fp32 16x16x4 core on the widened values.  Returns the old value."""
    return _lib.load().ldm_window_attention_bwd_bf16_core(v)


def gconv3x3_bf16_tiled(v):
    """1 (default): LDS-tiled bf16 grouped conv where the shape allows; 0: the direct kernel.  Bit-identical.  Returns the old value."""
    return _lib.load().ldm_gconv3x3_bf16_tiled(v)


def gemm_tn_ring(v):
    """1 (default): gemm_tn_bf16 on 256 x 256 tiles (one workgroup per CU) where the shape allows; 0: the 128-row kernel.  Returns the old value."""
    return _lib.load().ldm_gemm_tn_ring(v)


def tn_splits_one_round(tiles, m, slots=512, min_rows=256):
    """Splits of the pixel reduction of the fp32 TN kernel such that tiles x splits fills ONE round of its workgroup slots (two per
    CU) as fully as possible: the splits need not divide m (rows per split = m / splits rounded up to 32, the last one shorter).
    Powers of two left 9 x 64 = 576 workgroups on 512 slots -- two rounds, the second one an eighth full -- for EVERY dense 3x3 conv
    (9 (C / 128)^2 tiles)."""
    s = max(1, min(slots // max(tiles, 1), m // min_rows))

    def fits(v):
        return (v - 1) * (((m + v - 1) // v + 31) // 32 * 32) < m

    # multiples of 8 keep the kernel's XCD-aware order (the tiles of one split on one XCD: they share its rows of A and B in L2)
    if s >= 8:
        v = s - s % 8
        while v >= 8 and not fits(v):
            v -= 8
        if v >= 8 and v * 20 >= s * 17:                        # ... unless that leaves more than 15 % of the slots empty
            return v
    while s > 1 and not fits(s):
        s -= 1
    return s


def conv3x3_wgrad(dy, x, B, H, W, cin, cout, want_colsum=True):
    """(dW [cout, 9 * cin] fp32 with columns (tap, ci), column sums of dy [cout] or None) of a dense 3x3 conv (zero pad 1) from the
    gradient rows dy [B*H*W, cout] and the input rows x [B*H*W, cin]: implicit im2col inside the weight-gradient GEMM."""
    m = B * H * W
    npad, kpad = _lib.load().ldm_conv3x3_wgrad_npad(cout), (9 * cin + 127) // 128 * 128
    tiles = (npad // min(npad, 128)) * (kpad // 128)           # tile height 32 / 64 for cout <= 32 / 64, else 128
    s = tn_splits_one_round(tiles, m)
    dev = dy.device
    parts = torch.empty(s, npad, kpad, device=dev, dtype=torch.float32)
    cs = torch.empty(s, npad, device=dev, dtype=torch.float32) if want_colsum else None
    _call("ldm_conv3x3_wgrad_f32", _dev(dy, "dy"), dy.shape[1], _dev(x, "x"), _dev(parts, "parts"), _opt(cs, "colsum"), B, H, W, cin, cout, s)
    if s == 1:
        full, csum = parts[0], (cs[0] if want_colsum else None)
    else:
        full = torch.empty(npad, kpad, device=dev, dtype=torch.float32)
        if want_colsum:
            csum = torch.empty(npad, device=dev, dtype=torch.float32)
            reduce_partials_pair(parts, npad * kpad, full, cs, npad, csum, s)
        else:
            csum = None
            reduce_partials(parts, s, npad * kpad, full)
    return full[:cout, :9 * cin], (csum[:cout] if want_colsum else None)
