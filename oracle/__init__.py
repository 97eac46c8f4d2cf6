"""ctypes front-end of the CPU oracle (oracle/fn2_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by flownet2_amd/.  All arrays are host numpy float32, C-contiguous NCHW.

Parity status: see the header of fn2_oracle.c (pinned against the reference's own kernels through
oracle/_ref + tests/golden for every layer restated there).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfn2_oracle.so")

MULTIPLY, SUBTRACT = 0, 1
FILL_ZERO, FILL_NAN = 1, 2
NEAREST, LINEAR, CUBIC, AREA = 1, 2, 3, 4


class CorrParams(C.Structure):
    _fields_ = [("pad", C.c_int), ("kernel_size", C.c_int), ("max_displacement", C.c_int),
                ("stride1", C.c_int), ("stride2", C.c_int), ("corr_type", C.c_int), ("do_abs", C.c_int),
                ("single_direction", C.c_int)]


class L1Params(C.Structure):
    _fields_ = [("l2_per_location", C.c_int), ("l2_prescale_by_channels", C.c_int),
                ("normalize_by_num_entries", C.c_int), ("epsilon", C.c_float), ("plateau", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fn2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfn2_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _check(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} failed with status {rc}")


def num_threads() -> int:
    return lib().fn2_oracle_num_threads()


def corr_params(pad=0, kernel_size=1, max_displacement=0, stride1=1, stride2=1, corr_type=MULTIPLY, do_abs=0, single_direction=0):
    return CorrParams(pad, kernel_size, max_displacement, stride1, stride2, corr_type, do_abs, single_direction)


def correlation_out_shape(p: CorrParams, Cc, H, W):
    tc, th, tw = C.c_int(), C.c_int(), C.c_int()
    _check(lib().fn2_correlation_out_shape_cpu(C.byref(p), Cc, H, W, C.byref(tc), C.byref(th), C.byref(tw)), "correlation_out_shape")
    return tc.value, th.value, tw.value


def correlation_forward(p: CorrParams, b0, b1):
    b0, b1 = _f32(b0), _f32(b1)
    N, Cc, H, W = b0.shape
    tc, th, tw = correlation_out_shape(p, Cc, H, W)
    top = np.empty((N, tc, th, tw), np.float32)
    _check(lib().fn2_correlation_forward_cpu(C.byref(p), _p(b0), _p(b1), _p(top), N, Cc, H, W), "correlation_forward")
    return top


def correlation_backward(p: CorrParams, b0, b1, top_diff):
    b0, b1, top_diff = _f32(b0), _f32(b1), _f32(top_diff)
    N, Cc, H, W = b0.shape
    d0, d1 = np.empty_like(b0), np.empty_like(b1)
    _check(lib().fn2_correlation_backward_cpu(C.byref(p), _p(b0), _p(b1), _p(top_diff), _p(d0), _p(d1), N, Cc, H, W), "correlation_backward")
    return d0, d1


def correlation1d_out_shape(p: CorrParams, Cc, H, W):
    tc, th, tw = C.c_int(), C.c_int(), C.c_int()
    _check(lib().fn2_correlation1d_out_shape_cpu(C.byref(p), Cc, H, W, C.byref(tc), C.byref(th), C.byref(tw)), "correlation1d_out_shape")
    return tc.value, th.value, tw.value


def correlation1d_forward(p: CorrParams, b0, b1):
    b0, b1 = _f32(b0), _f32(b1)
    N, Cc, H, W = b0.shape
    tc, th, tw = correlation1d_out_shape(p, Cc, H, W)
    top = np.empty((N, tc, th, tw), np.float32)
    _check(lib().fn2_correlation1d_forward_cpu(C.byref(p), _p(b0), _p(b1), _p(top), N, Cc, H, W), "correlation1d_forward")
    return top


def correlation1d_backward(p: CorrParams, b0, b1, top_diff):
    b0, b1, top_diff = _f32(b0), _f32(b1), _f32(top_diff)
    N, Cc, H, W = b0.shape
    d0, d1 = np.empty_like(b0), np.empty_like(b1)
    _check(lib().fn2_correlation1d_backward_cpu(C.byref(p), _p(b0), _p(b1), _p(top_diff), _p(d0), _p(d1), N, Cc, H, W), "correlation1d_backward")
    return d0, d1


def flow_warp_forward(image, flow, fill_value=FILL_ZERO):
    image, flow = _f32(image), _f32(flow)
    N, Cc, H, W = image.shape
    assert flow.shape == (N, 2, H, W)
    out = np.empty_like(image)
    _check(lib().fn2_flow_warp_forward_cpu(_p(image), _p(flow), _p(out), N, Cc, H, W, fill_value), "flow_warp_forward")
    return out


def flow_warp_backward(image, flow, warped_diff, propagate_image=True, propagate_flow=True):
    image, flow, warped_diff = _f32(image), _f32(flow), _f32(warped_diff)
    N, Cc, H, W = image.shape
    di, df = np.empty_like(image), np.empty_like(flow)
    _check(lib().fn2_flow_warp_backward_cpu(_p(image), _p(flow), _p(warped_diff), _p(di), _p(df), N, Cc, H, W,
                                            int(propagate_image), int(propagate_flow)), "flow_warp_backward")
    return di, df


def resample_forward(x, Hout, Wout, type=LINEAR, antialias=True):
    x = _f32(x)
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, Hout, Wout), np.float32)
    _check(lib().fn2_resample_forward_cpu(_p(x), _p(out), N, Cc, H, W, Hout, Wout, type, int(antialias)), "resample_forward")
    return out


def l1_params(l2_per_location=False, l2_prescale_by_channels=False, normalize_by_num_entries=False,
              epsilon=1e-2, plateau=0.0):
    return L1Params(int(l2_per_location), int(l2_prescale_by_channels), int(normalize_by_num_entries),
                    float(epsilon), float(plateau))


def l1loss_forward(p: L1Params, b0, b1=None):
    b0 = _f32(b0)
    b1 = _f32(b1) if b1 is not None else None
    N, Cc, H, W = b0.shape
    loss, norm = C.c_float(), C.c_float()
    _check(lib().fn2_l1loss_forward_cpu(C.byref(p), _p(b0), _p(b1), C.byref(loss), C.byref(norm), N, Cc, H, W), "l1loss_forward")
    return loss.value, norm.value


def l1loss_backward(p: L1Params, b0, b1, top_diff, normalize_coeff):
    b0 = _f32(b0)
    b1 = _f32(b1) if b1 is not None else None
    N, Cc, H, W = b0.shape
    d0 = np.empty_like(b0)
    d1 = np.empty_like(b0) if b1 is not None else None
    _check(lib().fn2_l1loss_backward_cpu(C.byref(p), _p(b0), _p(b1), C.c_float(top_diff), C.c_float(normalize_coeff),
                                         _p(d0), _p(d1), N, Cc, H, W), "l1loss_backward")
    return d0, d1


class L1Scale(C.Structure):
    _fields_ = [("bottom0", C.c_void_p), ("bottom1", C.c_void_p), ("bottom0_diff", C.c_void_p), ("bottom1_diff", C.c_void_p),
                ("N", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("loss_weight", C.c_float)]


def _l1_scales(b0s, b1s, weights, d0s=None, d1s=None):
    arr = (L1Scale * len(b0s))()
    for k, b0 in enumerate(b0s):
        arr[k].bottom0 = b0.ctypes.data
        arr[k].bottom1 = b1s[k].ctypes.data if b1s[k] is not None else None
        arr[k].bottom0_diff = d0s[k].ctypes.data if d0s is not None else None
        arr[k].bottom1_diff = d1s[k].ctypes.data if d1s is not None and d1s[k] is not None else None
        arr[k].N, arr[k].C, arr[k].H, arr[k].W = b0.shape
        arr[k].loss_weight = float(weights[k])
    return arr


def l1loss_forward_multi(p: L1Params, b0s, b1s, weights):
    """(total, losses [n], norms [n]) of the loss layers of a net, net.cpp:565-579 order."""
    b0s = [_f32(b) for b in b0s]
    b1s = [_f32(b) if b is not None else None for b in b1s]
    n = len(b0s)
    losses, norms, total = np.empty(n, np.float32), np.empty(n, np.float32), C.c_float()
    _check(lib().fn2_l1loss_forward_multi_cpu(C.byref(p), n, _l1_scales(b0s, b1s, weights), _p(losses), _p(norms), C.byref(total)), "l1loss_forward_multi")
    return total.value, losses, norms


def l1loss_backward_multi(p: L1Params, b0s, b1s, weights, total_diff, norms):
    b0s = [_f32(b) for b in b0s]
    b1s = [_f32(b) if b is not None else None for b in b1s]
    d0s = [np.empty_like(b) for b in b0s]
    d1s = [np.empty_like(b) if b1s[k] is not None else None for k, b in enumerate(b0s)]
    norms = np.ascontiguousarray(norms, np.float32)
    _check(lib().fn2_l1loss_backward_multi_cpu(C.byref(p), len(b0s), _l1_scales(b0s, b1s, weights, d0s, d1s), C.c_float(total_diff), _p(norms)),
           "l1loss_backward_multi")
    return d0s, d1s


def channel_norm_forward(x):
    x = _f32(x)
    N, Cc, H, W = x.shape
    out = np.empty((N, 1, H, W), np.float32)
    _check(lib().fn2_channel_norm_forward_cpu(_p(x), _p(out), N, Cc, H, W), "channel_norm_forward")
    return out


def channel_norm_backward(x, top, top_diff):
    x, top, top_diff = _f32(x), _f32(top), _f32(top_diff)
    N, Cc, H, W = x.shape
    d = np.empty_like(x)
    _check(lib().fn2_channel_norm_backward_cpu(_p(x), _p(top), _p(top_diff), _p(d), N, Cc, H, W), "channel_norm_backward")
    return d


def downsample_forward(x, Hout, Wout):
    x = _f32(x)
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, Hout, Wout), np.float32)
    _check(lib().fn2_downsample_forward_cpu(_p(x), _p(out), N, Cc, H, W, Hout, Wout), "downsample_forward")
    return out


def downsample_forward_multi(x, sizes):
    """fn2_downsample_forward_multi_cpu: the Downsample layers of `sizes` on one bottom, one after another."""
    import ctypes as C
    x = _f32(x)
    N, Cc, H, W = x.shape
    outs = [np.empty((N, Cc, int(h), int(w)), np.float32) for h, w in sizes]
    n = len(outs)
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    hs, ws = (C.c_int * n)(*[int(h) for h, _ in sizes]), (C.c_int * n)(*[int(w) for _, w in sizes])
    f = lib().fn2_downsample_forward_multi_cpu
    f.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)] + [C.c_int] * 5
    _check(f(x.ctypes.data, ptrs, hs, ws, n, N, Cc, H, W), "downsample_forward_multi")
    return outs


def predict_flow_conv_forward(x, weight, bias=None):
    x, weight = _f32(x), _f32(weight)
    bias = _f32(bias) if bias is not None else None
    N, Cc, H, W = x.shape
    assert weight.shape == (2, Cc, 3, 3)
    out = np.empty((N, 2, H, W), np.float32)
    _check(lib().fn2_predict_flow_conv_forward_cpu(_p(x), _p(weight), _p(bias), _p(out), N, Cc, H, W), "predict_flow_conv_forward")
    return out


def bias_leaky_relu_forward(x, bias=None, negative_slope=0.1):
    """Returns leaky_relu(x + bias[c]) (the C twin works in place on a copy)."""
    out = np.array(_f32(x), copy=True)
    bias = _f32(bias) if bias is not None else None
    N, Cc, H, W = out.shape
    _check(lib().fn2_bias_leaky_relu_forward_cpu(_p(out), _p(bias), N, Cc, H, W, C.c_float(negative_slope)), "bias_leaky_relu_forward")
    return out


def bias_leaky_relu_backward(top_data, top_diff, negative_slope=0.1):
    y, g = _f32(top_data), _f32(top_diff)
    N, Cc, H, W = y.shape
    d, db = np.empty_like(g), np.empty(Cc, np.float32)
    _check(lib().fn2_bias_leaky_relu_backward_cpu(_p(y), _p(g), _p(d), _p(db), N, Cc, H, W, C.c_float(negative_slope)), "bias_leaky_relu_backward")
    return d, db


def im2col_forward(x, kernel, pad, stride):
    x = _f32(x)
    N, Cc, H, W = x.shape
    Hc, Wc = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    col = np.empty((N, Cc * kernel * kernel, Hc * Wc), np.float32)
    _check(lib().fn2_im2col_forward_cpu(_p(x), _p(col), N, Cc, H, W, kernel, pad, stride), "im2col_forward")
    return col


def col2im_bias_relu_forward(col, bias, N, Cc, H, W, kernel, pad, stride, relu=True, negative_slope=0.1):
    col = _f32(col)
    bias = _f32(bias) if bias is not None else None
    out = np.empty((N, Cc, H, W), np.float32)
    _check(lib().fn2_col2im_bias_relu_forward_cpu(_p(col), _p(bias), _p(out), N, Cc, H, W, kernel, pad, stride, int(bool(relu)),
                                                  C.c_float(negative_slope)), "col2im_bias_relu_forward")
    return out


def conv_k7s2_relu_forward(x, weight, bias=None, negative_slope=0.1):
    x, weight = _f32(x), _f32(weight)
    bias = _f32(bias) if bias is not None else None
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    assert weight.shape == (Cout, Cin, 7, 7)
    out = np.empty((N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1), np.float32)
    _check(lib().fn2_conv_k7s2_relu_forward_cpu(_p(x), _p(weight), _p(bias), _p(out), N, Cin, H, W, Cout, C.c_float(negative_slope)),
           "conv_k7s2_relu_forward")
    return out


def conv_mfma_pack_weights(weight):
    w = _f32(weight)
    Cout, Cin, k, _ = w.shape
    L = lib()
    L.fn2_conv_mfma_packed_floats_cpu.restype = C.c_size_t
    n = L.fn2_conv_mfma_packed_floats_cpu(Cout, Cin, k)
    assert n > 0, "unsupported weight shape"
    packed = np.empty(n, np.float32)
    _check(L.fn2_conv_mfma_pack_weights_cpu(_p(w), _p(packed), Cout, Cin, k), "conv_mfma_pack_weights")
    return packed


def conv_mfma_pack_weights_view(blob, Cout, Cin, kernel, src_cout, src_cin, stride_cout, stride_cin, flip=False):
    """The packed operand of the strided [Cout][Cin][k][k] view of `blob` (any contiguous float array)."""
    w = _f32(blob)
    L = lib()
    L.fn2_conv_mfma_packed_floats_cpu.restype = C.c_size_t
    n = L.fn2_conv_mfma_packed_floats_cpu(Cout, Cin, kernel)
    assert n > 0, "unsupported weight shape"
    packed = np.empty(n, np.float32)
    _check(L.fn2_conv_mfma_pack_weights_view_cpu(_p(w), _p(packed), Cout, Cin, kernel, src_cout, src_cin, C.c_longlong(stride_cout),
                                                  C.c_longlong(stride_cin), int(bool(flip))), "conv_mfma_pack_weights_view")
    return packed


def conv_mfma_forward(x, packed, bias, Cout, kernel, stride, pad, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    x, packed = _f32(x), _f32(packed)
    bias = _f32(bias) if bias is not None else None
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Ho, Wo = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    if out is None:
        out = np.zeros((N, Cout, Ho, Wo), np.float32)
    _check(lib().fn2_conv_mfma_forward_cpu(_p(x), _p(packed), _p(bias), _p(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                           kernel, stride, pad, int(bool(relu)), C.c_float(negative_slope)), "conv_mfma_forward")
    return out


def conv_plane_forward(x, packed, bias, Cout, stride, pad, ksplit, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None, kernel=3):
    """CPU twin of fn2_conv_plane_k_forward (packed = conv_mfma_pack_weights(weight [Cout, Cin, kernel, kernel])); ksplit = fn2_conv_plane_k_ksplit()."""
    x = _f32(x)
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Ho, Wo = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    if out is None:
        out = np.zeros((N, Cout, Ho, Wo), np.float32)
    bias = _f32(bias) if bias is not None else None
    _check(lib().fn2_conv_plane_k_forward_cpu(_p(x), _p(packed), _p(bias), _p(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                              int(kernel), stride, pad, int(bool(relu)), C.c_float(negative_slope), int(ksplit)), "conv_plane_forward")
    return out


def conv_k7s2_wgrad(top_diff, bottom, parts):
    """CPU twin of fn2_conv_k7s2_wgrad: weight gradient [Cout, Cin, 7, 7] of the 7x7 / 2 / 3 stem; parts = fn2_conv_k7s2_wgrad_ksplit()."""
    d, b = _f32(top_diff), _f32(bottom)
    N, Cout = d.shape[:2]
    Cin, H, W = b.shape[1:]
    dw = np.zeros((Cout, Cin, 7, 7), np.float32)
    _check(lib().fn2_conv_k7s2_wgrad_cpu(_p(d), _p(b), _p(dw), N, Cin, H, W, Cout, 0, int(parts)), "conv_k7s2_wgrad")
    return dw


def scale_shift_forward(x, scale, shift=None, out=None, out_c0=0):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    if out is None:
        out = np.zeros_like(x)
    sh = None if shift is None else np.ascontiguousarray(shift, np.float32)
    _check(lib().fn2_scale_shift_forward_cpu(_p(x), _p(out), _p(sh), N, Cc, H, W, out.shape[1], out_c0, C.c_float(scale)), "scale_shift_forward")
    return out


def tconv_forward(x, weight, bias, kernel, pad, out_hw=None, relu=False, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    """CPU twin of fn2_tconv_forward; weight: the unpacked [Cin, Cout, k, k] blob."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(weight, np.float32)
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Cout = w.shape[1]
    Ho, Wo = out_hw if out_hw is not None else (2 * (H - 1) + kernel - 2 * pad, 2 * (W - 1) + kernel - 2 * pad)
    if out is None:
        out = np.zeros((N, Cout, Ho, Wo), np.float32)
    bias = None if bias is None else np.ascontiguousarray(bias, np.float32)
    _check(lib().fn2_tconv_forward_cpu(_p(x), _p(w), _p(bias), _p(out), N, Cin, H, W, Ctot, in_c0, Cout, Ho, Wo, out.shape[1], out_c0,
                                       int(kernel), int(pad), int(bool(relu)), C.c_float(negative_slope)), "tconv_forward")
    return out


def conv_wgrad(a, b, kernel, stride, pad, ksplit, out=None, accumulate=False, a_c0=0, Ca=None, b_c0=0, Cb=None):
    """CPU twin of fn2_conv_wgrad: dw[ca][cb][ky][kx] (+)= sum a[n, a_c0+ca, y, x] * b[n, b_c0+cb, stride y + ky - pad, stride x + kx - pad];
    ksplit = fn2_conv_wgrad_ksplit() (it fixes the summation order)."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    N, Atot, Ha, Wa = a.shape
    _, Btot, Hb, Wb = b.shape
    Ca = Atot - a_c0 if Ca is None else Ca
    Cb = Btot - b_c0 if Cb is None else Cb
    if out is None:
        out, accumulate = np.zeros((Ca, Cb, kernel, kernel), np.float32), False
    _check(lib().fn2_conv_wgrad_cpu(_p(a), _p(b), _p(out), N, Ca, Ha, Wa, Atot, a_c0, Cb, Hb, Wb, Btot, b_c0, int(kernel), int(stride), int(pad),
                                    int(bool(accumulate)), int(ksplit)), "conv_wgrad")
    return out


def deconv_plane_pack_weights(weight):
    """weight [Cin, Cout, 4, 4] (Caffe's deconvolution blob) -> the per-parity-class MFMA operand order of fn2_deconv_plane_forward."""
    w = _f32(weight)
    Cin, Cout = w.shape[:2]
    assert w.shape[2:] in ((4, 4), (3, 3))           # 3x3: read as the 4x4 blob whose fourth tap row / column are zero
    L = lib()
    L.fn2_deconv_plane_packed_floats_cpu.restype = C.c_size_t
    n = L.fn2_deconv_plane_packed_floats_cpu(Cin, Cout)
    assert n > 0, "unsupported weight shape"
    packed = np.empty(n, np.float32)
    _check(L.fn2_deconv_plane_pack_weights_k_cpu(_p(w), _p(packed), Cin, Cout, int(w.shape[2])), "deconv_plane_pack_weights")
    return packed


def deconv_plane_forward(x, packed, bias, Cout, ksplit, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    """CPU twin of fn2_deconv_plane_forward; ksplit = fn2_deconv_plane_ksplit()."""
    x, packed = _f32(x), _f32(packed)
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    if out is None:
        out = np.zeros((N, Cout, 2 * H, 2 * W), np.float32)
    bias = _f32(bias) if bias is not None else None
    _check(lib().fn2_deconv_plane_forward_cpu(_p(x), _p(packed), _p(bias), _p(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                              int(bool(relu)), C.c_float(negative_slope), int(ksplit)), "deconv_plane_forward")
    return out


def conv_wino_pack_weights(weight):
    w = _f32(weight)
    Cout, Cin, k, _ = w.shape
    assert k == 3
    L = lib()
    L.fn2_conv_wino_packed_floats_cpu.restype = C.c_size_t
    n = L.fn2_conv_wino_packed_floats_cpu(Cout, Cin)
    assert n > 0, "unsupported weight shape"
    packed = np.empty(n, np.float32)
    _check(L.fn2_conv_wino_pack_weights_cpu(_p(w), _p(packed), Cout, Cin), "conv_wino_pack_weights")
    return packed


def conv_wino_forward(x, packed, bias, Cout, pad=1, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    x, packed = _f32(x), _f32(packed)
    bias = _f32(bias) if bias is not None else None
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    if out is None:
        out = np.zeros((N, Cout, H + 2 * pad - 2, W + 2 * pad - 2), np.float32)
    _check(lib().fn2_conv_wino_forward_cpu(_p(x), _p(packed), _p(bias), _p(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                           pad, int(bool(relu)), C.c_float(negative_slope)), "conv_wino_forward")
    return out


def upsample_flow_deconv_forward(x, weight, bias=None):
    x, weight = _f32(x), _f32(weight)
    bias = _f32(bias) if bias is not None else None
    N, Cc, H, W = x.shape
    assert Cc == 2 and weight.shape == (2, 2, 4, 4)
    out = np.empty((N, 2, 2 * H, 2 * W), np.float32)
    _check(lib().fn2_upsample_flow_deconv_forward_cpu(_p(x), _p(weight), _p(bias), _p(out), N, H, W), "upsample_flow_deconv_forward")
    return out


def predict_flow_conv_backward(x, weight, top_diff):
    """(bottom_diff, weight_diff, bias_diff) of predict_flow (Convolution{3,1,1} C -> 2), accumulated in double."""
    x, weight, g = _f32(x), _f32(weight), _f32(top_diff)
    N, Cc, H, W = x.shape
    assert weight.shape == (2, Cc, 3, 3) and g.shape == (N, 2, H, W)
    dx, dw, db = np.empty_like(x), np.empty_like(weight), np.empty(2, np.float32)
    _check(lib().fn2_predict_flow_conv_backward_cpu(_p(x), Cc, 0, _p(weight), _p(g), _p(dx), _p(dw), _p(db), N, Cc, H, W, 0), "predict_flow_conv_backward")
    return dx, dw, db


def upsample_flow_deconv_backward(x, weight, top_diff):
    """(bottom_diff, weight_diff, bias_diff) of upsample_flow (Deconvolution{4,2,1} 2 -> 2), accumulated in double."""
    x, weight, g = _f32(x), _f32(weight), _f32(top_diff)
    N, Cc, H, W = x.shape
    assert Cc == 2 and weight.shape == (2, 2, 4, 4) and g.shape == (N, 2, 2 * H, 2 * W)
    dx, dw, db = np.empty_like(x), np.empty_like(weight), np.empty(2, np.float32)
    _check(lib().fn2_upsample_flow_deconv_backward_cpu(_p(x), _p(weight), _p(g), _p(dx), _p(dw), _p(db), N, H, W, 0), "upsample_flow_deconv_backward")
    return dx, dw, db


# ------------------------------------------------------------------------------------------------
# CustomData sample format (Datum wire format, writer packing, decode) -- host arrays throughout
# ------------------------------------------------------------------------------------------------
class DatumView(C.Structure):
    _fields_ = [("channels", C.c_int), ("height", C.c_int), ("width", C.c_int), ("label", C.c_int), ("encoded", C.c_int),
                ("data", C.c_void_p), ("data_bytes", C.c_size_t), ("float_data_count", C.c_size_t)]


def _ints(v):
    return (C.c_int * max(1, len(v)))(*[int(x) for x in v]), len(v)


def datum_parse(record: bytes):
    """-> dict(channels, height, width, label, encoded, data (bytes or None), float_data (array or None))"""
    a = np.frombuffer(record, dtype=np.uint8)
    v = DatumView()
    L = lib()
    L.fn2_datum_parse_cpu.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(DatumView)]
    _check(L.fn2_datum_parse_cpu(C.c_void_p(a.ctypes.data), a.size, C.byref(v)), "datum_parse")
    data = bytes(a[v.data - a.ctypes.data: v.data - a.ctypes.data + v.data_bytes]) if v.data else None
    fl = None
    if v.float_data_count:
        fl = np.empty(v.float_data_count, np.float32)
        L.fn2_datum_float_data_cpu.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _check(L.fn2_datum_float_data_cpu(C.c_void_p(a.ctypes.data), a.size, C.c_void_p(fl.ctypes.data), fl.size), "datum_float_data")
    return dict(channels=v.channels, height=v.height, width=v.width, label=v.label, encoded=bool(v.encoded), data=data, float_data=fl)


def datum_serialize(channels, height, width, data: bytes, label=0) -> bytes:
    a = np.frombuffer(data, dtype=np.uint8)
    L = lib()
    L.fn2_datum_serialize_cpu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    L.fn2_datum_serialize_cpu.restype = C.c_longlong
    need = L.fn2_datum_serialize_cpu(channels, height, width, C.c_void_p(a.ctypes.data), a.size, label, None, 0)
    _check(min(need, 0), "datum_serialize")
    out = np.empty(need, np.uint8)
    _check(min(L.fn2_datum_serialize_cpu(channels, height, width, C.c_void_p(a.ctypes.data), a.size, label, C.c_void_p(out.ctypes.data), out.size), 0),
           "datum_serialize")
    return out.tobytes()


def custom_data_sample_bytes(channels, H, W, slice_points, encodings) -> int:
    sp, nsp = _ints(slice_points)
    en, nen = _ints(encodings)
    L = lib()
    L.fn2_custom_data_sample_bytes_cpu.restype = C.c_size_t
    return L.fn2_custom_data_sample_bytes_cpu(channels, H, W, sp, nsp, en, nen)


def custom_data_encode_sample(img0_hwc, img1_hwc, flow_chw=None, occlusion=None) -> bytes:
    a, b = np.ascontiguousarray(img0_hwc, np.uint8), np.ascontiguousarray(img1_hwc, np.uint8)
    H, W = a.shape[:2]
    f = np.ascontiguousarray(flow_chw, np.float32) if flow_chw is not None else None
    o = np.ascontiguousarray(occlusion).astype(np.uint8) if occlusion is not None else None
    out = np.empty(10 * H * W + (H * W - 1) // 8 + 1, np.uint8)
    p = lambda x: C.c_void_p(x.ctypes.data) if x is not None else None
    L = lib()
    L.fn2_custom_data_encode_sample_cpu.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    _check(L.fn2_custom_data_encode_sample_cpu(p(a), p(b), p(f), p(o), H, W, p(out), out.size), "custom_data_encode_sample")
    return out.tobytes()


def custom_data_decode(samples: np.ndarray, channels, H, W, slice_points=(), encodings=(), mean=None, scale=1.0, float_data=False):
    """samples: [N, stride] uint8 (or float32 with float_data) host array -> list of float32 [N, slice channels, H, W]."""
    samples = np.ascontiguousarray(samples)
    N = samples.shape[0]
    stride = samples.shape[1] * samples.itemsize
    bounds = [0] + [int(s) for s in slice_points] + [channels]
    tops = [np.empty((N, max(b - a, 0), H, W), np.float32) for a, b in zip(bounds, bounds[1:])]
    sp, nsp = _ints(slice_points)
    en, nen = _ints(encodings)
    ptrs = (C.c_void_p * len(tops))(*[t.ctypes.data for t in tops])
    m = np.ascontiguousarray(mean, np.float32) if mean is not None else None
    L = lib()
    L.fn2_custom_data_decode_forward_cpu.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                                     C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_float, C.POINTER(C.c_void_p)]
    _check(L.fn2_custom_data_decode_forward_cpu(C.c_void_p(samples.ctypes.data), stride, N, channels, H, W, sp, nsp, en, nen, int(float_data),
                                                C.c_void_p(m.ctypes.data) if m is not None else None, float(scale), ptrs), "custom_data_decode")
    return tops


def augmentation_matrix(coeffs, crop_width, crop_height, bottom_width, bottom_height, invert=False):
    c = np.ascontiguousarray(coeffs, np.float32)
    out = np.empty(6, np.float32)
    _check(lib().fn2_augmentation_matrix_cpu(_p(c), crop_width, crop_height, bottom_width, bottom_height, int(invert), _p(out)), "augmentation_matrix")
    return out


def flow_augmentation_forward(flow, coeffs1, coeffs2, crop_height, crop_width):
    flow = _f32(flow)
    N, _, H, W = flow.shape
    c1, c2 = _f32(coeffs1).reshape(N, 42), _f32(coeffs2).reshape(N, 42)
    top = np.empty((N, 2, crop_height, crop_width), np.float32)
    _check(lib().fn2_flow_augmentation_forward_cpu(_p(flow), _p(c1), _p(c2), _p(top), N, H, W, crop_height, crop_width), "flow_augmentation_forward")
    return top


class DataAugParams(C.Structure):
    _fields_ = [("crop_width", C.c_int), ("crop_height", C.c_int), ("max_multiplier", C.c_float), ("has_chromatic_eigvec", C.c_int),
                ("chromatic_eigvec", C.c_float * 9), ("mean_mode", C.c_int), ("noise_seed", C.c_ulonglong), ("noise_stream", C.c_ulonglong)]


def philox4x32_10(counter, key):
    c = (C.c_uint32 * 4)(*[int(v) & 0xffffffff for v in counter])
    k = (C.c_uint32 * 2)(*[int(v) & 0xffffffff for v in key])
    out = (C.c_uint32 * 4)()
    _check(lib().fn2_philox4x32_10_cpu(c, k, out), "philox")
    return [int(v) for v in out]


def data_augmentation_forward(bottom, coeffs=None, crop_height=0, crop_width=0, mean=None, mean_mode=0, max_multiplier=255.0, chromatic_eigvec=None,
                              noise_seed=0, noise_stream=0):
    bottom = _f32(bottom)
    N, Cc, H, W = bottom.shape
    crop = crop_width > 0 and crop_height > 0
    ch, cw = (crop_height, crop_width) if crop else (H, W)
    p = DataAugParams(crop_width, crop_height, max_multiplier, int(chromatic_eigvec is not None))
    if chromatic_eigvec is not None:
        p.chromatic_eigvec = (C.c_float * 9)(*[float(v) for v in chromatic_eigvec])
    p.mean_mode = mean_mode
    p.noise_seed, p.noise_stream = int(noise_seed), int(noise_stream)
    co = _f32(coeffs).reshape(N, 42) if coeffs is not None else None
    m = _f32(mean) if mean is not None else None
    top = np.empty((N, Cc, max(ch, 1), max(cw, 1)), np.float32)
    _check(lib().fn2_data_augmentation_forward_cpu(C.byref(p), _p(bottom), _p(co), _p(m), _p(top), N, Cc, H, W), "data_augmentation_forward")
    return top


def custom_data_stage_records(records):
    """-> (samples uint8 [N, sample_bytes], (channels, H, W), labels) through the oracle's staging restatement."""
    n = len(records)
    bufs = [np.frombuffer(r, dtype=np.uint8) for r in records]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    ch, h, w, nb = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    labels = (C.c_int * n)()
    L = lib()
    L.fn2_custom_data_stage_records_cpu.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    _check(L.fn2_custom_data_stage_records_cpu(ptrs, lens, n, None, 0, C.byref(ch), C.byref(h), C.byref(w), C.byref(nb), labels), "custom_data_stage_records")
    out = np.zeros((n, nb.value), np.uint8)
    _check(L.fn2_custom_data_stage_records_cpu(ptrs, lens, n, C.c_void_p(out.ctypes.data), nb.value, None, None, None, None, None), "custom_data_stage_records")
    return out, (ch.value, h.value, w.value), list(labels)
