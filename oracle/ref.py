"""ctypes front-end of oracle/_ref/libfn2_ref.so: the REFERENCE's own layer classes (correlation, flow-warp,
resample, channel-norm, downsample), compiled in place from /root/reference as HIP by oracle/ref_build.sh.
TEST INFRASTRUCTURE ONLY.  GPU modes need an MI355X; the CPU modes (FlowWarp, ChannelNorm) run anywhere."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libfn2_ref.so")
# the same C shim (oracle/ref_shim.cpp) linked with OUR Caffe adapter instead of the reference's layer sources
ADAPTER_SO = os.path.join(_HERE, "..", "flownet2_amd", "csrc", "caffe_adapter", "_build", "libfn2_caffe_adapter_test.so")
_lib = None
_which = SO


def available() -> bool:
    return os.path.exists(SO)


def adapter_available() -> bool:
    return os.path.exists(ADAPTER_SO)


def use(which: str):
    """'ref' = the reference's own layers (oracle/_ref), 'adapter' = flownet2_amd's Caffe adapter."""
    global _lib, _which
    _which = SO if which == "ref" else ADAPTER_SO
    _lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_which)
        _lib.fn2ref_last_error.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _chk(rc):
    if rc != 0:
        raise RuntimeError("reference layer failed: " + lib().fn2ref_last_error().decode())


def correlation(b0, b1, pad, kernel_size, max_displacement, stride1, stride2, corr_type=0, top_diff=None):
    b0, b1 = _f(b0), _f(b1)
    N, Cc, H, W = b0.shape
    shape = (C.c_int * 4)()
    # first call for the shape only (top_out NULL)
    _chk(lib().fn2ref_correlation(pad, kernel_size, max_displacement, stride1, stride2, corr_type, _p(b0), _p(b1), N, Cc, H, W,
                                  None, shape, None, None, None))
    top = np.empty(tuple(shape), np.float32)
    if top_diff is None:
        _chk(lib().fn2ref_correlation(pad, kernel_size, max_displacement, stride1, stride2, corr_type, _p(b0), _p(b1), N, Cc, H, W,
                                      _p(top), shape, None, None, None))
        return top
    td = _f(top_diff)
    d0, d1 = np.empty_like(b0), np.empty_like(b1)
    _chk(lib().fn2ref_correlation(pad, kernel_size, max_displacement, stride1, stride2, corr_type, _p(b0), _p(b1), N, Cc, H, W,
                                  _p(top), shape, _p(td), _p(d0), _p(d1)))
    return top, d0, d1


def correlation1d(b0, b1, pad, kernel_size, max_displacement, stride1, stride2, corr_type=0, single_direction=0, top_diff=None):
    """Correlation1DLayer of the reference (correlation_layer1d.cpp/.cu); returns top, or (top, b0_diff, b1_diff)."""
    b0, b1 = _f(b0), _f(b1)
    N, Cc, H, W = b0.shape
    shape = (C.c_int * 4)()
    args = (pad, kernel_size, max_displacement, stride1, stride2, corr_type, single_direction, _p(b0), _p(b1), N, Cc, H, W)
    _chk(lib().fn2ref_correlation1d(*args, None, shape, None, None, None))
    top = np.empty(tuple(shape), np.float32)
    if top_diff is None:
        _chk(lib().fn2ref_correlation1d(*args, _p(top), shape, None, None, None))
        return top
    td = _f(top_diff)
    d0, d1 = np.empty_like(b0), np.empty_like(b1)
    _chk(lib().fn2ref_correlation1d(*args, _p(top), shape, _p(td), _p(d0), _p(d1)))
    return top, d0, d1


def flow_warp(image, flow, fill_value=1, warped_diff=None, cpu=False):
    image, flow = _f(image), _f(flow)
    N, Cc, H, W = image.shape
    out = np.empty_like(image)
    if warped_diff is None:
        _chk(lib().fn2ref_flow_warp(int(cpu), fill_value, _p(image), _p(flow), N, Cc, H, W, _p(out), None, None, None))
        return out
    wd = _f(warped_diff)
    di, df = np.empty_like(image), np.empty_like(flow)
    _chk(lib().fn2ref_flow_warp(int(cpu), fill_value, _p(image), _p(flow), N, Cc, H, W, _p(out), _p(wd), _p(di), _p(df)))
    return out, di, df


def resample(x, Hout, Wout, type=2, antialias=True):
    x = _f(x)
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, Hout, Wout), np.float32)
    _chk(lib().fn2ref_resample(type, int(antialias), _p(x), N, Cc, H, W, Hout, Wout, _p(out)))
    return out


def channel_norm(x, top_diff=None, cpu=False):
    x = _f(x)
    N, Cc, H, W = x.shape
    out = np.empty((N, 1, H, W), np.float32)
    if top_diff is None:
        _chk(lib().fn2ref_channel_norm(int(cpu), _p(x), N, Cc, H, W, _p(out), None, None))
        return out
    d = np.empty_like(x)
    _chk(lib().fn2ref_channel_norm(1, _p(x), N, Cc, H, W, _p(out), _p(_f(top_diff)), _p(d)))
    return out, d


def downsample(x, Hout, Wout):
    x = _f(x)
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, Hout, Wout), np.float32)
    _chk(lib().fn2ref_downsample(_p(x), N, Cc, H, W, Hout, Wout, _p(out)))
    return out


def l1loss(b0, b1=None, l2_per_location=False, l2_prescale_by_channels=False, normalize_by_num_entries=False,
           epsilon=1e-2, plateau=0.0, loss_weight=1.0):
    """L1LossLayer forward + backward: returns (loss, loss * loss_weight, bottom[0] diff, bottom[1] diff or None)."""
    b0 = _f(b0)
    b1 = _f(b1) if b1 is not None else None
    N, Cc, H, W = b0.shape
    loss, wl = C.c_float(), C.c_float()
    d0 = np.empty_like(b0)
    d1 = np.empty_like(b0) if b1 is not None else None
    L = lib()
    L.fn2ref_l1loss.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    _chk(L.fn2ref_l1loss(int(l2_per_location), int(l2_prescale_by_channels), int(normalize_by_num_entries), epsilon, plateau, loss_weight,
                         _p(b0), _p(b1), N, Cc, H, W, C.byref(loss), C.byref(wl), _p(d0), _p(d1)))
    return loss.value, wl.value, d0, d1


def flownetc_time(N, H, W, warmup=3, iterations=10, use_cache=True, max_layers=128):
    """fn2ref_flownetc_time (adapter build, use("adapter") first): a FlowNetC core forward chained from LayerRegistry-created layers -- the
    adapter's Convolution / Deconvolution / Correlation plug-ins, the reference's own in-place ReLU and Concat -- timed like `caffe time`
    (tools/caffe.cpp:346-366).  -> dict(total_ms, layers=[(name, ms)], packs, pack_reuses, output_finite)."""
    L = lib()
    if not hasattr(L, "fn2ref_flownetc_time"):
        raise RuntimeError("this shim build has no FlowNetC driver (built without the reference's ReLU / Concat sources)")
    total = C.c_double()
    ms = (C.c_double * max_layers)()
    names = C.create_string_buffer(48 * max_layers)
    n, packs, reuses, ok = C.c_int(), C.c_longlong(), C.c_longlong(), C.c_int()
    L.fn2ref_flownetc_time.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                                       C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
    _chk(L.fn2ref_flownetc_time(N, H, W, warmup, iterations, int(bool(use_cache)), C.byref(total), ms, names, max_layers, C.byref(n),
                                C.byref(packs), C.byref(reuses), C.byref(ok)))
    layers = [(names.raw[48 * i:48 * (i + 1)].split(b"\0", 1)[0].decode(), ms[i]) for i in range(min(n.value, max_layers))]
    return {"total_ms": total.value, "layers": layers, "packs": packs.value, "pack_reuses": reuses.value, "output_finite": bool(ok.value)}


def convolution(x, weight, bias=None, kernel=3, stride=1, pad=1, deconv=False, relu=False, negative_slope=0.1):
    """The reference's stock Convolution / Deconvolution layer (+ in-place ReLU) with the given weights."""
    x, weight = _f(x), _f(weight)
    bias = _f(bias) if bias is not None else None
    N, Cc, H, W = x.shape
    num_output = weight.shape[1] if deconv else weight.shape[0]
    L = lib()
    fp = C.POINTER(C.c_float)
    L.fn2ref_convolution.argtypes = [C.c_int] * 6 + [C.c_float, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.POINTER(C.c_int)]
    shape = (C.c_int * 4)()
    args = (int(deconv), kernel, stride, pad, num_output, int(relu), negative_slope, _p(x), N, Cc, H, W, _p(weight), _p(bias))
    _chk(L.fn2ref_convolution(*args, None, shape))
    out = np.empty(tuple(shape), np.float32)
    _chk(L.fn2ref_convolution(*args, _p(out), shape))
    return out


def convolution_by_registry(x, weight, bias=None, kernel=3, stride=1, pad=1, deconv=False):
    """Convolution / Deconvolution created by prototxt type string through LayerRegistry -- in the ADAPTER library these are the plug-ins of
    flownet2_amd/csrc/caffe_adapter (use("adapter") first)."""
    x, weight = _f(x), _f(weight)
    bias = _f(bias) if bias is not None else None
    N, Cc, H, W = x.shape
    num_output = weight.shape[1] if deconv else weight.shape[0]
    L = lib()
    fp = C.POINTER(C.c_float)
    L.fn2ref_convolution_by_registry.argtypes = [C.c_int] * 6 + [C.c_float, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.POINTER(C.c_int)]
    shape = (C.c_int * 4)()
    args = (int(deconv), kernel, stride, pad, num_output, 0, 0.0, _p(x), N, Cc, H, W, _p(weight), _p(bias))
    _chk(L.fn2ref_convolution_by_registry(*args, None, shape))
    out = np.empty(tuple(shape), np.float32)
    _chk(L.fn2ref_convolution_by_registry(*args, _p(out), shape))
    return out


def convolution_backward(x, weight, bias, top_diff, weight_diff0=None, bias_diff0=None, kernel=3, stride=1, pad=1, deconv=False, by_registry=False):
    """Forward + Backward of the reference's stock Convolution / Deconvolution (by_registry: of whatever LayerRegistry creates for the type
    string -- the adapter's plug-ins after use("adapter")).  The parameter diffs start from weight_diff0 / bias_diff0 (the reference
    accumulates into them).  Returns (bottom_diff, weight_diff, bias_diff or None)."""
    x, weight, top_diff = _f(x), _f(weight), _f(top_diff)
    bias = _f(bias) if bias is not None else None
    N, Cc, H, W = x.shape
    num_output = weight.shape[1] if deconv else weight.shape[0]
    wd0 = _f(weight_diff0) if weight_diff0 is not None else np.zeros_like(weight)
    bd0 = (_f(bias_diff0) if bias_diff0 is not None else np.zeros_like(bias)) if bias is not None else None
    dx, dw = np.empty_like(x), np.empty_like(weight)
    db = np.empty_like(bias) if bias is not None else None
    L = lib()
    fp = C.POINTER(C.c_float)
    fn = L.fn2ref_convolution_backward_by_registry if by_registry else L.fn2ref_convolution_backward
    fn.argtypes = [C.c_int] * 5 + [fp, C.c_int, C.c_int, C.c_int, C.c_int] + [fp] * 8
    _chk(fn(int(deconv), kernel, stride, pad, num_output, _p(x), N, Cc, H, W, _p(weight), _p(bias), _p(top_diff), _p(wd0), _p(bd0), _p(dx), _p(dw), _p(db)))
    return dx, dw, db


def custom_data(records, batch_size, slice_points=(), encodings=(), scale=1.0, subtract=(), range_start=0, range_end=-1,
                n_forward=1, with_labels=False):
    """The reference's CustomDataLayer (custom_data_layer.cpp) over an in-memory stand-in for LMDB.  records: list of (key, value
    bytes) = serialized Datums.  Runs on the CPU path (the layer has no other).  Returns (tops, labels): tops[s] is
    [n_forward * batch, slice channels, H, W]."""
    n = len(records)
    keys = (C.c_char_p * n)(*[k.encode() if isinstance(k, str) else k for k, _ in records])
    bufs = [np.frombuffer(v, dtype=np.uint8) for _, v in records]
    vals = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    sp = (C.c_int * max(1, len(slice_points)))(*slice_points)
    en = (C.c_int * max(1, len(encodings)))(*encodings)
    sub = (C.c_float * max(1, len(subtract)))(*subtract)
    ntop = len(slice_points) + 1
    shapes = (C.c_int * (4 * ntop))()
    L = lib()
    L.fn2ref_custom_data.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                     C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float, C.POINTER(C.c_float), C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_int)]
    # the top shapes come from the first datum (custom_data_layer.cpp:527-536); size the outputs from the record itself
    from . import datum_parse
    d = datum_parse(records[0][1])
    bounds = [0] + list(slice_points) + [d["channels"]]
    tops = [np.empty((n_forward * batch_size, max(b - a, 0), d["height"], d["width"]), np.float32) for a, b in zip(bounds, bounds[1:])]
    ptrs = (C.c_void_p * ntop)(*[t.ctypes.data for t in tops])
    labels = np.empty(n_forward * batch_size, np.float32) if with_labels else None
    _chk(L.fn2ref_custom_data(keys, vals, lens, n, batch_size, sp, len(slice_points), en, len(encodings), scale, sub, len(subtract),
                              range_start, range_end, n_forward, ptrs, C.c_void_p(labels.ctypes.data) if with_labels else None, shapes))
    for s, t in enumerate(tops):
        assert tuple(shapes[4 * s:4 * s + 4]) == (batch_size,) + t.shape[1:], (tuple(shapes[4 * s:4 * s + 4]), t.shape)
    return tops, labels


def flow_augmentation(flow, coeffs1, coeffs2, crop_height, crop_width):
    """FlowAugmentationLayer of the reference (flow_augmentation_layer.cpp/.cu + augmentation_layer_base.cpp); GPU only."""
    flow = _f(flow)
    N, _, H, W = flow.shape
    c1, c2 = _f(coeffs1).reshape(N, -1), _f(coeffs2).reshape(N, -1)
    top = np.empty((N, 2, crop_height, crop_width), np.float32)
    _chk(lib().fn2ref_flow_augmentation(_p(flow), _p(c1), _p(c2), c1.shape[1], N, H, W, crop_height, crop_width, _p(top)))
    return top


def data_augmentation(bottom, coeffs=None, crop_height=0, crop_width=0, max_multiplier=255.0, chromatic_eigvec=None, mean3=None):
    """DataAugmentationLayer of the reference with the coefficient blob given as bottom[1] (or absent: defaults).  GPU only."""
    bottom = _f(bottom)
    N, Cc, H, W = bottom.shape
    crop = crop_width > 0 and crop_height > 0
    top = np.empty((N, Cc, crop_height if crop else H, crop_width if crop else W), np.float32)
    co = _f(coeffs).reshape(N, -1) if coeffs is not None else None
    ev = _f(chromatic_eigvec) if chromatic_eigvec is not None else None
    m3 = _f(mean3) if mean3 is not None else None
    _chk(lib().fn2ref_data_augmentation(_p(bottom), _p(co), co.shape[1] if co is not None else 42, N, Cc, H, W, crop_height, crop_width,
                                        C.c_float(max_multiplier), _p(ev), _p(m3), _p(top)))
    return top
