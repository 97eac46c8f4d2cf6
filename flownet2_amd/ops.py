"""Raw operator calls: torch CUDA tensors in, C ABI (libflownet2_hip.so) underneath.

torch is plumbing here (device memory + the current HIP stream); all arithmetic happens in the HIP
kernels.  Every function validates what the reference's Reshape would CHECK, then forwards to the
C entry point; there is no PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import CorrParams, L1LossParams, check

MULTIPLY, SUBTRACT = 0, 1
FILL_ZERO, FILL_NAN = 1, 2
NEAREST, LINEAR, CUBIC, AREA = 1, 2, 3, 4


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, ndim: int = 4):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA (HIP) tensor; flownet2_amd has no CPU path")
    if t.dtype != torch.float32:
        raise ValueError(f"{name}: expected float32 (Dtype=float), got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} axes (NCHW), got shape {tuple(t.shape)}")
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def corr_params(pad=0, kernel_size=1, max_displacement=0, stride_1=1, stride_2=1, correlation_type=MULTIPLY, do_abs=False,
                single_direction=0):
    return CorrParams(int(pad), int(kernel_size), int(max_displacement), int(stride_1), int(stride_2),
                      int(correlation_type), int(bool(do_abs)), int(single_direction))


def correlation_out_shape(p: CorrParams, Cc: int, H: int, W: int):
    tc, th, tw = C.c_int(), C.c_int(), C.c_int()
    check(_lib.lib().fn2_correlation_out_shape(C.byref(p), Cc, H, W, C.byref(tc), C.byref(th), C.byref(tw)))
    return tc.value, th.value, tw.value


def correlation_forward(p: CorrParams, bottom0: torch.Tensor, bottom1: torch.Tensor, out: torch.Tensor | None = None,
                        out_c0: int = 0, relu: bool = False, negative_slope: float = 0.0):
    """top = Correlation(bottom0, bottom1).  With `out` wider than topC channels the layer writes the slice
    [out_c0, out_c0 + topC) of it (the Concat that follows it in FlowNetC); relu applies ReLU{negative_slope} on the way out."""
    b0, b1 = _chk(bottom0, "bottom[0]"), _chk(bottom1, "bottom[1]")
    if b0.shape != b1.shape:   # correlation_layer.cpp:45-47
        raise ValueError("Both bottom blobs must have same shape")
    N, Cc, H, W = b0.shape
    tc, th, tw = correlation_out_shape(p, Cc, H, W)
    top = out if out is not None else torch.empty((N, tc, th, tw), device=b0.device, dtype=torch.float32)
    assert top.is_contiguous() and top.shape[0] == N and tuple(top.shape[2:]) == (th, tw) and out_c0 + tc <= top.shape[1]
    if not relu and out_c0 == 0 and top.shape[1] == tc:
        check(_lib.lib().fn2_correlation_forward(C.byref(p), _ptr(b0), _ptr(b1), _ptr(top), N, Cc, H, W, None, 0, _stream()))
    else:
        check(_lib.lib().fn2_correlation_forward_fused(C.byref(p), _ptr(b0), _ptr(b1), _ptr(top), N, Cc, H, W, int(top.shape[1]), int(out_c0),
                                                       int(bool(relu)), C.c_float(float(negative_slope)), None, 0, _stream()))
    return top


def correlation_backward(p: CorrParams, bottom0, bottom1, top_diff, need0=True, need1=True):
    b0, b1, td = _chk(bottom0, "bottom[0]"), _chk(bottom1, "bottom[1]"), _chk(top_diff, "top.diff")
    N, Cc, H, W = b0.shape
    d0 = torch.empty_like(b0) if need0 else None
    d1 = torch.empty_like(b1) if need1 else None
    check(_lib.lib().fn2_correlation_backward(C.byref(p), _ptr(b0), _ptr(b1), _ptr(td), _ptr(d0), _ptr(d1),
                                              N, Cc, H, W, None, 0, _stream()))
    return d0, d1


def correlation1d_out_shape(p: CorrParams, Cc: int, H: int, W: int):
    tc, th, tw = C.c_int(), C.c_int(), C.c_int()
    check(_lib.lib().fn2_correlation1d_out_shape(C.byref(p), Cc, H, W, C.byref(tc), C.byref(th), C.byref(tw)))
    return tc.value, th.value, tw.value


def correlation1d_forward(p: CorrParams, bottom0: torch.Tensor, bottom1: torch.Tensor):
    b0, b1 = _chk(bottom0, "bottom[0]"), _chk(bottom1, "bottom[1]")
    if b0.shape != b1.shape:   # correlation_layer1d.cpp:48-50
        raise ValueError("Both bottom blobs must have same shape")
    N, Cc, H, W = b0.shape
    tc, th, tw = correlation1d_out_shape(p, Cc, H, W)
    top = torch.empty((N, tc, th, tw), device=b0.device, dtype=torch.float32)
    check(_lib.lib().fn2_correlation1d_forward(C.byref(p), _ptr(b0), _ptr(b1), _ptr(top), N, Cc, H, W, _stream()))
    return top


def correlation1d_backward(p: CorrParams, bottom0, bottom1, top_diff, need0=True, need1=True):
    b0, b1, td = _chk(bottom0, "bottom[0]"), _chk(bottom1, "bottom[1]"), _chk(top_diff, "top.diff")
    N, Cc, H, W = b0.shape
    d0 = torch.empty_like(b0) if need0 else None
    d1 = torch.empty_like(b1) if need1 else None
    check(_lib.lib().fn2_correlation1d_backward(C.byref(p), _ptr(b0), _ptr(b1), _ptr(td), _ptr(d0), _ptr(d1), N, Cc, H, W, _stream()))
    return d0, d1


def set_correlation_impl(impl):
    """Test hook: 0 / False = automatic choice, 1 / True = generic kernels, 3 = the general (dword LDS-DMA) MFMA forward
    even where the paired-parity kernel applies."""
    check(_lib.lib().fn2_debug_set_correlation_impl(int(impl)))


def set_resample_generic(on):
    """Test hook: True = always the per-output-pixel Resample kernels (no integer-factor up-sampling path)."""
    check(_lib.lib().fn2_debug_set_resample_generic(int(bool(on))))


def flow_warp_forward(image, flow, fill_value=FILL_ZERO):
    im, fl = _chk(image, "bottom[0] (image)"), _chk(flow, "bottom[1] (flow)")
    N, Cc, H, W = im.shape
    if fl.shape[0] != N:
        raise ValueError("Num of the inputs should be the same")               # flow_warp_layer.cpp:45
    if fl.shape[1] != 2:
        raise ValueError("Flow should have 2 channels: x-flow and y-flow")     # :46
    if fl.shape[3] != W or fl.shape[2] != H:
        raise ValueError("Width/Height of the inputs should be the same")      # :47-48
    out = torch.empty_like(im)
    check(_lib.lib().fn2_flow_warp_forward(_ptr(im), _ptr(fl), _ptr(out), N, Cc, H, W, int(fill_value), _stream()))
    return out


def _as_slice(arg, what):
    """A blob argument of the *_slices entry points: a plain [N,C,H,W] tensor, or (blob, c0, C) = channels [c0, c0 + C) of a wider
    contiguous blob.  -> (blob, channels of the blob, c0, C)."""
    if isinstance(arg, tuple):
        blob, c0, Cc = arg
        if isinstance(blob, torch.Tensor) and not blob.is_contiguous():
            raise ValueError(what + ": a sliced blob must be contiguous")
        blob = _chk(blob, what)
        if c0 < 0 or Cc < 1 or c0 + Cc > blob.shape[1]:
            raise ValueError("%s: channel slice [%d, %d) outside a blob of %d channels" % (what, c0, c0 + Cc, blob.shape[1]))
        return blob, int(blob.shape[1]), int(c0), int(Cc)
    t = _chk(arg, what)
    return t, int(t.shape[1]), 0, int(t.shape[1])


def flow_warp_forward_slices(image, flow, out=None, fill_value=FILL_ZERO):
    """FlowWarp whose image bottom and top may be channel slices (blob, c0, C) of wider blobs (the Concat around it disappears).
    Returns the top blob (a new [N,C,H,W] one if `out` is None)."""
    im, ictot, ic0, Cc = _as_slice(image, "bottom[0] (image)")
    fl, fctot, fc0, Cf = _as_slice(flow, "bottom[1] (flow)")
    N, _, H, W = im.shape
    if fl.shape[0] != N or Cf != 2 or tuple(fl.shape[2:]) != (H, W):
        raise ValueError("flow_warp: flow must be [N,2,H,W] of the image's size")
    if out is None:
        out = torch.empty((N, Cc, H, W), device=im.device, dtype=torch.float32)
    top, octot, oc0, Co = _as_slice(out, "top[0]")
    if Co != Cc or top.shape[0] != N or tuple(top.shape[2:]) != (H, W):
        raise ValueError("flow_warp: top slice does not match the image")
    check(_lib.lib().fn2_flow_warp_forward_slices(_ptr(im), ictot, ic0, _ptr(fl), fctot, fc0, _ptr(top), octot, oc0, N, Cc, H, W, int(fill_value), _stream()))
    return top


def flow_warp_backward(image, flow, warped_diff, propagate_image=True, propagate_flow=True):
    im, fl, wd = _chk(image, "image"), _chk(flow, "flow"), _chk(warped_diff, "top.diff")
    N, Cc, H, W = im.shape
    di, df = torch.empty_like(im), torch.empty_like(fl)
    nbytes = _lib.lib().fn2_flow_warp_backward_workspace_bytes(N, Cc, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=im.device)
    check(_lib.lib().fn2_flow_warp_backward(_ptr(im), _ptr(fl), _ptr(wd), _ptr(di), _ptr(df), N, Cc, H, W,
                                            int(propagate_image), int(propagate_flow), _ptr(ws), nbytes, _stream()))
    return di, df


def resample_forward(x, height, width, type=LINEAR, antialias=True):
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    if height < 1 or width < 1:
        raise ValueError("ResampleLayer must have top_height > 0 and top_width > 0")
    out = torch.empty((N, Cc, int(height), int(width)), device=x.device, dtype=torch.float32)
    check(_lib.lib().fn2_resample_forward(_ptr(x), _ptr(out), N, Cc, H, W, int(height), int(width), int(type), int(bool(antialias)), _stream()))
    return out


def resample_forward_slices(x, height, width, type=LINEAR, antialias=True, in_scale=1.0, out=None, out2=None, out2_scale=1.0):
    """Resample(x * in_scale) into a channel slice `out` = (blob, c0, C) (a new blob if None); out2 (optional slice) = top * out2_scale.
    The Eltwise scalings and the Concat the FlowNet2 graphs put around the layer, without their passes.  Returns the `out` blob."""
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    if height < 1 or width < 1:
        raise ValueError("ResampleLayer must have top_height > 0 and top_width > 0")
    if out is None:
        out = torch.empty((N, Cc, int(height), int(width)), device=x.device, dtype=torch.float32)
    top, octot, oc0, Co = _as_slice(out, "top[0]")
    if Co != Cc or top.shape[0] != N or tuple(top.shape[2:]) != (int(height), int(width)):
        raise ValueError("resample: top slice does not match")
    t2, o2ctot, o2c0 = None, 0, 0
    if out2 is not None:
        t2, o2ctot, o2c0, C2 = _as_slice(out2, "top[1]")
        if C2 != Cc or t2.shape[0] != N or tuple(t2.shape[2:]) != (int(height), int(width)):
            raise ValueError("resample: second top slice does not match")
    check(_lib.lib().fn2_resample_forward_slices(_ptr(x), C.c_float(float(in_scale)), _ptr(top), octot, oc0, _ptr(t2), o2ctot, o2c0,
                                                 C.c_float(float(out2_scale)), N, Cc, H, W, int(height), int(width), int(type),
                                                 int(bool(antialias)), _stream()))
    return top


def l1_params(l2_per_location=False, l2_prescale_by_channels=False, normalize_by_num_entries=False, epsilon=1e-2, plateau=0.0):
    return L1LossParams(int(bool(l2_per_location)), int(bool(l2_prescale_by_channels)), int(bool(normalize_by_num_entries)),
                        float(epsilon), float(plateau))


def l1loss_workspace(x: torch.Tensor) -> torch.Tensor:
    N, Cc, H, W = x.shape
    nbytes = _lib.lib().fn2_l1loss_workspace_bytes(N, Cc, H, W)
    return torch.empty(nbytes, dtype=torch.uint8, device=x.device)


def l1loss_forward(p: L1LossParams, bottom0, bottom1=None, workspace=None):
    """Returns (loss [0-axis device tensor], workspace).  workspace[:8] viewed as float32 = {loss, normalize_coeff}."""
    b0 = _chk(bottom0, "bottom[0]")
    b1 = _chk(bottom1, "bottom[1]") if bottom1 is not None else None
    if b1 is not None and b1.shape != b0.shape:
        raise ValueError("L1Loss: bottom blobs must have the same shape")
    N, Cc, H, W = b0.shape
    ws = workspace if workspace is not None else l1loss_workspace(b0)
    loss = torch.empty((), device=b0.device, dtype=torch.float32)
    check(_lib.lib().fn2_l1loss_forward(C.byref(p), _ptr(b0), _ptr(b1), _ptr(loss), N, Cc, H, W, _ptr(ws), ws.numel(), _stream()))
    return loss, ws


def l1loss_backward(p: L1LossParams, bottom0, bottom1, top_diff: float, workspace):
    b0 = _chk(bottom0, "bottom[0]")
    b1 = _chk(bottom1, "bottom[1]") if bottom1 is not None else None
    N, Cc, H, W = b0.shape
    d0 = torch.empty_like(b0)
    d1 = torch.empty_like(b0) if b1 is not None else None
    check(_lib.lib().fn2_l1loss_backward(C.byref(p), _ptr(b0), _ptr(b1), C.c_float(float(top_diff)), _ptr(d0), _ptr(d1),
                                         N, Cc, H, W, _ptr(workspace), workspace.numel(), _stream()))
    return d0, d1


_L1_SYNC = {}       # (device, stream) -> the zeroed arrival counters of fn2_l1loss_forward_multi (every call leaves them zero)


def _l1_scales(bottoms0, bottoms1, weights, diffs0=None, diffs1=None):
    n = len(bottoms0)
    arr = (_lib.L1LossScale * n)()
    for k in range(n):
        b0 = _chk(bottoms0[k], "bottom[0]")
        b1 = _chk(bottoms1[k], "bottom[1]") if bottoms1 is not None and bottoms1[k] is not None else None
        if b1 is not None and b1.shape != b0.shape:
            raise ValueError("L1Loss: bottom blobs must have the same shape")
        N, Cc, H, W = b0.shape
        arr[k].bottom0, arr[k].bottom1 = b0.data_ptr(), (b1.data_ptr() if b1 is not None else None)
        arr[k].bottom0_diff = diffs0[k].data_ptr() if diffs0 is not None else None
        arr[k].bottom1_diff = diffs1[k].data_ptr() if diffs1 is not None and diffs1[k] is not None else None
        arr[k].N, arr[k].C, arr[k].H, arr[k].W = N, Cc, H, W
        arr[k].loss_weight = float(weights[k])
    return arr


def l1loss_forward_multi(p: L1LossParams, bottoms0, bottoms1, weights):
    """Every L1Loss layer of a net in one launch (csrc/l1loss.hip: l1loss_fwd_multi).  Returns (total [0-axis device tensor] =
    sum_k weights[k] * loss_k in list order, losses [n], workspace); per scale bit-identical to l1loss_forward."""
    n = len(bottoms0)
    dev = bottoms0[0].device
    ws = torch.empty(int(_lib.lib().fn2_l1loss_multi_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    key = (dev.index, int(torch.cuda.current_stream().cuda_stream))
    sync = _L1_SYNC.get(key)
    if sync is None:
        sync = _L1_SYNC[key] = torch.zeros(int(_lib.lib().fn2_l1loss_multi_sync_bytes()), dtype=torch.uint8, device=dev)
    losses = torch.empty((n,), device=dev, dtype=torch.float32)
    total = torch.empty((), device=dev, dtype=torch.float32)
    arr = _l1_scales(bottoms0, bottoms1, weights)
    check(_lib.lib().fn2_l1loss_forward_multi(C.byref(p), n, arr, _ptr(losses), _ptr(total), _ptr(ws), ws.numel(), _ptr(sync), _stream()))
    return total, losses, ws


def l1loss_backward_multi(p: L1LossParams, bottoms0, bottoms1, weights, total_diff, workspace, need1=False):
    """Gradients of every scale for d(objective) / d(total) = total_diff (a DEVICE scalar tensor, or None for 1) in one launch."""
    n = len(bottoms0)
    d0 = [torch.empty_like(b) for b in bottoms0]
    d1 = [torch.empty_like(b) if (need1 and bottoms1 is not None and bottoms1[k] is not None) else None for k, b in enumerate(bottoms0)]
    arr = _l1_scales(bottoms0, bottoms1, weights, d0, d1)
    td = _chk(total_diff.reshape(1), "total_diff", ndim=1) if total_diff is not None else None
    check(_lib.lib().fn2_l1loss_backward_multi(C.byref(p), n, arr, _ptr(td), _ptr(workspace), workspace.numel(), _stream()))
    return d0, d1


def channel_norm_forward(x):
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    out = torch.empty((N, 1, H, W), device=x.device, dtype=torch.float32)
    check(_lib.lib().fn2_channel_norm_forward(_ptr(x), _ptr(out), N, Cc, H, W, _stream()))
    return out


def channel_norm_forward_slices(x, minus=None, out=None):
    """ChannelNorm(x - minus) (minus optional) over channel slices (blob, c0, C); out = (blob, c0, 1) or None for a new [N,1,H,W] blob."""
    b, bctot, bc0, Cc = _as_slice(x, "bottom[0]")
    N, _, H, W = b.shape
    m, mctot, mc0 = None, 0, 0
    if minus is not None:
        m, mctot, mc0, Cm = _as_slice(minus, "bottom[1]")
        if Cm != Cc or tuple(m.shape[2:]) != (H, W) or m.shape[0] != N:
            raise ValueError("channel_norm: the subtrahend does not match the bottom")
    if out is None:
        out = torch.empty((N, 1, H, W), device=b.device, dtype=torch.float32)
    top, tctot, tc0, Ct = _as_slice(out, "top[0]")
    if Ct != 1 or top.shape[0] != N or tuple(top.shape[2:]) != (H, W):
        raise ValueError("channel_norm: top slice must be one channel of the bottom's size")
    check(_lib.lib().fn2_channel_norm_forward_slices(_ptr(b), bctot, bc0, _ptr(m), mctot, mc0, _ptr(top), tctot, tc0, N, Cc, H, W, _stream()))
    return top


def channel_norm_backward(x, top, top_diff):
    x, top, td = _chk(x, "bottom[0]"), _chk(top, "top"), _chk(top_diff, "top.diff")
    N, Cc, H, W = x.shape
    d = torch.empty_like(x)
    check(_lib.lib().fn2_channel_norm_backward(_ptr(x), _ptr(top), _ptr(td), _ptr(d), N, Cc, H, W, _stream()))
    return d


def downsample_forward(x, top_height, top_width):
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    if top_height < 1 or top_width < 1:
        raise ValueError("DownsampleLayer must have top_height > 0 and top_width > 0")
    out = torch.empty((N, Cc, int(top_height), int(top_width)), device=x.device, dtype=torch.float32)
    check(_lib.lib().fn2_downsample_forward(_ptr(x), _ptr(out), N, Cc, H, W, int(top_height), int(top_width), _stream()))
    return out


def downsample_forward_multi(x, sizes):
    """Downsample(x) to every (height, width) of `sizes` in ONE launch (fn2_downsample_forward_multi: up to 8 tops, each at least 2 x 2 and of
    another size than x); the bits of downsample_forward per size."""
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    sizes = [(int(h), int(w)) for h, w in sizes]
    outs = [torch.empty((N, Cc, h, w), device=x.device, dtype=torch.float32) for h, w in sizes]
    n = len(sizes)
    ptrs = (C.c_void_p * n)(*[_ptr(o) for o in outs])
    hs, ws = (C.c_int * n)(*[h for h, _ in sizes]), (C.c_int * n)(*[w for _, w in sizes])
    check(_lib.lib().fn2_downsample_forward_multi(_ptr(x), ptrs, hs, ws, n, N, Cc, H, W, _stream()))
    return outs


def downsample_multi_supported(x_shape, sizes) -> bool:
    H, W = int(x_shape[2]), int(x_shape[3])
    return 1 <= len(sizes) <= 8 and all(int(h) >= 2 and int(w) >= 2 and (int(h), int(w)) != (H, W) for h, w in sizes)


def predict_flow_conv_forward(x, weight, bias=None):
    """Convolution{kernel 3, stride 1, pad 1, num_output 2} (the FlowNet predict_flow heads)."""
    x, w = _chk(x, "bottom[0]"), _chk(weight, "weight")
    N, Cc, H, W = x.shape
    if tuple(w.shape) != (2, Cc, 3, 3):
        raise ValueError(f"predict_flow weight must be [2,{Cc},3,3], got {tuple(w.shape)}")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    out = torch.empty((N, 2, H, W), device=x.device, dtype=torch.float32)
    nbytes = _lib.lib().fn2_predict_flow_conv_workspace_bytes(N, Cc, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    check(_lib.lib().fn2_predict_flow_conv_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(out), N, Cc, H, W, _ptr(ws), nbytes, _stream()))
    return out


def upsample_flow_deconv_forward(x, weight, bias=None, out=None, out_c0=0):
    """Deconvolution{kernel 4, stride 2, pad 1, num_output 2} on a 2-channel flow (the upsample_flow heads); with `out` the two
    channels go to out[:, out_c0:out_c0+2] of a wider blob (the refinement Concat)."""
    x, w = _chk(x, "bottom[0]"), _chk(weight, "weight")
    N, Cc, H, W = x.shape
    if Cc != 2 or tuple(w.shape) != (2, 2, 4, 4):
        raise ValueError("upsample_flow expects a 2-channel flow and a [2,2,4,4] weight")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    if out is not None:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (2 * H, 2 * W):
            raise ValueError(f"upsample_flow: top blob {tuple(out.shape)} does not match [{N},*,{2 * H},{2 * W}]")
        check(_lib.lib().fn2_upsample_flow_deconv_forward_into(_ptr(x), _ptr(w), _ptr(b), _ptr(out), N, H, W, out.shape[1], int(out_c0), _stream()))
        return out[:, out_c0:out_c0 + 2]
    out = torch.empty((N, 2, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(_lib.lib().fn2_upsample_flow_deconv_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(out), N, H, W, _stream()))
    return out


def bias_leaky_relu_(x, bias=None, negative_slope=0.1):
    """In place: x[n,c] = leaky_relu(x[n,c] + bias[c]) -- the bias term + ReLU layer that follow every FlowNet conv/deconv."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("bias_leaky_relu_: expected a contiguous float32 CUDA (HIP) NCHW tensor")
    N, Cc, H, W = x.shape
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    if b is not None and b.numel() != Cc:
        raise ValueError(f"bias must have {Cc} entries, got {b.numel()}")
    check(_lib.lib().fn2_bias_leaky_relu_forward(_ptr(x), _ptr(b), N, Cc, H, W, C.c_float(float(negative_slope)), _stream()))
    return x


def scale_shift_forward(x, scale, shift=None, out=None, out_c0=0):
    """out[:, out_c0 : out_c0 + C] = x * scale + shift[c] with the product and the sum rounded separately (csrc/bias_act.hip): the deploy
    head's Eltwise{1/255} + mean subtraction in one pass, written into a channel slice of `out` [N, *, H, W] (a new blob if None)."""
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    if out is None:
        out = torch.empty_like(x)
    else:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (H, W):
            raise ValueError("scale_shift: top blob does not match the bottom")
    s = _chk(shift, "shift", ndim=1) if shift is not None else None
    check(_lib.lib().fn2_scale_shift_forward(_ptr(x), _ptr(out), _ptr(s), N, Cc, H, W, out.shape[1], out_c0, C.c_float(float(scale)), _stream()))
    return out


def predict_flow_conv_backward_supported(N, C, H, W) -> bool:
    return bool(_lib.lib().fn2_predict_flow_conv_backward_supported(int(N), int(C), int(H), int(W)))


def predict_flow_conv_backward(x, weight, top_diff, need_x=True, need_w=True, need_b=True):
    """Backward of predict_flow (Convolution{3,1,1} C -> 2): (bottom_diff, weight_diff, bias_diff), None where not needed.  `x` may be a
    channel slice (blob, c0, C)."""
    xb, xctot, xc0, Cc = _as_slice(x, "bottom[0]")
    w, g = _chk(weight, "weight"), _chk(top_diff, "top.diff")
    N, _, H, W = xb.shape
    if tuple(w.shape) != (2, Cc, 3, 3) or tuple(g.shape) != (N, 2, H, W):
        raise ValueError("predict_flow_conv_backward: weight must be [2,C,3,3] and top_diff [N,2,H,W]")
    dx = torch.empty((N, Cc, H, W), device=g.device, dtype=torch.float32) if need_x else None
    dw = torch.empty_like(w) if need_w else None
    db = torch.empty(2, device=g.device, dtype=torch.float32) if need_b else None
    nbytes = _lib.lib().fn2_predict_flow_conv_backward_workspace_bytes(N, Cc, H, W) if (need_w or need_b) else 0
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=g.device)
    check(_lib.lib().fn2_predict_flow_conv_backward(_ptr(xb), xctot, xc0, _ptr(w), _ptr(g), _ptr(dx), _ptr(dw), _ptr(db), N, Cc, H, W, 0,
                                                    _ptr(ws), nbytes, _stream()))
    return dx, dw, db


def upsample_flow_deconv_backward(x, weight, top_diff, need_x=True, need_w=True, need_b=True):
    """Backward of upsample_flow (Deconvolution{4,2,1} 2 -> 2): (bottom_diff, weight_diff, bias_diff), None where not needed."""
    x, w, g = _chk(x, "bottom[0]"), _chk(weight, "weight"), _chk(top_diff, "top.diff")
    N, Cc, H, W = x.shape
    if Cc != 2 or tuple(w.shape) != (2, 2, 4, 4) or tuple(g.shape) != (N, 2, 2 * H, 2 * W):
        raise ValueError("upsample_flow_deconv_backward: bottom [N,2,H,W], weight [2,2,4,4], top_diff [N,2,2H,2W]")
    dx = torch.empty_like(x) if need_x else None
    dw = torch.empty_like(w) if need_w else None
    db = torch.empty(2, device=g.device, dtype=torch.float32) if need_b else None
    nbytes = _lib.lib().fn2_upsample_flow_deconv_backward_workspace_bytes(N, H, W) if (need_w or need_b) else 0
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=g.device)
    check(_lib.lib().fn2_upsample_flow_deconv_backward(_ptr(x), _ptr(w), _ptr(g), _ptr(dx), _ptr(dw), _ptr(db), N, H, W, 0, _ptr(ws), nbytes, _stream()))
    return dx, dw, db


def conv_k7s2_relu_supported(Cin, Hin, Win, Cout) -> bool:
    return bool(_lib.lib().fn2_conv_k7s2_relu_supported(int(Cin), int(Hin), int(Win), int(Cout)))


def conv_k7s2_wgrad_supported(N, Cin, Hin, Win, Cout) -> bool:
    return bool(_lib.lib().fn2_conv_k7s2_wgrad_supported(int(N), int(Cin), int(Hin), int(Win), int(Cout)))


def conv_k7s2_wgrad_ksplit(N, Cin, Hin, Win, Cout) -> int:
    return int(_lib.lib().fn2_conv_k7s2_wgrad_ksplit(int(N), int(Cin), int(Hin), int(Win), int(Cout)))


def conv_k7s2_wgrad(top_diff, bottom):
    """weight_diff [Cout, Cin, 7, 7] of the 7x7 / 2 / 3 stem convolution (csrc/conv_stem_wgrad.hip); top_diff [N, 64, Ho, Wo], bottom [N, Cin, H, W]."""
    d, x = _chk(top_diff, "top.diff"), _chk(bottom, "bottom[0]")
    N, Cin, H, W = x.shape
    Cout = d.shape[1]
    if tuple(d.shape) != (N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1):
        raise ValueError(f"conv_k7s2_wgrad: top_diff {tuple(d.shape)} does not belong to a bottom of {tuple(x.shape)}")
    dw = torch.empty((Cout, Cin, 7, 7), device=x.device, dtype=torch.float32)
    need = int(_lib.lib().fn2_conv_k7s2_wgrad_workspace_bytes(N, Cin, H, W, Cout))
    ws = _plane_workspace(x.device, need) if need else None
    check(_lib.lib().fn2_conv_k7s2_wgrad(_ptr(d), _ptr(x), _ptr(dw), N, Cin, H, W, Cout, 0, _ptr(ws), need, _stream()))
    return dw


def conv_k7s2_relu_forward(x, weight, bias=None, negative_slope=0.1):
    """leaky_relu(Convolution{kernel 7, stride 2, pad 3}(x) + bias): conv1 + ReLU1 of the FlowNet encoders, one kernel."""
    x, w = _chk(x, "bottom[0]"), _chk(weight, "weight")
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 7, 7):
        raise ValueError(f"stem weight must be [Cout,{Cin},7,7], got {tuple(w.shape)}")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    out = torch.empty((N, Cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    check(_lib.lib().fn2_conv_k7s2_relu_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(out), N, Cin, H, W, Cout,
                                                C.c_float(float(negative_slope)), _stream()))
    return out


CONV_ROUTES = {0: None, 1: "direct", 2: "wino", 3: "plane", 4: None, 5: None}     # 4 / 5: stem / flow head -- the graphs call those kernels by name (conv_k7s2_relu, predict_flow_conv)
DECONV_ROUTES = {0: None, 1: "gemm", 2: "plane", 3: None}      # 3: the 2-channel upsample_flow head (upsample_flow_deconv)


def conv_route(N, Cin, Hin, Win, Cout, kernel, stride, pad, force=False):
    """Which kernel family the LIBRARY picks for Convolution{kernel, stride, pad} Cin -> Cout on [N, Cin, Hin, Win] (fn2_conv_route,
    csrc/conv_route.cpp -- the same decision the Caffe adapter gets): "wino", "plane", "direct" or None."""
    d = _lib.ConvDesc(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(kernel), int(stride), int(pad))
    return CONV_ROUTES[int(_lib.lib().fn2_conv_route(C.byref(d), 1 if force else 0))]


def deconv_route(N, Cin, Hin, Win, Cout, kernel=4, stride=2, pad=1):
    """fn2_deconv_route: "gemm" (weight^T x bottom on the 1x1 kernel + col2im), "plane" (parity classes, small maps) or None."""
    d = _lib.ConvDesc(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(kernel), int(stride), int(pad))
    return DECONV_ROUTES[int(_lib.lib().fn2_deconv_route(C.byref(d), 0))]


BWD_ROUTES = {0: None, 1: "wino", 2: "tconv", 3: "plane", 4: "direct", 5: "deconv_plane"}


def conv_desc(N, Cin, Hin, Win, Cout, kernel, stride, pad):
    """fn2_conv_desc: bottom [N, Cin, Hin, Win] of a Convolution{kernel, stride, pad} (Cin -> Cout) or of a Deconvolution{4, 2, 1}."""
    return _lib.ConvDesc(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(kernel), int(stride), int(pad))


def conv_backward_data_route(desc, transposed=False) -> int:
    """Which own kernel computes bottom_diff of this layer (fn2_conv_backward_data_route; 0 = none, names: BWD_ROUTES)."""
    return int(_lib.lib().fn2_conv_backward_data_route(C.byref(desc), int(bool(transposed))))


def conv_backward_data_pack_weights(weight, desc, transposed, route):
    """The layer's weight blob ([Cout, Cin, k, k]; Deconvolution: [Cin, Cout, 4, 4]) -> the operand its data-gradient kernel reads."""
    w = _chk(weight, "weight")
    L, tr = _lib.lib(), int(bool(transposed))
    n = int(L.fn2_conv_backward_data_packed_weight_floats(C.byref(desc), tr, int(route)))
    if n == 0:
        raise ValueError("conv_backward_data_pack_weights: route %d is not this layer's" % route)
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    need = int(L.fn2_conv_backward_data_pack_workspace_bytes(C.byref(desc), tr, int(route)))
    ws = torch.empty(need // 4, device=w.device, dtype=torch.float32) if need else None
    check(L.fn2_conv_backward_data_pack_weights(C.byref(desc), tr, int(route), _ptr(w), _ptr(packed), _ptr(ws), need, _stream()))
    return packed


def conv_backward_data(top_diff, packed_weight, desc, transposed, route, top_c0=0):
    """bottom_diff [N, Cin, Hin, Win] of the layer from top_diff (a blob, the layer's channels starting at top_c0).  The kernels compute
    channels in groups: the result is the leading-channel VIEW of a blob that has room for them (no copy)."""
    d = _chk(top_diff, "top_diff")
    L, tr = _lib.lib(), int(bool(transposed))
    Cp = int(L.fn2_conv_backward_data_computed_channels(C.byref(desc), tr, int(route)))
    if Cp == 0:
        raise ValueError("conv_backward_data: route %d is not this layer's" % route)
    out = torch.empty((desc.N, Cp, desc.Hin, desc.Win), device=d.device, dtype=torch.float32)
    need = int(L.fn2_conv_backward_data_workspace_bytes_with_room(C.byref(desc), tr, int(route), Cp))      # (the blob has room: no padded copy in the scratch)
    ws = _plane_workspace(d.device, need) if need else None
    check(L.fn2_conv_backward_data(C.byref(desc), tr, int(route), _ptr(d), d.shape[1], int(top_c0), _ptr(packed_weight), _ptr(out), Cp, 0, Cp,
                                   _ptr(ws), need, _stream()))
    return out[:, :desc.Cin] if Cp != desc.Cin else out


def conv_backward_weights_supported(desc, transposed=False) -> bool:
    return bool(_lib.lib().fn2_conv_backward_weights_supported(C.byref(desc), int(bool(transposed))))


def conv_backward_weights(bottom, top_diff, desc, transposed=False, out=None, accumulate=False, bottom_c0=0, top_c0=0):
    """weight_diff of the layer ([Cout, Cin, k, k]; Deconvolution: [Cin, Cout, 4, 4]) (+)= ... (fn2_conv_backward_weights: the stem kernel or
    csrc/conv_wgrad.hip, deterministic)."""
    x, d = _chk(bottom, "bottom"), _chk(top_diff, "top_diff")
    L, tr = _lib.lib(), int(bool(transposed))
    k = desc.kernel
    shape = (desc.Cin, desc.Cout, k, k) if transposed else (desc.Cout, desc.Cin, k, k)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
        accumulate = False
    else:
        out = _chk(out, "weight diff")
        if tuple(out.shape) != shape:
            raise ValueError("conv_backward_weights: weight diff has the wrong shape")
    need = int(L.fn2_conv_backward_weights_workspace_bytes(C.byref(desc), tr))
    ws = _plane_workspace(x.device, need) if need else None
    check(L.fn2_conv_backward_weights(C.byref(desc), tr, _ptr(x), x.shape[1], int(bottom_c0), _ptr(d), d.shape[1], int(top_c0), _ptr(out),
                                      int(bool(accumulate)), _ptr(ws), need, _stream()))
    return out


def conv_backward_data_masked_supported(desc, transposed, route) -> bool:
    return bool(_lib.lib().fn2_conv_backward_data_masked_supported(C.byref(desc), int(bool(transposed)), int(route)))


def conv_backward_data_masked(top_diff, packed, desc, transposed, route, bottom_data, negative_slope, top_c0=0, data_c0=0):
    """bottom_diff = (weight^T x top_diff) * leaky_relu'(bottom_data): the data gradient with ReLUBackward of the layer in front folded into
    the kernel's epilogue (fn2_conv_backward_data_masked; the transposed-convolution route).  bottom_data = this layer's bottom blob, the
    activated output of the layer in front (may be a channel slice: data_c0)."""
    d, y = _chk(top_diff, "top_diff"), _chk(bottom_data, "bottom data")
    out = torch.empty((desc.N, desc.Cin, desc.Hin, desc.Win), device=d.device, dtype=torch.float32)
    check(_lib.lib().fn2_conv_backward_data_masked(C.byref(desc), int(bool(transposed)), int(route), _ptr(d), d.shape[1], int(top_c0), _ptr(packed),
                                                   _ptr(out), desc.Cin, 0, _ptr(y), y.shape[1], int(data_c0), C.c_float(float(negative_slope)), _stream()))
    return out


def conv_backward_weights_bias_fused(desc, transposed=False) -> bool:
    return bool(_lib.lib().fn2_conv_backward_weights_bias_fused(C.byref(desc), int(bool(transposed))))


def conv_backward_weights_bias(bottom, top_diff, desc, transposed=False, out=None):
    """(weight_diff, bias_diff) of a layer from one pass over top_diff (fn2_conv_backward_weights_bias: the stem kernel sums the operand it
    feeds to the matrix pipe).  `out`: write weight_diff there (a gradient-bucket slot)."""
    x, d = _chk(bottom, "bottom"), _chk(top_diff, "top_diff")
    L, tr = _lib.lib(), int(bool(transposed))
    k = desc.kernel
    dw = out if out is not None else torch.empty((desc.Cout, desc.Cin, k, k), device=x.device, dtype=torch.float32)
    db = torch.empty(desc.Cout, device=x.device, dtype=torch.float32)
    need = int(L.fn2_conv_backward_weights_workspace_bytes(C.byref(desc), tr))
    ws = _plane_workspace(x.device, need) if need else None
    check(L.fn2_conv_backward_weights_bias(C.byref(desc), tr, _ptr(x), _ptr(d), _ptr(dw), _ptr(db), 0, _ptr(ws), need, _stream()))
    return dw, db


def conv_backward_bias(top_diff, C_, top_c0=0, out=None, accumulate=False):
    """bias_diff[c] (+)= sum over n, y, x of top_diff[n, top_c0 + c] (backward_gpu_bias, base_conv_layer.cpp:389-393), fixed order."""
    d = _chk(top_diff, "top_diff")
    N, Ctot, H, W = d.shape
    if out is None:
        out = torch.empty(C_, device=d.device, dtype=torch.float32)
        accumulate = False
    need = int(_lib.lib().fn2_bias_leaky_relu_backward_workspace_bytes(N, C_, H, W))
    ws = _plane_workspace(d.device, need)
    check(_lib.lib().fn2_conv_backward_bias(_ptr(d), Ctot, int(top_c0), _ptr(out), N, int(C_), H, W, int(bool(accumulate)), _ptr(ws), need, _stream()))
    return out


def conv_mfma_supported(Cin, Hin, Win, Cout, kernel, stride, pad) -> bool:
    return bool(_lib.lib().fn2_conv_mfma_supported(int(Cin), int(Hin), int(Win), int(Cout), int(kernel), int(stride), int(pad)))


def conv_mfma_pack_weights(weight):
    """weight [Cout, Cin, k, k] -> the MFMA operand order fn2_conv_mfma_forward reads (once per weight update)."""
    w = _chk(weight, "weight")
    Cout, Cin, k, k2 = w.shape
    n = _lib.lib().fn2_conv_mfma_packed_floats(Cout, Cin, k)
    if k != k2 or n == 0:
        raise ValueError(f"conv_mfma: unsupported weight shape {tuple(w.shape)}")
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().fn2_conv_mfma_pack_weights(_ptr(w), _ptr(packed), Cout, Cin, k, _stream()))
    return packed


def conv_mfma_pack_weights_view(blob, Cout, Cin, kernel, src_cout, src_cin, stride_cout, stride_cin, flip=False):
    """The operand fn2_conv_mfma_forward reads for the strided [Cout][Cin][k][k] VIEW of a contiguous weight blob: element (co, ci, tap) =
    blob.flatten()[co * stride_cout + ci * stride_cin + (k*k - 1 - tap if flip else tap)] for co < src_cout, ci < src_cin, 0 beyond --
    channel-swapped, rotated and zero-padded operands in one launch (csrc/conv_mfma.hip: pack_weights_view)."""
    w = _chk(blob, "weight", ndim=blob.dim())
    if not w.is_contiguous():
        raise ValueError("conv_mfma_pack_weights_view: the blob itself must be contiguous (the view is described by the strides)")
    n = _lib.lib().fn2_conv_mfma_packed_floats(int(Cout), int(Cin), int(kernel))
    if n == 0:
        raise ValueError(f"conv_mfma: unsupported operand [{Cout},{Cin},{kernel},{kernel}]")
    last = (src_cout - 1) * stride_cout + (src_cin - 1) * stride_cin + kernel * kernel - 1
    if last >= w.numel():
        raise ValueError("conv_mfma_pack_weights_view: the view reaches beyond the blob")
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().fn2_conv_mfma_pack_weights_view(_ptr(w), _ptr(packed), int(Cout), int(Cin), int(kernel), int(src_cout), int(src_cin),
                                                     int(stride_cout), int(stride_cin), int(bool(flip)), _stream()))
    return packed


def conv_mfma_forward(x, packed_weight, bias, Cout, kernel, stride, pad, relu=True, negative_slope=0.1, out=None, out_c0=0,
                      in_c0=0, Cin=None):
    """act(Convolution{kernel, stride, pad}(x[:, in_c0:in_c0+Cin]) + bias) -> out[:, out_c0:out_c0+Cout] (a new blob if out is None)."""
    x = _chk(x, "bottom[0]")
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Ho, Wo = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (Ho, Wo):
            raise ValueError(f"conv_mfma: top blob {tuple(out.shape)} does not match [{N},*,{Ho},{Wo}]")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    pw = _chk(packed_weight, "packed weight", ndim=1)
    check(_lib.lib().fn2_conv_mfma_forward(_ptr(x), _ptr(pw), _ptr(b), _ptr(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                           kernel, stride, pad, int(bool(relu)), C.c_float(float(negative_slope)), _stream()))
    return out


def set_conv_variant(v: int):
    check(_lib.lib().fn2_debug_set_conv_variant(int(v)))


def conv_num_variants() -> int:
    return int(_lib.lib().fn2_conv_mfma_num_variants())


def conv_wino_supported(Cin, Hin, Win, Cout, pad) -> bool:
    return bool(_lib.lib().fn2_conv_wino_supported(int(Cin), int(Hin), int(Win), int(Cout), int(pad)))


def conv_wino_pack_weights(weight):
    """weight [Cout, Cin, 3, 3] -> U = G g G^T in MFMA operand order (once per weight update)."""
    w = _chk(weight, "weight")
    Cout, Cin, k, k2 = w.shape
    n = _lib.lib().fn2_conv_wino_packed_floats(Cout, Cin)
    if k != 3 or k2 != 3 or n == 0:
        raise ValueError(f"conv_wino: unsupported weight shape {tuple(w.shape)}")
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().fn2_conv_wino_pack_weights(_ptr(w), _ptr(packed), Cout, Cin, _stream()))
    return packed


def conv_wino_forward(x, packed_weight, bias, Cout, pad=1, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    """act(Convolution{3, 1, pad}(x[:, in_c0:in_c0+Cin]) + bias) -> out[:, out_c0:out_c0+Cout], Winograd F(2x2, 3x3)."""
    x = _chk(x, "bottom[0]")
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (Ho, Wo):
            raise ValueError(f"conv_wino: top blob {tuple(out.shape)} does not match [{N},*,{Ho},{Wo}]")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    pw = _chk(packed_weight, "packed weight", ndim=1)
    check(_lib.lib().fn2_conv_wino_forward(_ptr(x), _ptr(pw), _ptr(b), _ptr(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                           pad, int(bool(relu)), C.c_float(float(negative_slope)), _stream()))
    return out


def conv_plane_supported(N, Cin, Hin, Win, Cout, stride, pad) -> bool:
    return bool(_lib.lib().fn2_conv_plane_supported(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(stride), int(pad)))


def conv_plane_ksplit(N, Cin, Hin, Win, Cout, stride, pad) -> int:
    return int(_lib.lib().fn2_conv_plane_ksplit(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(stride), int(pad)))


_PLANE_WS = {}      # (device, stream) -> the split-K partial-sum workspace (grown on demand, reused by every layer: launches are stream-ordered)
_PLANE_WS_RETIRED = []   # outgrown workspaces are kept alive: a hipGraph captured earlier has their address baked into its kernels


def _plane_workspace(device, need):
    """Per (device, stream) scratch of at least `need` bytes.  Never freed or shrunk: a regrow keeps the old block alive
    (_PLANE_WS_RETIRED), so a graph captured with the old pointer keeps replaying into memory nobody else owns; separate streams get
    separate blocks (no ordering exists between them)."""
    key = (device, int(torch.cuda.current_stream(device).cuda_stream))
    ws = _PLANE_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            _PLANE_WS_RETIRED.append(ws)
        ws = _PLANE_WS[key] = torch.empty((need + 3) // 4, device=device, dtype=torch.float32)
    return ws


def conv_plane_k_supported(N, Cin, Hin, Win, Cout, kernel, stride, pad) -> bool:
    return bool(_lib.lib().fn2_conv_plane_k_supported(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(kernel), int(stride), int(pad)))


def conv_plane_k_ksplit(N, Cin, Hin, Win, Cout, kernel, stride, pad) -> int:
    return int(_lib.lib().fn2_conv_plane_k_ksplit(int(N), int(Cin), int(Hin), int(Win), int(Cout), int(kernel), int(stride), int(pad)))


def conv_plane_forward(x, packed_weight, bias, Cout, stride, pad, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None, kernel=3):
    """act(Convolution{kernel, stride, pad}(x[:, in_c0:in_c0+Cin]) + bias) -> out[:, out_c0:out_c0+Cout] for small feature maps
    (csrc/conv_plane.hip); kernel 3, 4 with stride 2 / pad 1, or 5 with stride 2 / pad 2; packed_weight = conv_mfma_pack_weights(weight)."""
    x = _chk(x, "bottom[0]")
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Ho, Wo = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (Ho, Wo):
            raise ValueError(f"conv_plane: top blob {tuple(out.shape)} does not match [{N},*,{Ho},{Wo}]")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    pw = _chk(packed_weight, "packed weight", ndim=1)
    need = int(_lib.lib().fn2_conv_plane_k_workspace_bytes(N, Cin, H, W, Cout, kernel, stride, pad))
    ws = _plane_workspace(x.device, need) if need else None
    check(_lib.lib().fn2_conv_plane_k_forward(_ptr(x), _ptr(pw), _ptr(b), _ptr(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                              kernel, stride, pad, int(bool(relu)), C.c_float(float(negative_slope)),
                                              _ptr(ws if need else None), need, _stream()))
    return out


def tconv_supported(Cin, Hin, Win, Cout, Hout, Wout, kernel, pad) -> bool:
    return bool(_lib.lib().fn2_tconv_supported(int(Cin), int(Hin), int(Win), int(Cout), int(Hout), int(Wout), int(kernel), int(pad)))


def tconv_pack_weights(weight):
    """weight [Cin, Cout, k, k] (Caffe's Deconvolution blob; for a data gradient the Convolution's own [Cout_conv, Cin_conv, k, k] blob)
    -> the MFMA operand order fn2_tconv_forward reads: fn2_conv_mfma_pack_weights of the [Cout][Cin][k][k] view."""
    w = weight.detach().contiguous()
    A, B, k, _ = w.shape                                    # operand [Cout = B][Cin = A][k][k]: the blob with its channel axes swapped
    return conv_mfma_pack_weights_view(w, B, A, k, B, A, k * k, B * k * k)


def tconv_forward(x, packed_weight, bias, Cout, kernel, pad, out_hw=None, relu=False, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    """Transposed convolution, stride 2 (csrc/tconv_mfma.hip): the Deconvolution forward (+ bias + ReLU) or a stride-2 Convolution's
    data gradient.  out_hw: output size (default 2 (H - 1) + kernel - 2 pad)."""
    x = _chk(x, "bottom[0]")
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    Ho, Wo = out_hw if out_hw is not None else (2 * (H - 1) + kernel - 2 * pad, 2 * (W - 1) + kernel - 2 * pad)
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (Ho, Wo):
            raise ValueError(f"tconv: top blob {tuple(out.shape)} does not match [{N},*,{Ho},{Wo}]")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    pw = _chk(packed_weight, "packed weight", ndim=1)
    check(_lib.lib().fn2_tconv_forward(_ptr(x), _ptr(pw), _ptr(b), _ptr(out), N, Cin, H, W, Ctot, in_c0, Cout, Ho, Wo, out.shape[1], out_c0,
                                       int(kernel), int(pad), int(bool(relu)), C.c_float(float(negative_slope)), _stream()))
    return out


def set_tconv_variant(v: int):
    check(_lib.lib().fn2_debug_set_tconv_variant(int(v)))


def tconv_num_variants() -> int:
    return int(_lib.lib().fn2_tconv_num_variants())


def conv_wgrad_supported(N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad) -> bool:
    return bool(_lib.lib().fn2_conv_wgrad_supported(int(N), int(Ca), int(Ha), int(Wa), int(Cb), int(Hb), int(Wb), int(kernel), int(stride), int(pad)))


def conv_wgrad_ksplit(N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad) -> int:
    return int(_lib.lib().fn2_conv_wgrad_ksplit(int(N), int(Ca), int(Ha), int(Wa), int(Cb), int(Hb), int(Wb), int(kernel), int(stride), int(pad)))


def conv_wgrad(a, b, kernel, stride, pad, out=None, accumulate=False, a_c0=0, Ca=None, b_c0=0, Cb=None):
    """dw[ca][cb][ky][kx] (+)= sum a[n, a_c0 + ca, y, x] * b[n, b_c0 + cb, stride y + ky - pad, stride x + kx - pad] (csrc/conv_wgrad.hip).
    Convolution: a = top_diff, b = bottom -> [Cout, Cin, k, k]; Deconvolution: a = bottom, b = top_diff -> [Cin, Cout, k, k]."""
    a, b = _chk(a, "a"), _chk(b, "b")
    N, Atot, Ha, Wa = a.shape
    Nb, Btot, Hb, Wb = b.shape
    if N != Nb:
        raise ValueError("conv_wgrad: batch sizes differ")
    Ca = Atot - a_c0 if Ca is None else Ca
    Cb = Btot - b_c0 if Cb is None else Cb
    if out is None:
        out = torch.empty((Ca, Cb, kernel, kernel), device=a.device, dtype=torch.float32)
        accumulate = False
    else:
        out = _chk(out, "weight diff")
        if tuple(out.shape) != (Ca, Cb, kernel, kernel):
            raise ValueError("conv_wgrad: weight diff has the wrong shape")
    need = int(_lib.lib().fn2_conv_wgrad_workspace_bytes(N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad))
    ws = _plane_workspace(a.device, need) if need else None
    check(_lib.lib().fn2_conv_wgrad(_ptr(a), _ptr(b), _ptr(out), N, Ca, Ha, Wa, Atot, a_c0, Cb, Hb, Wb, Btot, b_c0, int(kernel), int(stride), int(pad),
                                    int(bool(accumulate)), _ptr(ws), need, _stream()))
    return out


def deconv_plane_supported(N, Cin, Hin, Win, Cout) -> bool:
    return bool(_lib.lib().fn2_deconv_plane_supported(int(N), int(Cin), int(Hin), int(Win), int(Cout)))


def deconv_plane_ksplit(N, Cin, Hin, Win, Cout) -> int:
    return int(_lib.lib().fn2_deconv_plane_ksplit(int(N), int(Cin), int(Hin), int(Win), int(Cout)))


def deconv_plane_pack_weights(weight):
    """weight [Cin, Cout, 4, 4] (Caffe's deconvolution blob) -> per-parity-class MFMA operand order (once per weight update).  A
    [Cin, Cout, 3, 3] blob is read as the 4x4 one whose fourth tap row and column are zero (the transposed 3x3 / 2 / 1 convolution of a
    data gradient on a small map), without a padded copy."""
    w = _chk(weight, "weight")
    Cin, Cout, k, k2 = w.shape
    n = _lib.lib().fn2_deconv_plane_packed_floats(Cin, Cout)
    if k not in (3, 4) or k2 != k or n == 0:
        raise ValueError(f"deconv_plane: unsupported weight shape {tuple(w.shape)}")
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().fn2_deconv_plane_pack_weights_k(_ptr(w), _ptr(packed), Cin, Cout, k, _stream()))
    return packed


def deconv_plane_forward(x, packed_weight, bias, Cout, relu=True, negative_slope=0.1, out=None, out_c0=0, in_c0=0, Cin=None):
    """act(Deconvolution{4, 2, 1}(x[:, in_c0:in_c0+Cin]) + bias) -> out[:, out_c0:out_c0+Cout] (csrc/conv_plane.hip, MODE 1)."""
    x = _chk(x, "bottom[0]")
    N, Ctot, H, W = x.shape
    Cin = Ctot - in_c0 if Cin is None else Cin
    if out is None:
        out = torch.empty((N, Cout, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (2 * H, 2 * W):
            raise ValueError(f"deconv_plane: top blob {tuple(out.shape)} does not match [{N},*,{2 * H},{2 * W}]")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    pw = _chk(packed_weight, "packed weight", ndim=1)
    need = int(_lib.lib().fn2_deconv_plane_workspace_bytes(N, Cin, H, W, Cout))
    ws = _plane_workspace(x.device, need) if need else None
    check(_lib.lib().fn2_deconv_plane_forward(_ptr(x), _ptr(pw), _ptr(b), _ptr(out), N, Cin, H, W, Ctot, in_c0, Cout, out.shape[1], out_c0,
                                              int(bool(relu)), C.c_float(float(negative_slope)), _ptr(ws if need else None), need, _stream()))
    return out


def set_plane_variant(v: int):
    check(_lib.lib().fn2_debug_set_plane_variant(int(v)))


def set_plane_ksplit(k: int):
    check(_lib.lib().fn2_debug_set_plane_ksplit(int(k)))


def set_batch_invariant(on: bool):
    """fn2_set_batch_invariant: summation orders that would depend on N are computed for a batch of one sample."""
    check(_lib.lib().fn2_set_batch_invariant(1 if on else 0))


def get_batch_invariant() -> bool:
    return bool(_lib.lib().fn2_get_batch_invariant())


def plane_num_variants() -> int:
    return int(_lib.lib().fn2_conv_plane_num_variants())


def set_wino_variant(v: int):
    check(_lib.lib().fn2_debug_set_wino_variant(int(v)))


def wino_num_variants() -> int:
    return int(_lib.lib().fn2_conv_wino_num_variants())


def im2col_forward(x, kernel, pad, stride):
    """[N,C,H,W] -> col [N, C*k*k, Hc*Wc] (Caffe's im2col row order), batched."""
    x = _chk(x, "bottom[0]")
    N, Cc, H, W = x.shape
    Hc, Wc = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    col = torch.empty((N, Cc * kernel * kernel, Hc * Wc), device=x.device, dtype=torch.float32)
    check(_lib.lib().fn2_im2col_forward(_ptr(x), _ptr(col), N, Cc, H, W, int(kernel), int(pad), int(stride), _stream()))
    return col


def col2im_bias_relu_forward(col, bias, N, Cc, H, W, kernel, pad, stride, relu=True, negative_slope=0.1, out=None, out_c0=0):
    """col [N, C*k*k, Hc*Wc] -> image [N,C,H,W] (+ bias[c], optional leaky ReLU): the tail of a Deconvolution forward; with `out` the
    image goes to out[:, out_c0:out_c0+C] of a wider blob."""
    if not (col.is_cuda and col.dtype == torch.float32 and col.is_contiguous()):
        raise ValueError("col2im: expected a contiguous float32 CUDA (HIP) column blob")
    Hc, Wc = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    if col.numel() != N * Cc * kernel * kernel * Hc * Wc:
        raise ValueError(f"col2im: column blob has {col.numel()} entries, expected {N * Cc * kernel * kernel * Hc * Wc}")
    b = _chk(bias, "bias", ndim=1) if bias is not None else None
    if out is not None:
        _chk(out, "top[0]")
        if out.shape[0] != N or tuple(out.shape[2:]) != (H, W):
            raise ValueError(f"col2im: top blob {tuple(out.shape)} does not match [{N},*,{H},{W}]")
        check(_lib.lib().fn2_col2im_bias_relu_forward_into(_ptr(col), _ptr(b), _ptr(out), N, Cc, H, W, int(kernel), int(pad), int(stride),
                                                           int(bool(relu)), C.c_float(float(negative_slope)), out.shape[1], int(out_c0), _stream()))
        return out[:, out_c0:out_c0 + Cc]
    out = torch.empty((N, Cc, H, W), device=col.device, dtype=torch.float32)
    check(_lib.lib().fn2_col2im_bias_relu_forward(_ptr(col), _ptr(b), _ptr(out), N, Cc, H, W, int(kernel), int(pad), int(stride),
                                                  int(bool(relu)), C.c_float(float(negative_slope)), _stream()))
    return out


def bias_leaky_relu_backward(top_data, top_diff, negative_slope=0.1, need_bias_diff=True):
    """(bottom_diff, bias_diff): gradient of x -> leaky_relu(x + bias) given the OUTPUT blob and its gradient.  Either may be a channel slice
    (blob, c0, C) of a wider blob -- the gradient a Concat hands its bottoms; an output that was written straight into its consumer's Concat
    blob -- and is then read in place."""
    y, yctot, yc0, Cc = _as_slice(top_data, "top.data")
    g, gctot, gc0, gC = _as_slice(top_diff, "top.diff")
    N, _, H, W = y.shape
    if gC != Cc or g.shape[0] != N or tuple(g.shape[2:]) != (H, W):
        raise ValueError("top.data and top.diff must have the same shape")
    d = torch.empty((N, Cc, H, W), device=y.device, dtype=torch.float32)
    db = torch.empty(Cc, device=y.device, dtype=torch.float32) if need_bias_diff else None
    nbytes = _lib.lib().fn2_bias_leaky_relu_backward_workspace_bytes(N, Cc, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=y.device)
    check(_lib.lib().fn2_bias_leaky_relu_backward_slices2(_ptr(y), yctot, yc0, _ptr(g), gctot, gc0, _ptr(d), _ptr(db), N, Cc, H, W,
                                                          C.c_float(float(negative_slope)), _ptr(ws), nbytes, _stream()))
    return d, db


AUG_NUM_PARAMS = 42      # AugmentationCoeff fields, caffe.proto:436-486


def augmentation_matrix(coeffs, crop_width, crop_height, bottom_width, bottom_height, invert=False):
    """HOST: one coefficient array (coeff_to_array layout) -> (t0, t1, t2, t3, t4, t5) of tTransMat::fromCoeff (optionally inverted)."""
    import numpy as np
    c = np.ascontiguousarray(coeffs, np.float32)
    if c.shape != (AUG_NUM_PARAMS,):
        raise ValueError(f"coefficient array must have {AUG_NUM_PARAMS} entries")
    out = np.empty(6, np.float32)
    check(_lib.lib().fn2_augmentation_matrix(C.c_void_p(c.ctypes.data), crop_width, crop_height, bottom_width, bottom_height, int(invert),
                                             C.c_void_p(out.ctypes.data)))
    return out


def flow_augmentation_forward(flow, coeffs1, coeffs2, crop_height, crop_width):
    """flow: CUDA [N,2,H,W]; coeffs1 / coeffs2: [N,42] coefficient arrays of image 1 / image 2 (read on the host, like the reference's
    cpu_data(), flow_augmentation_layer.cu:123-124).  Returns [N,2,crop_height,crop_width]."""
    import numpy as np
    fl = _chk(flow, "bottom[0] (flow)")
    N, Cc, H, W = fl.shape
    if Cc != 2:
        raise ValueError("Flow data must have two channels")                      # flow_augmentation_layer.cpp:52
    host = []
    for c in (coeffs1, coeffs2):
        a = c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c)
        a = np.ascontiguousarray(a, np.float32).reshape(N, -1)
        if a.shape[1] != AUG_NUM_PARAMS:
            raise ValueError(f"coefficient blobs must hold {AUG_NUM_PARAMS} values per sample")
        host.append(a)
    top = torch.empty((N, 2, int(crop_height), int(crop_width)), device=fl.device, dtype=torch.float32)
    check(_lib.lib().fn2_flow_augmentation_forward(_ptr(fl), C.c_void_p(host[0].ctypes.data), C.c_void_p(host[1].ctypes.data), _ptr(top),
                                                   N, H, W, int(crop_height), int(crop_width), _stream()))
    return top


MEAN_NONE, MEAN_PER_CHANNEL, MEAN_PER_PIXEL = 0, 1, 2


def data_aug_params(crop_width=0, crop_height=0, max_multiplier=255.0, chromatic_eigvec=None, mean_mode=MEAN_NONE, noise_seed=0, noise_stream=0):
    p = _lib.DataAugParams(int(crop_width), int(crop_height), float(max_multiplier), int(chromatic_eigvec is not None))
    p.noise_seed, p.noise_stream = int(noise_seed), int(noise_stream)
    if chromatic_eigvec is not None:
        if len(chromatic_eigvec) != 9:
            raise ValueError("chromatic_eigvec must have 9 entries")
        p.chromatic_eigvec = (C.c_float * 9)(*[float(v) for v in chromatic_eigvec])
    p.mean_mode = int(mean_mode)
    return p


def data_augmentation_forward(p, bottom, coeffs=None, mean=None):
    """bottom: CUDA [N,C,H,W]; coeffs: [N,42] coefficient arrays (host side, like the reference's cpu_data()) or None = defaults;
    mean: CUDA tensor of C (per channel) or C*crop_h*crop_w (per pixel) values, as p.mean_mode says.  Returns [N,C,crop_h,crop_w]."""
    import numpy as np
    x = _chk(bottom, "bottom[0]")
    N, Cc, H, W = x.shape
    crop = p.crop_width > 0 and p.crop_height > 0
    ch, cw = (p.crop_height, p.crop_width) if crop else (H, W)
    host = None
    if coeffs is not None:
        host = coeffs.detach().cpu().numpy() if isinstance(coeffs, torch.Tensor) else np.asarray(coeffs)
        host = np.ascontiguousarray(host, np.float32).reshape(N, -1)
        if host.shape[1] != AUG_NUM_PARAMS:
            raise ValueError(f"coefficient blob must hold {AUG_NUM_PARAMS} values per sample")
    m = None
    if p.mean_mode != MEAN_NONE:
        m = _chk(mean, "mean", ndim=None)
        want = Cc if p.mean_mode == MEAN_PER_CHANNEL else Cc * ch * cw
        if m.numel() != want:
            raise ValueError(f"mean: expected {want} values")
    top = torch.empty((N, Cc, max(ch, 0), max(cw, 0)), device=x.device, dtype=torch.float32)
    nws = _lib.lib().fn2_data_augmentation_workspace_bytes(N)
    ws = torch.empty(nws, device=x.device, dtype=torch.uint8)
    check(_lib.lib().fn2_data_augmentation_forward(C.byref(p), _ptr(x), C.c_void_p(host.ctypes.data) if host is not None else None, _ptr(m),
                                                   _ptr(top), N, Cc, H, W, _ptr(ws), nws, _stream()))
    return top
