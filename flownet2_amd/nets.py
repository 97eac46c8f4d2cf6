"""FlowNetC / FlowNetS graphs as the reference's deploy prototxts wire them.

TOPOLOGY CAVEAT: the prototxt templates are NOT in the reference tree (models/download-models.sh:3-10
fetches them); the graphs below restate them from the FlowNet / FlowNet2 papers and the layer names
third-party converters use.  Every topology assumption lives in this file.

The custom layers (Correlation, Resample, FlowWarp, ChannelNorm, L1Loss, Downsample) go through
`backend` = flownet2_amd.functional (HIP kernels).  Convolution / Deconvolution / ReLU / Eltwise /
Concat are the stock Caffe layers of the reference (conv_layer.cpp:8-40, deconv_layer.cpp:8-45,
relu_layer.cu:8-14): dense fp32 contractions executed by MIOpen through torch (conv2d /
conv_transpose2d); Caffe's weight layouts ([Cout,Cin,kh,kw], deconv [Cin,Cout,kh,kw],
base_conv_layer.cpp:125-139) are torch's, so .caffemodel blobs load without transposition.
"""
from __future__ import annotations

import math
import weakref
from typing import Dict, Optional


import torch
import torch.nn.functional as F

NEG_SLOPE = 0.1          # ReLU negative_slope of every FlowNet conv/deconv
FLOW_SCALE = 20.0        # deploy multiplies predict_flow2 by 20 (training divides GT by 20)
SD_FLOW_SCALE = 0.05     # FlowNet-SD predicts in units of 0.05 px: its deploy branch multiplies by 0.05 (released FlowNet2 graph; flownetsd_flow2 / div_flow in the third-party ports)
LOSS_WEIGHTS = {6: 0.32, 5: 0.08, 4: 0.02, 3: 0.01, 2: 0.005}   # Appendix B (memory)

# name: (kind, Cin, Cout, k, stride, pad)
_ENC_TAIL = [("conv4", 256, 512, 3, 2, 1), ("conv4_1", 512, 512, 3, 1, 1), ("conv5", 512, 512, 3, 2, 1),
             ("conv5_1", 512, 512, 3, 1, 1), ("conv6", 512, 1024, 3, 2, 1), ("conv6_1", 1024, 1024, 3, 1, 1)]
_DEC = [("Convolution1", "conv", 1024, 2), ("deconv5", "deconv", 1024, 512), ("upsample_flow6to5", "deconv", 2, 2),
        ("Convolution2", "conv", 1026, 2), ("deconv4", "deconv", 1026, 256), ("upsample_flow5to4", "deconv", 2, 2),
        ("Convolution3", "conv", 770, 2), ("deconv3", "deconv", 770, 128), ("upsample_flow4to3", "deconv", 2, 2),
        ("Convolution4", "conv", 386, 2), ("deconv2", "deconv", 386, 64), ("upsample_flow3to2", "deconv", 2, 2),
        ("Convolution5", "conv", 194, 2)]


def layer_table(kind: str = "C", in_channels: int = 6):
    """[(name, 'conv'|'deconv', Cin, Cout, k, stride, pad)] in execution order."""
    t = []
    if kind == "C":
        t += [("conv1", "conv", 3, 64, 7, 2, 3), ("conv2", "conv", 64, 128, 5, 2, 2), ("conv3", "conv", 128, 256, 5, 2, 2),
              ("conv_redir", "conv", 256, 32, 1, 1, 0), ("conv3_1", "conv", 473, 256, 3, 1, 1)]
    else:
        t += [("conv1", "conv", in_channels, 64, 7, 2, 3), ("conv2", "conv", 64, 128, 5, 2, 2), ("conv3", "conv", 128, 256, 5, 2, 2),
              ("conv3_1", "conv", 256, 256, 3, 1, 1)]
    t += [(n, "conv", ci, co, k, s, p) for (n, ci, co, k, s, p) in _ENC_TAIL]
    for (n, k, ci, co) in _DEC:
        t.append((n, k, ci, co, 3, 1, 1) if k == "conv" else (n, k, ci, co, 4, 2, 1))
    return t


def init_params(kind: str = "C", seed: int = 0, device="cpu", in_channels: int = 6) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (no .caffemodel can be downloaded here): He-normal for the leaky
    convs, small predict_flow / upsample heads so flows come out O(1-10 px); biases 0."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    for (name, k, ci, co, ks, s, p) in layer_table(kind, in_channels):
        shape = (co, ci, ks, ks) if k == "conv" else (ci, co, ks, ks)
        fan_in = ci * ks * ks if k == "conv" else ci * ks * ks / (s * s)
        std = math.sqrt(2.0 / ((1 + NEG_SLOPE ** 2) * fan_in))
        if name.startswith("Convolution"):
            std *= 0.5
        if name.startswith("upsample_flow"):
            std = 0.25
        params[name + ".w"] = (torch.randn(shape, generator=g) * std).to(device)
        params[name + ".b"] = torch.zeros(co).to(device)
    return params


def num_params(params) -> int:
    return sum(int(v.numel()) for v in params.values())


def _conv(x, P, name, stride, pad, act=True, backend=None, relu_chain=None):
    """Convolution (+ ReLU{negative_slope 0.1}).  With a backend that has conv_bias_leaky_relu the library runs the
    bias-free convolution and bias + activation are one in-place pass (csrc/bias_act.hip) instead of two."""
    w = P[name + ".w"]
    if _TRACE_CONV:
        y = _conv_routed(x, P, name, stride, pad, act, backend)
        print("conv route %-16s in %-22s k%d s%d act=%d -> %s" % (name, tuple(x.shape), w.shape[2], stride, act, _LAST_ROUTE[0]))
        return y
    if relu_chain is not None:
        return conv_forward(x, w, P[name + ".b"], stride, pad, act, backend, relu_chain=relu_chain)
    return _conv_routed(x, P, name, stride, pad, act, backend)


_TRACE_CONV = __import__("os").environ.get("FN2_TRACE_CONV") == "1"
_LAST_ROUTE = [""]


def _conv_routed(x, P, name, stride, pad, act, backend):
    return conv_forward(x, P[name + ".w"], P[name + ".b"], stride, pad, act, backend)


def conv_forward(x, w, b, stride, pad, act, backend, slope=None, relu_chain=None):
    """Convolution{w, stride, pad} + bias (+ ReLU{NEG_SLOPE} when act) on explicit tensors: the routing every graph of this file and the
    Convolution layer of the prototxt executor (flownet2_amd.layers.ConvolutionLayer) share -- same kernels, same bits."""
    P = {"x.w": w, "x.b": b}
    name = "x"
    slope = NEG_SLOPE if slope is None else slope
    _LAST_ROUTE[0] = "library conv2d"
    if w.shape[2] == 7 and w.shape[1] % 4 == 0 and backend is not None and hasattr(backend, "conv_mfma_relu"):
        # the 12-channel stems of FlowNet2's stacked nets are whole channel quads: the direct MFMA kernel does them in one pass
        # (the register-resident stem kernel needs two passes with partial sums through memory)
        y = backend.conv_mfma_relu(x, w, P[name + ".b"], stride, pad, slope, act)
        if y is not None:
            _LAST_ROUTE[0] = "fn2 MFMA conv"
            return y
    if act and stride == 2 and pad == 3 and w.shape[2] == 7 and backend is not None and hasattr(backend, "conv_k7s2_relu"):
        y = (backend.conv_k7s2_relu(x, w, P[name + ".b"], slope) if relu_chain is None      # conv1 + ReLU1 in one kernel (csrc/conv_stem.hip)
             else backend.conv_k7s2_relu(x, w, P[name + ".b"], slope, relu_chain=relu_chain))
        if y is None and relu_chain is not None:
            raise RuntimeError("relu_chain: the stem kernel does not take this layer (its consumer was told it would)")
        if y is not None:
            _LAST_ROUTE[0] = "stem kernel"
            return y
    if relu_chain is not None and _LAST_ROUTE[0] != "stem kernel":
        raise RuntimeError("relu_chain: only the stem layer is a chain producer on this route")
    if w.shape[2] in (1, 3, 5) and backend is not None and hasattr(backend, "conv_mfma_relu"):
        y = backend.conv_mfma_relu(x, w, P[name + ".b"], stride, pad, slope, act)    # Winograd / direct MFMA convolution (1x1: a plain MFMA GEMM), bias + ReLU fused
        if y is not None:
            _LAST_ROUTE[0] = "fn2 MFMA conv"
            return y
    conv2d = getattr(backend, "lib_conv2d", None) or (lambda xx, ww, bb, s, p: F.conv2d(xx, ww, bb, stride=s, padding=p))
    if act and backend is not None and hasattr(backend, "conv_bias_leaky_relu"):
        return backend.conv_bias_leaky_relu(conv2d(x, w, None, stride, pad), P[name + ".b"], slope)
    y = conv2d(x, w, P[name + ".b"], stride, pad)
    return F.leaky_relu(y, slope) if act else y


_CONSTS: Dict[tuple, torch.Tensor] = {}


def _const(device, values):
    """Small fp32 device constants (mean, output scale), uploaded once: no host-to-device copy inside a step, so a
    step can be captured into a hipGraph."""
    key = (str(device), tuple(float(v) for v in values))
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(key[1], device=device, dtype=torch.float32)
    return _CONSTS[key]


_WT_CACHE: Dict[int, tuple] = {}      # id(weight tensor) -> (weak reference to it, its _version, transposed copy)


def _transposed_deconv_weight(w):
    """weight [Cin, Cout, 4, 4] -> [Cout*16, Cin] contiguous (the A operand of the deconvolution GEMM).  Cached per
    parameter tensor OBJECT (the entry dies with the tensor, so a new tensor that reuses the id or the storage address
    never sees it) and rebuilt when the tensor is modified in place (torch bumps `_version`) or -- for a tensor that requires
    grad -- when an optimizer has stepped since (functional._cached: fused optimizers write without touching `_version`)."""
    from . import functional as Fn
    return Fn._cached(_WT_CACHE, id(w), w, lambda: w.detach().reshape(w.shape[0], w.shape[1] * 16).t().contiguous())


def _deconv(x, P, name, act=True, backend=None):
    return deconv_forward(x, P[name + ".w"], P[name + ".b"], act, backend)


def deconv_forward(x, w, b, act=True, backend=None, slope=None):
    """Deconvolution{4, 2, 1} + bias (+ ReLU): the routing shared with flownet2_amd.layers.DeconvolutionLayer."""
    P = {"x.w": w, "x.b": b}
    name = "x"
    slope = NEG_SLOPE if slope is None else slope
    if act and backend is not None and hasattr(backend, "deconv_mfma_relu") and w.shape[0] >= 64:
        y = backend.deconv_mfma_relu(x, w, P[name + ".b"], slope, True)      # one MFMA kernel, a parity class per wave (csrc/conv_plane.hip)
        if y is not None:
            return y
    if act and backend is not None and hasattr(backend, "deconv_gemm_relu") and w.shape[0] >= 64:
        training = torch.is_grad_enabled() and (w.requires_grad or x.requires_grad)
        y = backend.deconv_gemm_relu(x, None if training else _transposed_deconv_weight(w), P[name + ".b"], w.shape[1], 4, 2, 1, slope,
                                     weight=w)
        if y is not None:
            return y
    deconv2d = getattr(backend, "lib_conv_transpose2d", None) or (lambda xx, ww, bb, s, p: F.conv_transpose2d(xx, ww, bb, stride=s, padding=p))
    if act and backend is not None and hasattr(backend, "conv_bias_leaky_relu"):
        return backend.conv_bias_leaky_relu(deconv2d(x, P[name + ".w"], None, 2, 1), P[name + ".b"], slope)
    y = deconv2d(x, P[name + ".w"], P[name + ".b"], 2, 1)
    return F.leaky_relu(y, slope) if act else y


def _conv_into_concat(x, P, name, stride, pad, extra_channels, backend, relu_chain=None):
    """A convolution whose output is the FIRST input of a Concat (a decoder skip tensor): where the own MFMA kernels apply and no
    gradient is needed, the Concat blob [N, Cout + extra_channels, H, W] is allocated here and the convolution writes its channels
    straight into it (ConcatLayer, concat_layer.cu:8-52, becomes a no-op for this input).  Returns (blob or None, the layer's output --
    a channel-slice view of the blob, or an ordinary tensor)."""
    w = P[name + ".w"]
    # The blob route writes the later Concat inputs in place (deconvolution, upsampled flow).  Without autograd what those calls return is
    # discarded.  With autograd (round 5, CONCAT_IN_PLACE_TRAINING) every producer is an autograd function whose output is its slice view
    # of the blob -- a plain buffer outside the graph -- and _ConcatInPlace ties the slices together for the consumer: no Concat copy in a
    # training step either.
    training = torch.is_grad_enabled() and (x.requires_grad or _any_requires_grad(P))
    if (backend is not None and hasattr(backend, "conv_mfma_relu") and x.is_cuda and w.shape[2] in (3, 5)
            and (not training or (CONCAT_IN_PLACE_TRAINING[0] and hasattr(backend, "conv_backward")))):
        k = w.shape[2]
        ho, wo = (x.shape[2] + 2 * pad - k) // stride + 1, (x.shape[3] + 2 * pad - k) // stride + 1
        blob = torch.empty((x.shape[0], w.shape[0] + extra_channels, ho, wo), device=x.device, dtype=x.dtype)
        y = (backend.conv_mfma_relu(x, w, P[name + ".b"], stride, pad, NEG_SLOPE, True, out=blob, out_c0=0) if relu_chain is None
             else backend.conv_mfma_relu(x, w, P[name + ".b"], stride, pad, NEG_SLOPE, True, out=blob, out_c0=0, relu_chain=relu_chain))
        if y is not None:
            return blob, (y if (training and y.requires_grad) else blob[:, :w.shape[0]])
    if relu_chain is not None:
        raise RuntimeError("relu_chain: conv '%s' left the own-kernel route after its producer was told it would fold the ReLU derivative" % name)
    return None, _conv(x, P, name, stride, pad, backend=backend)


def _conv_into(x, P, name, stride, pad, blob, c0, backend):
    """Convolution + bias + ReLU written into channels [c0, c0 + Cout) of `blob` by the own kernels, as an autograd function when a gradient
    is needed (the returned slice view carries the graph); None when no own kernel takes the layer."""
    return backend.conv_mfma_relu(x, P[name + ".w"], P[name + ".b"], stride, pad, NEG_SLOPE, True, out=blob, out_c0=c0)


def _any_requires_grad(P) -> bool:
    vals = P.P.values() if isinstance(P, _Prefixed) else P.values()
    return any(v.requires_grad for v in vals)


def _stage_deconv(P, x, dname, blob, cs, cd, backend, allow_copy=True):
    """ReLU(deconv(x)) written into channels [cs, cs + cd) of a refinement stage's Concat blob.  Returns what the kernel wrapper
    returned (with autograd: the slice view that carries the graph), or None after the copy fallback."""
    d = None
    w = P[dname + ".w"]
    training = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
    if hasattr(backend, "deconv_mfma_relu"):
        d = backend.deconv_mfma_relu(x, w, P[dname + ".b"], NEG_SLOPE, True, out=blob, out_c0=cs)
    if d is None and hasattr(backend, "deconv_gemm_relu") and w.shape[0] >= 64:
        # GEMM (weight^T x bottom), then our col2im + bias + ReLU pass straight into the blob
        d = backend.deconv_gemm_relu(x, None if training else _transposed_deconv_weight(w), P[dname + ".b"], cd, 4, 2, 1, NEG_SLOPE,
                                     weight=w if training else None, out=blob, out_c0=cs)
    if d is None and allow_copy:
        blob[:, cs:cs + cd].copy_(_deconv(x, P, dname, backend=backend))
    return d


def _refine_stage(P, skip, x, dname, flow, uname, backend):
    """One refinement Concat [skip | ReLU(deconv(x)) | upsampled flow].  `skip` is a tensor or a pair (concat blob, tensor) from
    _conv_into_concat: with a blob the Concat layer (concat_layer.cu:8-52) has nothing to copy -- the skip convolution already wrote its
    channels there, and the deconvolution and the upsampled flow are written behind them by their own kernels."""
    def up(t, name, out=None, out_c0=0):
        if backend is not None and hasattr(backend, "upsample_flow_deconv"):
            if out is not None:
                return backend.upsample_flow_deconv(t, P[name + ".w"], P[name + ".b"], out=out, out_c0=out_c0)
            return backend.upsample_flow_deconv(t, P[name + ".w"], P[name + ".b"])
        return _deconv(t, P, name, act=False)

    blob, s = skip if isinstance(skip, tuple) else (None, skip)
    if blob is None:
        return torch.cat([s, _deconv(x, P, dname, backend=backend), up(flow, uname)], 1)
    cs, cd = s.shape[1], P[dname + ".w"].shape[1]
    assert blob.shape[1] == cs + cd + 2
    stage_params = [P[n] for n in (dname + ".w", dname + ".b", uname + ".w", uname + ".b")]
    if torch.is_grad_enabled() and (s.requires_grad or x.requires_grad or flow.requires_grad or any(p.requires_grad for p in stage_params)):
        # training (also a partial freeze whose first trainable tensors are this stage's own deconvolution / up-sampling weights): the
        # three producers are autograd functions that wrote / write their slices; _ConcatInPlace hands the consumer the blob
        d = _stage_deconv(P, x, dname, blob, cs, cd, backend, allow_copy=False)
        u = up(flow, uname, out=blob, out_c0=cs + cd) if (d is not None and hasattr(backend, "upsample_flow_deconv")) else None
        if d is None or u is None:          # a producer without an own kernel for this shape: the stock Concat (a copy) for this stage
            return torch.cat([s, _deconv(x, P, dname, backend=backend), up(flow, uname)], 1)
        part = lambda t, c0, c: t if t.requires_grad else blob[:, c0:c0 + c]      # (a frozen producer returned the blob it wrote into)
        return _ConcatInPlace.apply(blob, s, part(d, cs, cd), part(u, cs + cd, 2))
    _stage_deconv(P, x, dname, blob, cs, cd, backend)
    if hasattr(backend, "upsample_flow_deconv"):
        up(flow, uname, out=blob, out_c0=cs + cd)
    else:
        blob[:, cs + cd:].copy_(up(flow, uname))
    return blob


def _decoder(P, conv6_1, conv5_1, conv4_1, conv3_1, conv2, backend=None):
    """The refinement stages.  Every skip argument is a tensor or a pair (concat blob, tensor) from _conv_into_concat: with a blob the
    Concat layer (concat_layer.cu:8-52) has nothing to copy -- the skip convolution already wrote its channels there, and the
    deconvolution and the upsampled flow are written behind them by their own kernels."""
    # predict_flow (3x3 conv -> 2 ch) and upsample_flow (4x4/2 deconv 2 -> 2) go through the backend's flow-head
    # kernels when it has them (HIP: flow_head.hip); otherwise the stock conv path
    def pf(x, name):
        if backend is not None and hasattr(backend, "predict_flow_conv"):
            return backend.predict_flow_conv(x, P[name + ".w"], P[name + ".b"])
        return _conv(x, P, name, 1, 1, act=False, backend=backend)

    def stage(skip, x, dname, flow, uname):
        return _refine_stage(P, skip, x, dname, flow, uname, backend)

    # (Round 4 measured the flow head of every level on a second HIP stream beside that level's deconvolution: 2.366 ms per FlowNetC step against
    # 2.343 without -- the fork / join event pairs cost more than ~90 us of head kernels can hide behind GEMMs that fill every CU.  The branch was
    # removed in round 5; what a second stream pays for is LONG independent chains: FlowNet-SD beside the CSS stack, weight gradients beside data
    # gradients.)

    flow6 = pf(conv6_1, "Convolution1")
    c5 = stage(conv5_1, conv6_1, "deconv5", flow6, "upsample_flow6to5")
    flow5 = pf(c5, "Convolution2")
    c4 = stage(conv4_1, c5, "deconv4", flow5, "upsample_flow5to4")
    flow4 = pf(c4, "Convolution3")
    c3 = stage(conv3_1, c4, "deconv3", flow4, "upsample_flow4to3")
    flow3 = pf(c3, "Convolution4")
    c2 = stage(conv2, c3, "deconv2", flow3, "upsample_flow3to2")
    flow2 = pf(c2, "Convolution5")
    return {2: flow2, 3: flow3, 4: flow4, 5: flow5, 6: flow6}


def _skip_conv(x, P, name, stride, pad, dname, backend, relu_chain=None):
    """An encoder convolution whose output is also the first input of a refinement Concat: (concat blob or None, output tensor)."""
    return _conv_into_concat(x, P, name, stride, pad, P[dname + ".w"].shape[1] + 2, backend, relu_chain=relu_chain)


RELU_CHAIN = [True]      # A/B hook: False = every layer undoes its own ReLU in a pass of its own (rounds 1-5)


def _stem_chain(P, x, backend):
    """conv1 -> conv2 in a TRAINING graph: conv2 is conv1's only consumer, so ReLUBackward of conv1 is folded into the epilogue of conv2's data
    gradient (the transposed 5x5 / 2 convolution) and conv1's bias gradient comes out of its weight-gradient kernel: no pass over the largest
    activation of the net (147 MB at batch 8 @448x320: read twice, written once) between the two.  Returns the (producer, consumer) handles
    for the two layers, or (None, None)."""
    if not (RELU_CHAIN[0] and CONCAT_IN_PLACE_TRAINING[0] and torch.is_grad_enabled() and x.is_cuda and hasattr(backend, "relu_chain_supported")):
        return None, None
    w1, w2 = P["conv1.w"], P["conv2.w"]
    if not (w1.requires_grad and w2.requires_grad and P["conv1.b"].requires_grad) or x.requires_grad:
        return None, None
    if w1.shape[2] != 7 or w1.shape[1] % 4 == 0 or x.shape[3] % 8 != 0:       # (conv1 must take the stem kernel: whole blobs)
        return None, None
    c1_shape = (x.shape[0], w1.shape[0], (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1)
    if not backend.relu_chain_supported(c1_shape, w2, 2, 2):
        return None, None
    cell = {"masked": False, "slope": NEG_SLOPE}
    return (1, cell), (2, cell)


CONCAT_IN_PLACE_TRAINING = [True]      # A/B hook: False = torch.cat for the refinement Concats of a training graph (rounds 1-4)


class _ConcatInPlace(torch.autograd.Function):
    """Concat (concat_layer.cu:8-52 / :62-90) of tensors that already ARE consecutive channel slices of `blob`: nothing to copy forward, the
    gradient of each is its slice of top_diff (a view: the producers' backward kernels read it in place)."""

    @staticmethod
    def forward(ctx, blob, *parts):
        ctx.cuts = [p.shape[1] for p in parts]
        assert sum(ctx.cuts) == blob.shape[1] and all(p.data_ptr() >= blob.data_ptr() for p in parts)
        return blob.view_as(blob)

    @staticmethod
    def backward(ctx, g):
        out, c0 = [None], 0
        for c in ctx.cuts:
            out.append(g[:, c0:c0 + c])
            c0 += c
        return tuple(out)


TOWER_SPLIT_OPS = [True]      # A/B hook: False = plain slices of the stacked tower batch in the training graph


class _SplitTowers(torch.autograd.Function):
    """The stacked siamese batch [2N, ...] -> (first tower, second tower).  Two plain slices cost the backward pass two zero-fills of the
    stacked shape, two strided copies and an add (SliceBackward twice, then the sum); here it is ONE concatenation of the two gradients."""

    @staticmethod
    def forward(ctx, x):
        n = x.shape[0] // 2
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        return torch.cat([ga, gb], 0)


class _StackedAndFirstTower(torch.autograd.Function):
    """The stacked batch as it is (for the next tower layer) AND its first tower (the skip connection into the refinement).  The skip's
    gradient is added into the first half of the stacked gradient in place -- instead of a zero-fill of the stacked shape, a strided copy
    and a full-size add."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x[:x.shape[0] // 2]

    @staticmethod
    def backward(ctx, g_all, g_first):
        # g_all is the data gradient of the ONE layer that read output 0 (conv3's transposed convolution), produced for this node alone: both
        # outputs are locals of flownet_c_core -- nobody outside can hang a tensor hook or retain_grad on them, and a retained graph
        # recomputes g_all on its next pass -- so the skip gradient is added in place (a clone is 146 MB of traffic per step).  A gradient
        # that arrives as a view of something else is NOT ours to write: copy then.
        if g_all._base is not None or not g_all.is_contiguous():
            g_all = g_all.clone(memory_format=torch.contiguous_format)
        g_all[:g_first.shape[0]].add_(g_first)
        return g_all


def flownet_c_core(P, img0, img1, backend, towers=None):
    """Pre-processed images [N,3,H,W] (H, W multiples of 64) -> {scale: flow prediction /20}.  `towers`: the two images already
    stacked along the batch axis [2N,3,H,W] (the deploy head writes them there directly)."""
    # siamese towers share weights (param { name: } sharing, net.cpp:451-540): one batch of 2N through conv1-3
    x = towers if towers is not None else torch.cat([img0, img1], 0)
    n = x.shape[0] // 2
    chain1, chain2 = _stem_chain(P, x, backend)
    c1 = _conv(x, P, "conv1", 2, 3, backend=backend, relu_chain=chain1)
    # conv2 of BOTH towers goes into a [2N, 128 + 64 + 2, h, w] blob: its first N samples are the concat2 blob of the refinement
    blob2, c2 = _skip_conv(c1, P, "conv2", 2, 2, "deconv2", backend, relu_chain=chain2)
    training = TOWER_SPLIT_OPS[0] and torch.is_grad_enabled() and c2.requires_grad
    c2_first = None
    if training:
        c2, c2_first = _StackedAndFirstTower.apply(c2)
    c3 = _conv(c2, P, "conv3", 2, 2, backend=backend)
    c3a, c3b = _SplitTowers.apply(c3) if (training and c3.requires_grad) else (c3[:n], c3[n:])
    cat = redir = None
    cr = P["conv_redir.w"].shape[0]
    if hasattr(backend, "correlation_relu_into") and c3a.is_cuda:
        # the correlation writes its 441 activated planes straight into the [conv_redir | corr] blob (no ReLU pass, no Concat pass), and
        # conv_redir (1x1: the own MFMA GEMM kernel) its 32 channels in front of them
        cat = torch.empty((n, cr + 441, c3a.shape[2], c3a.shape[3]), device=c3a.device, dtype=c3a.dtype)
        in_graph = (CONCAT_IN_PLACE_TRAINING[0] and torch.is_grad_enabled() and (c3a.requires_grad or c3b.requires_grad)
                    and hasattr(backend, "conv_backward"))
        if in_graph:
            # training: both producers are autograd functions that write their slice of the blob (the fused correlation + ReLU; conv_redir)
            redir = _conv_into(c3a, P, "conv_redir", 1, 0, cat, 0, backend)
            corr = backend.correlation_relu_into(c3a, c3b, cat, cr, NEG_SLOPE, pad=20, kernel_size=1, max_displacement=20, stride_1=1, stride_2=2,
                                                 training=True)
            if redir is not None and redir.requires_grad:
                cat = _ConcatInPlace.apply(cat, redir, corr)
            else:                                       # conv_redir without an own kernel for this shape (or frozen): the stock Concat
                cat = torch.cat([_conv(c3a, P, "conv_redir", 1, 0, backend=backend), corr], 1)
        elif backend.correlation_relu_into(c3a, c3b, cat, cr, NEG_SLOPE, pad=20, kernel_size=1, max_displacement=20,
                                           stride_1=1, stride_2=2) is None:
            cat = None
        else:
            into = None
            if hasattr(backend, "conv_mfma_relu") and not (torch.is_grad_enabled() and _any_requires_grad(P)):
                into = backend.conv_mfma_relu(c3a, P["conv_redir.w"], P["conv_redir.b"], 1, 0, NEG_SLOPE, True, out=cat, out_c0=0)
            if into is None:
                cat[:, :cr].copy_(_conv(c3a, P, "conv_redir", 1, 0, backend=backend))
    if cat is None:
        redir = _conv(c3a, P, "conv_redir", 1, 0, backend=backend)
        corr = backend.correlation(c3a, c3b, pad=20, kernel_size=1, max_displacement=20, stride_1=1, stride_2=2)
        cat = torch.cat([redir, F.leaky_relu(corr, NEG_SLOPE)], 1)
    blob3, c31 = _skip_conv(cat, P, "conv3_1", 1, 1, "deconv3", backend)
    c4 = _conv(c31, P, "conv4", 2, 1, backend=backend)
    blob4, c41 = _skip_conv(c4, P, "conv4_1", 1, 1, "deconv4", backend)
    c5 = _conv(c41, P, "conv5", 2, 1, backend=backend)
    blob5, c51 = _skip_conv(c5, P, "conv5_1", 1, 1, "deconv5", backend)
    c6 = _conv(c51, P, "conv6", 2, 1, backend=backend)
    c61 = _conv(c6, P, "conv6_1", 1, 1, backend=backend)
    c2s = c2_first if c2_first is not None else c2[:n]
    skip2 = (blob2[:n], c2s) if blob2 is not None else c2s
    return _decoder(P, c61, (blob5, c51), (blob4, c41), (blob3, c31), skip2, backend)


def flownet_s_core(P, x, backend=None):
    c1 = _conv(x, P, "conv1", 2, 3, backend=backend)
    blob2, c2 = _skip_conv(c1, P, "conv2", 2, 2, "deconv2", backend)
    c3 = _conv(c2, P, "conv3", 2, 2, backend=backend)
    blob3, c31 = _skip_conv(c3, P, "conv3_1", 1, 1, "deconv3", backend)
    c4 = _conv(c31, P, "conv4", 2, 1, backend=backend)
    blob4, c41 = _skip_conv(c4, P, "conv4_1", 1, 1, "deconv4", backend)
    c5 = _conv(c41, P, "conv5", 2, 1, backend=backend)
    blob5, c51 = _skip_conv(c5, P, "conv5_1", 1, 1, "deconv5", backend)
    c6 = _conv(c51, P, "conv6", 2, 1, backend=backend)
    c61 = _conv(c6, P, "conv6_1", 1, 1, backend=backend)
    return _decoder(P, c61, (blob5, c51), (blob4, c41), (blob3, c31), (blob2, c2), backend)


def adapted_size(h: int, w: int, divisor: int = 64):
    """scripts/run-flownet.py:43-45."""
    return int(math.ceil(h / divisor) * divisor), int(math.ceil(w / divisor) * divisor)


def deploy_forward(kind: str, P, img0, img1, backend, mean: Optional[torch.Tensor] = None):
    """Deploy net: raw 0..255 BGR images [N,3,H,W] -> predict_flow_final [N,2,H,W] in pixels.

    Head/tail as in SURVEY.md Appendix B: Eltwise(1/255) -> Resample(ADAPTED, LINEAR) -> mean
    subtraction -> net -> x20 -> Resample(TARGET, LINEAR) -> diag(SCALE_WIDTH, SCALE_HEIGHT) 1x1 conv.
    """
    N, _, H, W = img0.shape
    ah, aw = adapted_size(H, W)
    neg_default = mean is None
    if mean is None:
        mean = _const(img0.device, (0.411, 0.433, 0.45))   # BGR order of a typical RGB mean
    if (ah, aw) == (H, W) and img0.is_cuda and not torch.is_grad_enabled() and hasattr(backend, "scale_shift"):
        # At the ADAPTED size the LINEAR Resample is the identity (one tap of weight 1), so the head is x * (1 / 255) - mean per image:
        # one pass per image (product and difference rounded separately: the bits of the Eltwise and mean-subtraction layers), written
        # straight into the blob the towers (C: batch axis) or conv1 (S: channel axis) read -- instead of scale, subtract and concat.
        neg_mean = _const(img0.device, (-0.411, -0.433, -0.45)) if neg_default else (-mean).contiguous()
        if kind == "C":
            x = torch.empty((2 * N, 3, H, W), device=img0.device, dtype=torch.float32)
            backend.scale_shift(img0, 1.0 / 255.0, neg_mean, out=x[:N])
            backend.scale_shift(img1, 1.0 / 255.0, neg_mean, out=x[N:])
            flows = flownet_c_core(P, None, None, backend, towers=x)
        else:
            x = torch.empty((N, 6, H, W), device=img0.device, dtype=torch.float32)
            backend.scale_shift(img0, 1.0 / 255.0, neg_mean, out=x, out_c0=0)
            backend.scale_shift(img1, 1.0 / 255.0, neg_mean, out=x, out_c0=3)
            flows = flownet_s_core(P, x, backend)
    else:
        pre = []
        for im in (img0, img1):
            x = im * (1.0 / 255.0)                                                 # Eltwise, coeff 1/255
            if (ah, aw) != (H, W):
                x = backend.resample(x, ah, aw)                                    # Resample to ADAPTED size
            pre.append(x - mean.view(1, 3, 1, 1))                                  # DataAugmentation mean subtraction (deploy slice)
        if kind == "C":
            flows = flownet_c_core(P, pre[0], pre[1], backend)
        else:
            flows = flownet_s_core(P, torch.cat(pre, 1), backend)
    if img0.is_cuda and not torch.is_grad_enabled() and hasattr(backend, "resample_slices"):
        flow = backend.resample_slices(flows[2], H, W, in_scale=FLOW_SCALE)         # Eltwise{20} folded into the Resample (same roundings)
        if (ah, aw) == (H, W):
            return flow                                                              # diag(1, 1): x * 1.0f is x
    else:
        flow = flows[2] * FLOW_SCALE                                                # Eltwise, coeff 20
        flow = backend.resample(flow, H, W)                                         # Resample to TARGET size (x4 up-sampling)
    scale = _const(flow.device, (W / float(aw), H / float(ah)))   # run-flownet.py:47-48
    return flow * scale.view(1, 2, 1, 1)                                            # 1x1 conv, diagonal filler


def loss_targets_ahead(gt_flow, backend, divisors=None):
    """The Downsample(GT * 0.05) pyramid of multiscale_loss issued ahead of the forward pass on a second stream (backend.downsample_ahead:
    it depends on the ground truth alone).  Returns an opaque handle for multiscale_loss(..., targets=handle), or None when the backend
    has no such form."""
    if not (hasattr(backend, "downsample_ahead") and gt_flow.is_cuda):
        return None
    H, W = gt_flow.shape[2], gt_flow.shape[3]
    divisors = divisors or {s: 2 ** s for s in LOSS_WEIGHTS}
    gt = gt_flow * (1.0 / FLOW_SCALE)
    sizes = [(H // divisors[s], W // divisors[s]) for s in LOSS_WEIGHTS]
    tgts, ev = backend.downsample_ahead(gt, sizes)
    return {"targets": dict(zip(LOSS_WEIGHTS, tgts)), "event": ev, "sizes": dict(zip(LOSS_WEIGHTS, sizes))}


def multiscale_loss(flows, gt_flow, backend, targets=None):
    """Training loss: per scale Downsample(GT * 0.05) -> L1Loss{l2_per_location, normalize_by_num_entries}.  `targets`: the handle of
    loss_targets_ahead (the same Downsample launches, issued at the start of the step on a second stream)."""
    if hasattr(backend, "l1_loss_multi") and gt_flow.is_cuda:
        # the five loss layers in one launch per direction, the weighted sum (Net::ForwardFromTo's loss += ...) included
        scales = list(LOSS_WEIGHTS.items())
        preds = [flows[s] for s, _ in scales]
        if targets is not None and all(tuple(targets["sizes"][s]) == tuple(flows[s].shape[2:]) for s, _ in scales):
            backend.wait_ahead(targets["event"])
            tgts = [targets["targets"][s] for s, _ in scales]
        else:
            gt = gt_flow * (1.0 / FLOW_SCALE)
            tgts = [backend.downsample(gt, p.shape[2], p.shape[3]) for p in preds]
        return backend.l1_loss_multi(preds, tgts, [w for _, w in scales], l2_per_location=True, normalize_by_num_entries=True)[0]
    gt = gt_flow * (1.0 / FLOW_SCALE)
    total = 0.0
    for s, w in LOSS_WEIGHTS.items():
        pred = flows[s]
        tgt = backend.downsample(gt, pred.shape[2], pred.shape[3])
        total = total + w * backend.l1_loss(pred, tgt, l2_per_location=True, normalize_by_num_entries=True)
    return total


# FLOP model (multiply-add = 2 flops) used by bench.py
def conv_flops(kind: str, h: int, w: int, in_channels: int = 6) -> float:
    fl = 0.0
    size = {}
    hh, ww = h, w
    res = {"conv1": 2, "conv2": 4, "conv3": 8, "conv_redir": 8, "conv3_1": 8, "conv4": 16, "conv4_1": 16, "conv5": 32, "conv5_1": 32,
           "conv6": 64, "conv6_1": 64, "Convolution1": 64, "deconv5": 32, "upsample_flow6to5": 32, "Convolution2": 32,
           "deconv4": 16, "upsample_flow5to4": 16, "Convolution3": 16, "deconv3": 8, "upsample_flow4to3": 8, "Convolution4": 8,
           "deconv2": 4, "upsample_flow3to2": 4, "Convolution5": 4}
    for (name, k, ci, co, ks, s, p) in layer_table(kind, in_channels):
        oh, ow = hh // res[name], ww // res[name]
        macs = oh * ow * co * ci * ks * ks if k == "conv" else (oh // 2) * (ow // 2) * ci * co * ks * ks
        mult = 2 if (kind == "C" and name in ("conv1", "conv2", "conv3")) else 1
        fl += 2.0 * macs * mult
    return fl


# ------------------------------------------------------------------------------------------------------------------
# Full FlowNet2 (BASELINE.json configs[2]): FlowNetC -> FlowNetS -> FlowNetS ("CSS") || FlowNet-SD -> fusion net.
# Topology from SURVEY.md Appendix B (memory of the paper / third-party converters -- the prototxts are not in the
# reference tree).  Weight names carry the prefixes of the released caffemodels: net2_, net3_, netsd_, fuse_.
# ------------------------------------------------------------------------------------------------------------------
_SD_TABLE = [("conv0", "conv", 6, 64, 3, 1, 1), ("conv1", "conv", 64, 64, 3, 2, 1), ("conv1_1", "conv", 64, 128, 3, 1, 1),
             ("conv2", "conv", 128, 128, 3, 2, 1), ("conv2_1", "conv", 128, 128, 3, 1, 1), ("conv3", "conv", 128, 256, 3, 2, 1),
             ("conv3_1", "conv", 256, 256, 3, 1, 1), ("conv4", "conv", 256, 512, 3, 2, 1), ("conv4_1", "conv", 512, 512, 3, 1, 1),
             ("conv5", "conv", 512, 512, 3, 2, 1), ("conv5_1", "conv", 512, 512, 3, 1, 1), ("conv6", "conv", 512, 1024, 3, 2, 1),
             ("conv6_1", "conv", 1024, 1024, 3, 1, 1),
             ("Convolution1", "conv", 1024, 2, 3, 1, 1), ("deconv5", "deconv", 1024, 512, 4, 2, 1), ("upsample_flow6to5", "deconv", 2, 2, 4, 2, 1),
             ("interconv5", "conv", 1026, 512, 3, 1, 1), ("Convolution2", "conv", 512, 2, 3, 1, 1),
             ("deconv4", "deconv", 1026, 256, 4, 2, 1), ("upsample_flow5to4", "deconv", 2, 2, 4, 2, 1),
             ("interconv4", "conv", 770, 256, 3, 1, 1), ("Convolution3", "conv", 256, 2, 3, 1, 1),
             ("deconv3", "deconv", 770, 128, 4, 2, 1), ("upsample_flow4to3", "deconv", 2, 2, 4, 2, 1),
             ("interconv3", "conv", 386, 128, 3, 1, 1), ("Convolution4", "conv", 128, 2, 3, 1, 1),
             ("deconv2", "deconv", 386, 64, 4, 2, 1), ("upsample_flow3to2", "deconv", 2, 2, 4, 2, 1),
             ("interconv2", "conv", 194, 64, 3, 1, 1), ("Convolution5", "conv", 64, 2, 3, 1, 1)]
_FUSE_TABLE = [("conv0", "conv", 11, 64, 3, 1, 1), ("conv1", "conv", 64, 64, 3, 2, 1), ("conv1_1", "conv", 64, 128, 3, 1, 1),
               ("conv2", "conv", 128, 128, 3, 2, 1), ("conv2_1", "conv", 128, 128, 3, 1, 1),
               ("Convolution5", "conv", 128, 2, 3, 1, 1), ("deconv1", "deconv", 128, 32, 4, 2, 1), ("upsample_flow2to1", "deconv", 2, 2, 4, 2, 1),
               ("interconv1", "conv", 162, 32, 3, 1, 1), ("Convolution6", "conv", 32, 2, 3, 1, 1),
               ("deconv0", "deconv", 162, 16, 4, 2, 1), ("upsample_flow1to0", "deconv", 2, 2, 4, 2, 1),
               ("interconv0", "conv", 82, 16, 3, 1, 1), ("Convolution7", "conv", 16, 2, 3, 1, 1)]


_SD_RES = {"conv0": 1, "conv1": 2, "conv1_1": 2, "conv2": 4, "conv2_1": 4, "conv3": 8, "conv3_1": 8, "conv4": 16, "conv4_1": 16, "conv5": 32,
           "conv5_1": 32, "conv6": 64, "conv6_1": 64, "Convolution1": 64, "deconv5": 32, "upsample_flow6to5": 32, "interconv5": 32,
           "Convolution2": 32, "deconv4": 16, "upsample_flow5to4": 16, "interconv4": 16, "Convolution3": 16, "deconv3": 8,
           "upsample_flow4to3": 8, "interconv3": 8, "Convolution4": 8, "deconv2": 4, "upsample_flow3to2": 4, "interconv2": 4, "Convolution5": 4}
_FUSE_RES = {"conv0": 1, "conv1": 2, "conv1_1": 2, "conv2": 4, "conv2_1": 4, "Convolution5": 4, "deconv1": 2, "upsample_flow2to1": 2,
             "interconv1": 2, "Convolution6": 2, "deconv0": 1, "upsample_flow1to0": 1, "interconv0": 1, "Convolution7": 1}


def _table_flops(table, res, h, w):
    fl = 0.0
    for (name, k, ci, co, ks, s, p) in table:
        oh, ow = h // res[name], w // res[name]          # output size of the layer (a deconv's input is half of it)
        fl += 2.0 * (oh * ow * co * ci * ks * ks if k == "conv" else (oh // 2) * (ow // 2) * ci * co * ks * ks)
    return fl


def flownet2_conv_flops(h: int, w: int) -> float:
    """Multiply-add flops of the conv / deconv layers of one full FlowNet2 forward (C + S + S + SD + fusion) per image pair."""
    return conv_flops("C", h, w) + 2 * conv_flops("S", h, w, 12) + _table_flops(_SD_TABLE, _SD_RES, h, w) + _table_flops(_FUSE_TABLE, _FUSE_RES, h, w)


def _init_table(table, prefix, g, params, device):
    for (name, k, ci, co, ks, s, p) in table:
        shape = (co, ci, ks, ks) if k == "conv" else (ci, co, ks, ks)
        fan_in = ci * ks * ks if k == "conv" else ci * ks * ks / (s * s)
        std = math.sqrt(2.0 / ((1 + NEG_SLOPE ** 2) * fan_in))
        if name.startswith("Convolution"):
            std *= 0.5
        if name.startswith("upsample_flow"):
            std = 0.25
        params[prefix + name + ".w"] = (torch.randn(shape, generator=g) * std).to(device)
        params[prefix + name + ".b"] = torch.zeros(co).to(device)


def init_params_flownet2(seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}
    _init_table(layer_table("C"), "", g, P, device)
    _init_table(layer_table("S", 12), "net2_", g, P, device)
    _init_table(layer_table("S", 12), "net3_", g, P, device)
    _init_table(_SD_TABLE, "netsd_", g, P, device)
    _init_table(_FUSE_TABLE, "fuse_", g, P, device)
    return P


class _Prefixed(dict):
    """View of a parameter dict under a layer-name prefix."""

    def __init__(self, P, prefix):
        super().__init__()
        self.P, self.prefix = P, prefix

    def __getitem__(self, k):
        return self.P[self.prefix + k]


def _pf(P, x, name, backend):
    if backend is not None and hasattr(backend, "predict_flow_conv"):
        return backend.predict_flow_conv(x, P[name + ".w"], P[name + ".b"])
    return _conv(x, P, name, 1, 1, act=False, backend=backend)


def flownet_sd_core(P, x, backend):
    c0 = _conv(x, P, "conv0", 1, 1, backend=backend)
    c1 = _conv(_conv(c0, P, "conv1", 2, 1, backend=backend), P, "conv1_1", 1, 1, backend=backend)
    # the skip tensors of the two finest decoder levels are written straight into their Concat blobs (66 / 66 % of those copies)
    blob2, c2 = _conv_into_concat(_conv(c1, P, "conv2", 2, 1, backend=backend), P, "conv2_1", 1, 1, 64 + 2, backend)
    blob3, c3 = _conv_into_concat(_conv(c2, P, "conv3", 2, 1, backend=backend), P, "conv3_1", 1, 1, 128 + 2, backend)
    blob4, c4 = _skip_conv(_conv(c3, P, "conv4", 2, 1, backend=backend), P, "conv4_1", 1, 1, "deconv4", backend)
    blob5, c5 = _skip_conv(_conv(c4, P, "conv5", 2, 1, backend=backend), P, "conv5_1", 1, 1, "deconv5", backend)
    c6 = _conv(_conv(c5, P, "conv6", 2, 1, backend=backend), P, "conv6_1", 1, 1, backend=backend)
    flow6 = _pf(P, c6, "Convolution1", backend)
    cat5 = _refine_stage(P, (blob5, c5), c6, "deconv5", flow6, "upsample_flow6to5", backend)
    flow5 = _pf(P, _conv(cat5, P, "interconv5", 1, 1, act=False, backend=backend), "Convolution2", backend)
    cat4 = _refine_stage(P, (blob4, c4), cat5, "deconv4", flow5, "upsample_flow5to4", backend)
    flow4 = _pf(P, _conv(cat4, P, "interconv4", 1, 1, act=False, backend=backend), "Convolution3", backend)
    cat3 = _refine_stage(P, (blob3, c3), cat4, "deconv3", flow4, "upsample_flow4to3", backend)
    flow3 = _pf(P, _conv(cat3, P, "interconv3", 1, 1, act=False, backend=backend), "Convolution4", backend)
    cat2 = _refine_stage(P, (blob2, c2), cat3, "deconv2", flow3, "upsample_flow3to2", backend)
    return _pf(P, _conv(cat2, P, "interconv2", 1, 1, act=False, backend=backend), "Convolution5", backend)      # 1/4 resolution, units of 1/SD_FLOW_SCALE px


def fusion_core(P, x, backend):
    # conv0 / conv1_1 write their 64 / 128 channels straight into the 82- / 162-channel Concat blobs of the decoder (387 MB and 191 MB
    # at 768x384 batch 4: the two largest copies of a FlowNet2 forward)
    blob0, c0 = _conv_into_concat(x, P, "conv0", 1, 1, 16 + 2, backend)
    blob1, c1 = _conv_into_concat(_conv(c0, P, "conv1", 2, 1, backend=backend), P, "conv1_1", 1, 1, 32 + 2, backend)
    c2 = _conv(_conv(c1, P, "conv2", 2, 1, backend=backend), P, "conv2_1", 1, 1, backend=backend)
    flow2 = _pf(P, c2, "Convolution5", backend)
    cat1 = _refine_stage(P, (blob1, c1), c2, "deconv1", flow2, "upsample_flow2to1", backend)
    flow1 = _pf(P, _conv(cat1, P, "interconv1", 1, 1, act=False, backend=backend), "Convolution6", backend)
    cat0 = _refine_stage(P, (blob0, c0), cat1, "deconv0", flow1, "upsample_flow1to0", backend)
    return _pf(P, _conv(cat0, P, "interconv0", 1, 1, act=False, backend=backend), "Convolution7", backend)        # full resolution, pixels


def flownet2_deploy_forward(P, img0, img1, backend, mean: Optional[torch.Tensor] = None):
    """Full FlowNet2: raw 0..255 BGR pairs [N,3,H,W] -> flow [N,2,H,W] in pixels.  Four FlowWarp + four ChannelNorm
    + six Resample calls per forward."""
    N, _, H, W = img0.shape
    ah, aw = adapted_size(H, W)
    neg_mean = _const(img0.device, (-0.411, -0.433, -0.45)) if mean is None else (-mean).contiguous()
    if mean is None:
        mean = _const(img0.device, (0.411, 0.433, 0.45))
    if img0.is_cuda and not torch.is_grad_enabled() and hasattr(backend, "resample_slices"):
        return _flownet2_pick_streams(P, img0, img1, backend, neg_mean)
    if (ah, aw) == (H, W) and img0.is_cuda and not torch.is_grad_enabled() and hasattr(backend, "scale_shift"):
        # LINEAR Resample at equal size is the identity: scale and mean in one pass per image (two roundings, like the two layers)
        a = backend.scale_shift(img0, 1.0 / 255.0, neg_mean)
        b = backend.scale_shift(img1, 1.0 / 255.0, neg_mean)
    else:
        a = backend.resample(img0 * (1.0 / 255.0), ah, aw) - mean.view(1, 3, 1, 1)
        b = backend.resample(img1 * (1.0 / 255.0), ah, aw) - mean.view(1, 3, 1, 1)

    def refine_input(flow_q):                       # flow_q: 1/4 resolution, units px/20
        flow = backend.resample(flow_q * FLOW_SCALE, ah, aw)                       # x20, Resample x4 (LINEAR)
        warped = backend.flow_warp(b, flow)                                          # FlowWarp(img1, flow)
        err = backend.channel_norm(a - warped)                                       # ChannelNorm(img0 - warped)
        return flow, warped, err

    flow1_q = flownet_c_core(P, a, b, backend)[2]
    f1, w1, e1 = refine_input(flow1_q)
    flow2_q = flownet_s_core(_Prefixed(P, "net2_"), torch.cat([a, b, w1, f1 * (1.0 / FLOW_SCALE), e1], 1), backend)[2]
    f2, w2, e2 = refine_input(flow2_q)
    flow3_q = flownet_s_core(_Prefixed(P, "net3_"), torch.cat([a, b, w2, f2 * (1.0 / FLOW_SCALE), e2], 1), backend)[2]
    flow_css = backend.resample(flow3_q * FLOW_SCALE, ah, aw, type=1)               # NEAREST into the fusion net (Appendix B)
    flow_sd = backend.resample(flownet_sd_core(_Prefixed(P, "netsd_"), torch.cat([a, b], 1), backend) * SD_FLOW_SCALE, ah, aw, type=1)
    err_css = backend.channel_norm(a - backend.flow_warp(b, flow_css))
    err_sd = backend.channel_norm(a - backend.flow_warp(b, flow_sd))
    fuse_in = torch.cat([a, flow_sd, flow_css, backend.channel_norm(flow_sd), backend.channel_norm(flow_css), err_sd, err_css], 1)
    flow = fusion_core(_Prefixed(P, "fuse_"), fuse_in, backend)
    flow = backend.resample(flow, H, W)
    scale = _const(flow.device, (W / float(aw), H / float(ah)))
    return flow * scale.view(1, 2, 1, 1)


_SD_STREAM = {"mode": "auto", "streams": {}, "picked": {}}


def set_sd_side_stream(mode="auto"):
    """FlowNet-SD only reads the two images: it is independent of the FlowNetC -> S -> S stack until the fusion net.  "on" / True
    runs it on a second HIP stream beside that stack, "off" / False behind it; "auto" (default) times both layouts on the first calls of a
    geometry -- like the kernels time their tile variants -- and keeps the faster (on most boxes the second stream; two boxes of the pool ran
    batch 1 SLOWER with it: 5.65-5.72 against 5.35 ms, while batch 4 gained there too).  Two LONG independent chains share the chip well -- the coarse
    layers of either net leave CUs idle, most of all at batch 1: 5.38 -> 5.00 ms at batch 1 @1024x448, 10.31 -> 10.01 ms at batch 4 @768x384,
    15.20 -> 14.75 ms at batch 4 @1024x448 (round 5) -- where a handful of small kernels beside a chip-filling GEMM did not (the flow heads on a
    second stream, round 4).  Same kernels, same bits."""
    _SD_STREAM["mode"] = {True: "on", False: "off"}.get(mode, mode)
    _SD_STREAM["picked"] = {}


def _flownet2_pick_streams(P, img0, img1, backend, neg_mean):
    """set_sd_side_stream: "on" / "off" as told; "auto": call 0 of a geometry runs with the second stream (lazy initialisation, the kernels'
    own variant timing), calls 1 .. 4 alternate the two layouts between device synchronisations and are timed, then the faster one stays
    (the second stream unless it loses by more than 1 %).  Under graph capture nothing is timed (the second stream, or what was picked)."""
    import time
    mode = _SD_STREAM["mode"]
    run = lambda side: _flownet2_deploy_forward_slices(P, img0, img1, backend, neg_mean, side)
    if mode != "auto":
        return run(mode == "on")
    key = (img0.device, tuple(img0.shape))
    st = _SD_STREAM["picked"].get(key)
    if isinstance(st, bool):
        return run(st)
    if torch.cuda.is_current_stream_capturing():
        return run(True)
    if st is None:
        st = _SD_STREAM["picked"][key] = {"n": 0, True: [], False: []}
    n = st["n"]
    st["n"] += 1
    if n == 0:
        return run(True)
    side = n % 2 == 1
    torch.cuda.synchronize(img0.device)
    t0 = time.perf_counter()
    out = run(side)
    torch.cuda.synchronize(img0.device)
    st[side].append(time.perf_counter() - t0)
    if n >= 4:
        _SD_STREAM["picked"][key] = min(st[True]) <= 1.01 * min(st[False])
        _SD_STREAM.setdefault("timings_ms", {})[key] = {"second_stream": round(min(st[True]) * 1e3, 4), "one_stream": round(min(st[False]) * 1e3, 4)}
    return out


def sd_side_stream_timings():
    """{(device, image shape): {"second_stream": ms, "one_stream": ms}}: what "auto" measured (host clock around one synchronised forward,
    best of two per layout) when it decided -- bench.py prints it so that a box where the second stream loses leaves a record."""
    return dict(_SD_STREAM.get("timings_ms", {}))


def sd_side_stream_picks():
    """{(device, image shape): True / False} of the geometries "auto" has decided (bench.py reports it)."""
    return {k: v for k, v in _SD_STREAM["picked"].items() if isinstance(v, bool)}


def _sd_side_stream(dev, on):
    if not on:
        return None
    st = _SD_STREAM["streams"].get(dev)
    if st is None:
        st = _SD_STREAM["streams"][dev] = torch.cuda.Stream(device=dev)
    return st


def _flownet2_deploy_forward_slices(P, img0, img1, backend, neg_mean, sd_beside=True):
    """The same graph on the GPU backend without its glue passes: the Concat blobs are allocated once and every producer writes its
    channel slice (fn2_*_slices), the Eltwise scalings (x20 in front of a Resample, x0.05 behind it, img0 - warped in front of a
    ChannelNorm) ride in the kernels of their neighbours with the same roundings.  Per forward this removes 7 Concat copies and 19
    element-wise passes over full-resolution blobs; results equal the layer-by-layer graph bit for bit
    (tests/test_gpu_parity.py::test_flownet2_slice_path_is_bitwise_the_layer_graph)."""
    N, _, H, W = img0.shape
    ah, aw = adapted_size(H, W)
    dev = img0.device
    new = lambda c: torch.empty((N, c, ah, aw), device=dev, dtype=torch.float32)
    towers = torch.empty((2 * N, 3, ah, aw), device=dev, dtype=torch.float32)     # FlowNetC's siamese batch [img0 ; img1]
    blob = new(12)            # Concat of net2 / net3: [img0 | img1 | warped img1 | flow / 20 | brightness error]
    pair = new(6)             # Concat of FlowNet-SD: [img0 | img1]
    fuse = new(11)            # Concat of the fusion net: [img0 | flow_sd | flow_css | |flow_sd| | |flow_css| | err_sd | err_css]
    for k, img in enumerate((img0, img1)):
        if (ah, aw) == (H, W):   # LINEAR Resample at equal size is the identity: Eltwise{1/255} and the mean subtraction in one pass
            src, s = img, 1.0 / 255.0
        else:                    # Eltwise{1/255} folded into the Resample, then the mean subtraction
            src, s = backend.resample_slices(img, ah, aw, in_scale=1.0 / 255.0), 1.0
        backend.scale_shift(src, s, neg_mean, out=towers[k * N:(k + 1) * N])
        backend.scale_shift(src, s, neg_mean, out=blob, out_c0=3 * k)
        backend.scale_shift(src, s, neg_mean, out=pair, out_c0=3 * k)
        if k == 0:
            backend.scale_shift(src, s, neg_mean, out=fuse, out_c0=0)
    A, B, WARPED = (blob, 0, 3), (blob, 3, 3), (blob, 6, 3)

    def refine_input(flow_q):                       # flow_q: 1/4 resolution, units px/20 -> channels 6..11 of `blob`
        flow = backend.resample_slices(flow_q, ah, aw, in_scale=FLOW_SCALE, out2=(blob, 9, 2), out2_scale=1.0 / FLOW_SCALE)
        backend.flow_warp_slices(B, flow, out=WARPED)
        backend.channel_norm_slices(A, minus=WARPED, out=(blob, 11, 1))

    side = _sd_side_stream(dev, sd_beside)
    sd_q = None
    if side is not None:                            # FlowNet-SD beside the CSS stack (set_sd_side_stream)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)                      # `pair` is written
        with torch.cuda.stream(side):
            sd_q = flownet_sd_core(_Prefixed(P, "netsd_"), pair, backend)
    flow1_q = flownet_c_core(P, None, None, backend, towers=towers)[2]
    refine_input(flow1_q)
    flow2_q = flownet_s_core(_Prefixed(P, "net2_"), blob, backend)[2]
    refine_input(flow2_q)                           # net2's conv1 has read the blob (stream order): channels 6..11 are rewritten in place
    flow3_q = flownet_s_core(_Prefixed(P, "net3_"), blob, backend)[2]
    if side is not None:
        main.wait_stream(side)
        sd_q.record_stream(main)                    # allocated under the side stream, read (and freed) under this one
    else:
        sd_q = flownet_sd_core(_Prefixed(P, "netsd_"), pair, backend)
    backend.resample_slices(sd_q, ah, aw, type=1, in_scale=SD_FLOW_SCALE, out=(fuse, 3, 2))        # NEAREST into the fusion net (Appendix B)
    backend.resample_slices(flow3_q, ah, aw, type=1, in_scale=FLOW_SCALE, out=(fuse, 5, 2))
    backend.channel_norm_slices((fuse, 3, 2), out=(fuse, 7, 1))
    backend.channel_norm_slices((fuse, 5, 2), out=(fuse, 8, 1))
    backend.flow_warp_slices(B, (fuse, 3, 2), out=WARPED)
    backend.channel_norm_slices(A, minus=WARPED, out=(fuse, 9, 1))
    backend.flow_warp_slices(B, (fuse, 5, 2), out=WARPED)
    backend.channel_norm_slices(A, minus=WARPED, out=(fuse, 10, 1))
    flow = fusion_core(_Prefixed(P, "fuse_"), fuse, backend)
    if (ah, aw) == (H, W):
        return flow               # Resample to an equal size and the Eltwise{1, 1} behind it are identities
    flow = backend.resample(flow, H, W)
    scale = _const(flow.device, (W / float(aw), H / float(ah)))
    return flow * scale.view(1, 2, 1, 1)
