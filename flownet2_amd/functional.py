"""torch.autograd glue over the HIP operators (used by the FlowNet graphs in nets.py).

Forward and backward both run the hand-written HIP kernels through the C ABI; autograd only routes
tensors.  Gradient-stopping layers (Resample, Downsample: AllowBackward() == false in the
reference, resample_layer.hpp:25, downsample_layer.hpp:30) return no gradient.
"""
from __future__ import annotations

import os

import torch

from . import ops


_BATCH_INVARIANT = [False]


def set_batch_invariant(on: bool = True):
    """Make the bits of a sample independent of the batch it is computed in (scripts/run_flownet_many.py: one .flo per pair, whatever
    the batching and the sharding over GPUs -- run-flownet-many.py:27-81).  The own kernels are batch-invariant by construction except
    for the K split of the small-map kernels (fn2_set_batch_invariant fixes it to the one-sample value).  The library convolution that
    remains as the fallback for layer shapes outside FlowNet's families (`fallback_conv2d`: none of the five BASELINE configurations
    reaches it) chooses its kernels -- and with them the summation order -- by problem size, so in this mode it runs sample by sample."""
    _BATCH_INVARIANT[0] = bool(on)
    ops.set_batch_invariant(bool(on))


def batch_invariant() -> bool:
    return _BATCH_INVARIANT[0]


LIBRARY_FALLBACKS = [0]      # calls that left the own kernels (tests and the profile audits assert 0 on the BASELINE configurations)


def _note_fallback(what, x, w):
    LIBRARY_FALLBACKS[0] += 1
    if os.environ.get("FN2_STRICT") == "1":
        raise RuntimeError("flownet2_amd: no own kernel for %s bottom %s weight %s (FN2_STRICT=1)" % (what, tuple(x.shape), tuple(w.shape)))
    if os.environ.get("FN2_TRACE_FALLBACK") == "1":
        print("library fallback: %s bottom %s weight %s" % (what, tuple(x.shape), tuple(w.shape)), flush=True)


def lib_conv2d(x, w, b, stride, pad):
    """Last resort for a Convolution no own kernel serves (a layer shape outside FlowNet's families, or CPU tensors): the library's
    convolution through torch, counted in LIBRARY_FALLBACKS for CUDA tensors; sample by sample in batch-invariant mode."""
    if x.is_cuda:
        _note_fallback("Convolution{stride %d, pad %d}" % (stride, pad), x, w)
    if not _BATCH_INVARIANT[0] or not x.is_cuda:
        return torch.nn.functional.conv2d(x, w, b, stride=stride, padding=pad)
    return torch.cat([torch.nn.functional.conv2d(x[n:n + 1], w, b, stride=stride, padding=pad) for n in range(x.shape[0])], 0)


def lib_conv_transpose2d(x, w, b, stride, pad):
    if x.is_cuda:
        _note_fallback("Deconvolution{stride %d, pad %d}" % (stride, pad), x, w)
    if not _BATCH_INVARIANT[0] or not x.is_cuda:
        return torch.nn.functional.conv_transpose2d(x, w, b, stride=stride, padding=pad)
    return torch.cat([torch.nn.functional.conv_transpose2d(x[n:n + 1], w, b, stride=stride, padding=pad) for n in range(x.shape[0])], 0)


def scale_shift(x, scale, shift=None, out=None, out_c0=0):
    """Deploy head in one pass (no autograd): out[:, out_c0:out_c0+C] = x * scale + shift[c], product and sum rounded separately."""
    with torch.no_grad():
        return ops.scale_shift_forward(x.detach().contiguous(), scale, shift, out=out, out_c0=out_c0)


class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b0, b1, params):
        ctx.params = params
        ctx.save_for_backward(b0, b1)
        return ops.correlation_forward(params, b0, b1)

    @staticmethod
    def backward(ctx, g):
        b0, b1 = ctx.saved_tensors
        d0, d1 = ops.correlation_backward(ctx.params, b0, b1, g.contiguous(),
                                          need0=ctx.needs_input_grad[0], need1=ctx.needs_input_grad[1])
        return d0, d1, None


def correlation(b0, b1, pad=0, kernel_size=1, max_displacement=0, stride_1=1, stride_2=1, correlation_type=ops.MULTIPLY):
    p = ops.corr_params(pad, kernel_size, max_displacement, stride_1, stride_2, correlation_type)
    return _Correlation.apply(b0.contiguous(), b1.contiguous(), p)


class _CorrelationReluInto(torch.autograd.Function):
    """Correlation + ReLU written into a channel slice of the consumer's Concat blob, with autograd: the backward pass undoes the ReLU from the
    activated planes where they lie (csrc/bias_act.hip reads both slices in place) and runs the correlation's own backward kernels."""

    @staticmethod
    def forward(ctx, b0, b1, params, blob, c0, slope):
        ops.correlation_forward(params, b0, b1, out=blob, out_c0=c0, relu=True, negative_slope=slope)
        tc = ops.correlation_out_shape(params, b0.shape[1], b0.shape[2], b0.shape[3])[0]
        ctx.cfg = (params, c0, tc, slope)
        ctx.save_for_backward(b0, b1, blob)
        return blob[:, c0:c0 + tc]

    @staticmethod
    def backward(ctx, g):
        b0, b1, blob = ctx.saved_tensors
        params, c0, tc, slope = ctx.cfg
        gb, g0 = _channel_slice(g)
        d, _ = ops.bias_leaky_relu_backward((blob, c0, tc), (gb, g0, tc), slope, False)
        d0, d1 = ops.correlation_backward(params, b0, b1, d, need0=ctx.needs_input_grad[0], need1=ctx.needs_input_grad[1])
        return d0, d1, None, None, None, None


def correlation_relu_into(b0, b1, out, out_c0, negative_slope, pad=0, kernel_size=1, max_displacement=0, stride_1=1, stride_2=1, training=False):
    """Correlation + ReLU{negative_slope} written into the channel slice [out_c0, out_c0 + topC) of `out`: the cost volume lands
    in the [conv_redir | corr] blob conv3_1 reads, without the separate activation and concat passes.  With autograd: only when the caller
    asks for the graph-carrying form (training=True: the returned slice view carries the graph, nets._ConcatInPlace ties it to the blob);
    otherwise None (the caller builds the unfused graph)."""
    p = ops.corr_params(pad, kernel_size, max_displacement, stride_1, stride_2, ops.MULTIPLY)
    if torch.is_grad_enabled() and (b0.requires_grad or b1.requires_grad):
        if not training:
            return None
        return _CorrelationReluInto.apply(b0.contiguous(), b1.contiguous(), p, out, out_c0, negative_slope)
    return ops.correlation_forward(p, b0.contiguous(), b1.contiguous(), out=out, out_c0=out_c0, relu=True, negative_slope=negative_slope)


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, flow, fill_value):
        ctx.save_for_backward(image, flow)
        return ops.flow_warp_forward(image, flow, fill_value)

    @staticmethod
    def backward(ctx, g):
        image, flow = ctx.saved_tensors
        di, df = ops.flow_warp_backward(image, flow, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (di if ctx.needs_input_grad[0] else None), (df if ctx.needs_input_grad[1] else None), None


def flow_warp(image, flow, fill_value=ops.FILL_ZERO):
    return _FlowWarp.apply(image.contiguous(), flow.contiguous(), fill_value)


def resample(x, height, width, type=ops.LINEAR, antialias=True):
    with torch.no_grad():
        return ops.resample_forward(x.detach().contiguous(), height, width, type, antialias)


def resample_slices(x, height, width, type=ops.LINEAR, antialias=True, in_scale=1.0, out=None, out2=None, out2_scale=1.0):
    """Resample(x * in_scale) written into a channel slice (blob, c0, C); out2 = top * out2_scale into a second slice (no autograd)."""
    with torch.no_grad():
        return ops.resample_forward_slices(x.detach().contiguous(), height, width, type, antialias, in_scale, out, out2, out2_scale)


def flow_warp_slices(image, flow, out=None, fill_value=ops.FILL_ZERO):
    """FlowWarp over channel slices (blob, c0, C) of wider blobs (no autograd: deploy graphs)."""
    with torch.no_grad():
        return ops.flow_warp_forward_slices(image, flow, out, fill_value)


def channel_norm_slices(x, minus=None, out=None):
    """ChannelNorm(x - minus) over channel slices, top into one channel of a wider blob (no autograd: deploy graphs)."""
    with torch.no_grad():
        return ops.channel_norm_forward_slices(x, minus, out)


def downsample(x, top_height, top_width):
    with torch.no_grad():
        return ops.downsample_forward(x.detach().contiguous(), top_height, top_width)


def downsample_ahead(x, sizes):
    """Downsample(x) to every (height, width) of `sizes` on the second HIP stream (the one the weight gradients use), for consumers that
    need the result much later: the ground-truth pyramid of a training step depends on the ground truth alone, so it is issued at the START
    of the step and runs beside the first convolutions instead of between the forward and the backward pass (5 launches, 135 us of a
    FlowNetC step on the critical path).  Returns (tensors, event): the consumer's stream must wait for the event (wait_ahead)."""
    with torch.no_grad():
        x = x.detach().contiguous()
        many = (lambda: ops.downsample_forward_multi(x, sizes)) if ops.downsample_multi_supported(x.shape, sizes) and x.shape[0] > 0 else \
            (lambda: [ops.downsample_forward(x, h, w) for h, w in sizes])        # (one launch for the whole pyramid where its sizes allow)
        if not x.is_cuda:
            return [ops.downsample_forward(x, h, w) for h, w in sizes], None
        main = torch.cuda.current_stream(x.device)
        side = _WGRAD_SIDE["streams"].get(x.device)
        if side is None:
            side = _WGRAD_SIDE["streams"][x.device] = torch.cuda.Stream(device=x.device)
        side.wait_stream(main)                    # x was produced under the main stream
        with torch.cuda.stream(side):
            outs = many()
            ev = torch.cuda.Event()
            ev.record(side)
        x.record_stream(side)
        for t in outs:
            t.record_stream(main)
        return outs, ev


def wait_ahead(event):
    if event is not None:
        torch.cuda.current_stream().wait_event(event)


class _ChannelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        top = ops.channel_norm_forward(x)
        ctx.save_for_backward(x, top)
        return top

    @staticmethod
    def backward(ctx, g):
        x, top = ctx.saved_tensors
        return ops.channel_norm_backward(x, top, g.contiguous())


def channel_norm(x):
    return _ChannelNorm.apply(x.contiguous())


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b0, b1, params):
        loss, ws = ops.l1loss_forward(params, b0, b1)
        ctx.params, ctx.ws = params, ws
        ctx.save_for_backward(b0, b1)
        return loss

    @staticmethod
    def backward(ctx, g):
        b0, b1 = ctx.saved_tensors
        # the C ABI takes the loss weight as a host float, like top[0]->cpu_diff()[0] (l1loss_layer.cu:155)
        d0, d1 = ops.l1loss_backward(ctx.params, b0, b1, float(g), ctx.ws)
        return (d0 if ctx.needs_input_grad[0] else None), (d1 if (b1 is not None and ctx.needs_input_grad[1]) else None), None


def l1_loss(b0, b1=None, l2_per_location=False, l2_prescale_by_channels=False, normalize_by_num_entries=False,
            epsilon=1e-2, plateau=0.0):
    p = ops.l1_params(l2_per_location, l2_prescale_by_channels, normalize_by_num_entries, epsilon, plateau)
    return _L1Loss.apply(b0.contiguous(), b1.contiguous() if b1 is not None else None, p)


class _L1LossMulti(torch.autograd.Function):
    """All loss layers of a net: forward and backward one launch each, the weighted sum included; the upstream gradient stays on the
    device (no host read of a device scalar: _L1Loss's float(g) drains the stream once per layer)."""

    @staticmethod
    def forward(ctx, params, weights, n, *blobs):
        b0, b1 = list(blobs[:n]), list(blobs[n:])
        total, losses, ws = ops.l1loss_forward_multi(params, b0, b1, weights)
        ctx.params, ctx.weights, ctx.n, ctx.ws = params, weights, n, ws
        ctx.save_for_backward(*blobs)
        ctx.mark_non_differentiable(losses)
        return total, losses

    @staticmethod
    def backward(ctx, g, _g_losses):
        n = ctx.n
        blobs = ctx.saved_tensors
        b0, b1 = list(blobs[:n]), list(blobs[n:])
        need1 = any(ctx.needs_input_grad[3 + n:])
        d0, d1 = ops.l1loss_backward_multi(ctx.params, b0, b1, ctx.weights, g.contiguous(), ctx.ws, need1=need1)
        grads = [d if ctx.needs_input_grad[3 + k] else None for k, d in enumerate(d0)]
        grads += [d if ctx.needs_input_grad[3 + n + k] else None for k, d in enumerate(d1)]
        return (None, None, None) + tuple(grads)


def l1_loss_multi(preds, targets, weights, l2_per_location=False, l2_prescale_by_channels=False, normalize_by_num_entries=False,
                  epsilon=1e-2, plateau=0.0):
    """sum_k weights[k] * L1Loss(preds[k], targets[k]) -- the loss layers of a training net -- as (total, per-scale losses)."""
    p = ops.l1_params(l2_per_location, l2_prescale_by_channels, normalize_by_num_entries, epsilon, plateau)
    blobs = [t.contiguous() for t in preds] + [t.contiguous() for t in targets]
    return _L1LossMulti.apply(p, tuple(float(w) for w in weights), len(preds), *blobs)


class _PredictFlow(torch.autograd.Function):
    """predict_flow (Convolution{3,1,1} C -> 2): own forward (csrc/flow_head.hip) and own backward (csrc/flow_head_bwd.hip: weight,
    bias and bottom gradients as streaming kernels over NCHW -- a 2-channel side cannot feed a matrix tile)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.predict_flow_conv_forward(x.contiguous(), weight.contiguous(), bias)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        blob, c0 = _channel_slice(x)
        g = g.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        side = _wgrad_side_stream(g, x, w) if (need_w or need_b) and need_x else None
        if side is None:
            return ops.predict_flow_conv_backward((blob, c0, x.shape[1]), w, g, need_x, need_w, need_b)
        # round 6: only the data gradient is on the critical path of the backward pass; the head's weight / bias gradient (a pass over the
        # whole Concat blob in front of it) runs beside the data-gradient chain like the convolutions' weight gradients
        dx, _, _ = ops.predict_flow_conv_backward((blob, c0, x.shape[1]), w, g, True, False, False)
        main = torch.cuda.current_stream(g.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _, dw, db = ops.predict_flow_conv_backward((blob, c0, x.shape[1]), w, g, False, need_w, need_b)
        for t in (g, blob):
            t.record_stream(side)
        for t in (dw, db):
            if t is not None:
                t.record_stream(main)
        if dw is not None:
            _SIDE_PENDING.append((w, dw.data_ptr()))
        return dx, dw, db


class _UpsampleFlow(torch.autograd.Function):
    """upsample_flow (Deconvolution{4,2,1} 2 -> 2): own forward and own backward (csrc/flow_head_bwd.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, into=None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if into is None:
            return ops.upsample_flow_deconv_forward(x.contiguous(), weight.contiguous(), bias)
        ops.upsample_flow_deconv_forward(x.contiguous(), weight.contiguous(), bias, out=into[0], out_c0=into[1])
        return into[0][:, into[1]:into[1] + 2]          # (see _OwnForwardConv.forward)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.upsample_flow_deconv_backward(x.contiguous(), w, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                                       ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, None


def predict_flow_conv(x, weight, bias=None):
    """predict_flow (Convolution{3,1,1} -> 2 channels): own forward kernel; with autograd active own backward kernels too."""
    run = lambda xx, ww, bb: ops.predict_flow_conv_forward(xx.contiguous(), ww.contiguous(), bb)
    if _needs_grad(x, weight, bias):
        if ops.predict_flow_conv_backward_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3]):
            return _PredictFlow.apply(x, weight, bias)
        return _OwnForwardConv.apply(x, weight, bias, run, 1, 1, 0.0, False, False)
    return run(x, weight, bias)


def upsample_flow_deconv(x, weight, bias=None, out=None, out_c0=0):
    """upsample_flow (Deconvolution{4,2,1} 2 -> 2 channels); `out`: write into that channel slice of a Concat blob (no autograd)."""
    if out is not None and not _needs_grad(x, weight, bias):
        return ops.upsample_flow_deconv_forward(x.contiguous(), weight.contiguous(), bias, out=out, out_c0=out_c0)
    run = lambda xx, ww, bb: ops.upsample_flow_deconv_forward(xx.contiguous(), ww.contiguous(), bb)
    if _needs_grad(x, weight, bias):
        return _UpsampleFlow.apply(x, weight, bias, None if out is None else (out, out_c0))
    return run(x, weight, bias)


class _BiasLeakyReLU(torch.autograd.Function):
    """y -> leaky_relu(y + bias) in place on the convolution output (conv2d's backward does not need its output), with the
    fused backward: one pass for the activation gradient and the bias gradient."""

    @staticmethod
    def forward(ctx, y, bias, negative_slope):
        ctx.mark_dirty(y)
        ops.bias_leaky_relu_(y, bias, negative_slope)
        ctx.slope = negative_slope
        ctx.has_bias = bias is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        d, db = ops.bias_leaky_relu_backward(y, g.contiguous(), ctx.slope, ctx.has_bias and ctx.needs_input_grad[1])
        return d, db, None


def conv_bias_leaky_relu(y, bias, negative_slope=0.1):
    """Bias term + in-place leaky ReLU on a bias-free convolution output: one HIP pass forward, one backward."""
    if not y.is_contiguous():
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        return torch.nn.functional.leaky_relu(y, negative_slope)
    if torch.is_grad_enabled() and (y.requires_grad or (bias is not None and bias.requires_grad)):
        return _BiasLeakyReLU.apply(y, bias, negative_slope)
    return ops.bias_leaky_relu_(y, bias, negative_slope)


def conv_k7s2_relu(x, weight, bias, negative_slope=0.1, relu_chain=None):
    """Stem convolution + bias + leaky ReLU.  Returns None when the fused HIP kernel does not apply (autograd needed or
    unsupported shape): the caller then runs the library convolution."""
    if not ops.conv_k7s2_relu_supported(x.shape[1], x.shape[2], x.shape[3], weight.shape[0]):
        return None
    run = lambda xx, ww, bb: ops.conv_k7s2_relu_forward(xx.contiguous(), ww.contiguous(), bb, negative_slope)
    if _needs_grad(x, weight, bias):
        return _OwnForwardConv.apply(x, weight, bias, run, 2, 3, negative_slope, True, False, None, relu_chain)
    return run(x, weight, bias)


_PACKED = {}     # id(weight tensor) -> (weak reference, _version, packed copy) of FROZEN weights (see _cached)


def invalidate_weight_caches():
    """Drop every packed / transposed weight copy.  The caches notice in-place writes through the tensor itself (`_version`), but not
    writes through `.data` or a checkpoint load into `.data`: call this after such a write (parallel.broadcast_params does)."""
    from . import nets
    _bump_weight_generation()
    _PACKED.clear()
    _PACKED_U.clear()
    _PACKED_D.clear()
    _PACKED_T.clear()
    nets._WT_CACHE.clear()


_WEIGHT_GENERATION = [0]     # bumped by every torch optimizer step (global post-step hook below) and by invalidate_weight_caches()


def weight_generation() -> int:
    return _WEIGHT_GENERATION[0]


def _bump_weight_generation(*_a, **_k):
    _WEIGHT_GENERATION[0] += 1


try:        # every torch.optim.Optimizer.step() of this process, fused ones included (they write the parameters WITHOUT bumping `_version`)
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_step_hook
    _reg_step_hook(_bump_weight_generation)
    _HAVE_STEP_HOOK = True
except Exception:       # pragma: no cover - torch without the global hook: trainable tensors are never cached (round-4 behaviour)
    _HAVE_STEP_HOOK = False


def _cached(cache, key, w, make):
    """make() for weight tensor w, remembered under `key` while w is alive and unwritten.  Writes are noticed through `_version` -- which
    fused optimizers do NOT touch (torch.optim.Adam(fused=True) leaves it at 0: scripts/probes/stale_pack_probe.py, round 4: the cached
    operands of step 1 served every later step).  For a tensor that requires grad the entry therefore also carries the weight GENERATION,
    a counter every torch optimizer step bumps through the global post-step hook above: a trainable parameter is repacked once per
    optimizer step (not once per use), and inference under no_grad with default nn.Parameter weights keeps its packs.  Writes through
    `.data` / a custom optimizer that is not a torch.optim.Optimizer need invalidate_weight_caches()."""
    import weakref
    if w.requires_grad and (not _HAVE_STEP_HOOK or (w.is_cuda and torch.cuda.is_current_stream_capturing())):
        # (under hipGraph capture a trainable weight is packed INSIDE the graph: a replayed optimizer step fires no hook, and a pack left
        # out of the capture would serve every replay the operands of capture time)
        cache.pop(key, None)
        return make()
    gen = _WEIGHT_GENERATION[0] if w.requires_grad else -1
    hit = cache.get(key)
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[3] != gen:
        hit = (weakref.ref(w, lambda _r, k=key: cache.pop(k, None)), w._version, make(), gen)
        cache[key] = hit
    return hit[2]


def _packed_conv_weight(w):
    return _cached(_PACKED, id(w), w, lambda: ops.conv_mfma_pack_weights(w.detach()))


_PACKED_U = {}


def _packed_wino_weight(w):
    return _cached(_PACKED_U, id(w), w, lambda: ops.conv_wino_pack_weights(w.detach()))


_ROUTE_FORCE = [False]


def set_route_force(on: bool = True):
    """Test hook (FN2_ROUTE_FORCE of fn2_conv_route): the Winograd kernel wherever it applies, the small-map kernel whatever the map size."""
    _ROUTE_FORCE[0] = bool(on)


def _conv_mfma_pick(x, weight, stride, pad):
    """Which own kernel serves this layer: "wino", "plane", "direct" or None.  The decision is the library's (fn2_conv_route,
    csrc/conv_route.cpp: thresholds, batch-invariant mode), the same one the Caffe adapter's Convolution gets."""
    Cout, Cin, k, k2 = weight.shape
    if not x.is_cuda or k != k2:
        return None
    return ops.conv_route(x.shape[0], Cin, x.shape[2], x.shape[3], Cout, k, stride, pad, force=_ROUTE_FORCE[0])


def _channel_slice(x):
    """(blob, first channel): x itself, or -- when x is a channel-slice view `blob[:, c0:c0+C]` of a contiguous NCHW blob (a skip
    tensor that was written straight into its Concat blob) -- that blob and the offset, so that the kernels read it in place."""
    if x.is_contiguous():
        return x, 0
    b = x._base
    if (b is not None and b.dim() == 4 and b.is_contiguous() and x.dim() == 4 and x.stride() == b.stride() and x.shape[0] == b.shape[0]
            and x.shape[2:] == b.shape[2:]):
        plane = b.shape[2] * b.shape[3]
        off = x.storage_offset() - b.storage_offset()
        if off % plane == 0 and 0 <= off // plane and off // plane + x.shape[1] <= b.shape[1]:
            return b, off // plane
    return x.contiguous(), 0


def _conv_mfma_run(kind, x, weight, bias, stride, pad, negative_slope, act, out=None, out_c0=0):
    Cout, Cin, k, _ = weight.shape
    blob, c0 = _channel_slice(x)
    if kind == "wino":
        return ops.conv_wino_forward(blob, _packed_wino_weight(weight), bias, Cout, pad, act, negative_slope, out=out, out_c0=out_c0,
                                     in_c0=c0, Cin=Cin)
    if kind == "plane":
        return ops.conv_plane_forward(blob, _packed_conv_weight(weight), bias, Cout, stride, pad, act, negative_slope, out=out, out_c0=out_c0,
                                      in_c0=c0, Cin=Cin, kernel=k)
    return ops.conv_mfma_forward(blob, _packed_conv_weight(weight), bias, Cout, k, stride, pad, act, negative_slope,
                                 out=out, out_c0=out_c0, in_c0=c0, Cin=Cin)


class _OwnForwardConv(torch.autograd.Function):
    """Training form of the fused forward kernels: `runner(x, weight, bias)` is one of the own kernels (convolution or
    deconvolution + bias [+ leaky ReLU] in one launch or one GEMM route); the backward undoes the activation and reduces the bias
    gradient in one fused pass (csrc/bias_act.hip, from the saved OUTPUT) and hands the two convolution gradients to the library
    (aten::convolution_backward = MIOpen's bwd-data / bwd-weights kernels)."""

    @staticmethod
    def forward(ctx, x, weight, bias, runner, stride, pad, negative_slope, act, transposed, into=None, relu_chain=None):
        # autograd does not record inside forward(): the parameter OBJECT goes to the runner, so that the packed-weight caches
        # (keyed on the parameter and its _version) hit until the optimizer writes the weight.
        # into = (blob, first channel): the runner writes its channels into that slice of the consumer's Concat blob (a plain buffer
        # outside the graph; nets._ConcatInPlace ties the slices together) and the output is the slice view
        if into is None:
            y = runner(x, weight, bias)
        else:
            runner(x, weight, bias, into[0], into[1])
            co = weight.shape[1] if transposed else weight.shape[0]
            y = into[0][:, into[1]:into[1] + co]
        ctx.cfg = (stride, pad, negative_slope, act, bias is not None, transposed)
        ctx.into = into
        # relu_chain (round 6, set by the graph builder for a layer pair A -> B where B is A's ONLY consumer): bit 0 on A = "my top_diff arrives
        # already multiplied by my ReLU derivative" (B did it), bit 1 on B = "fold the ReLU derivative of the layer in front into my data gradient"
        ctx.relu_chain, ctx.chain_cell = (int(relu_chain[0]), relu_chain[1]) if relu_chain else (0, None)
        # the activated output is needed for the ReLU mask: a slice of a Concat blob is kept as the BLOB (a saved view comes back from autograd
        # as a plain strided tensor that has forgotten its base: reading it in place needs the blob and the offset)
        ctx.save_for_backward(x, weight, (into[0] if into is not None else y) if act else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        stride, pad, slope, act, has_bias, transposed = ctx.cfg
        if act and ctx.into is not None:
            y = (y, ctx.into[1], w.shape[1] if transposed else w.shape[0])          # (blob, first channel, channels)
        premasked = False
        if ctx.relu_chain & 1:
            # the consumer must have folded this layer's ReLU derivative into the gradient it handed over -- checked, not assumed: a consumer
            # that took another route (a library fallback, a frozen graph) would otherwise leave the gradient unmasked without a sound
            if not ctx.chain_cell.get("masked"):
                raise RuntimeError("relu_chain: the consumer of this layer did not fold the ReLU derivative into its data gradient")
            ctx.chain_cell["masked"] = False
            premasked = True
        gx, gw, db = conv_backward(x, w, y if act else None, g, stride, pad, slope, transposed, bool(ctx.needs_input_grad[0]),
                                   bool(ctx.needs_input_grad[1]), has_bias and ctx.needs_input_grad[2], premasked=premasked,
                                   mask_bottom=(ctx.chain_cell["slope"] if (ctx.relu_chain & 2) else None))
        if ctx.relu_chain & 2:
            ctx.chain_cell["masked"] = True
        return (gx, gw, db) + (None,) * (len(ctx.needs_input_grad) - 3)


_GRAD_SLOT = [None]          # parallel.GradientExchange with a collective: weight tensor -> its slot of the flat all-reduce bucket (or None)


def set_grad_slots(fn=None):
    """fn(weight) -> a fresh fp32 view shaped like `weight` into the flat bucket its gradient travels in (or None).  The weight-gradient
    kernels then write their result THERE (fn2_conv_backward_weights' `weight_diff` pointer) and autograd moves that view into `.grad`:
    the gradient is produced in the bucket instead of being copied into it by the hook (156.7 MB read + written per FlowNetC step)."""
    _GRAD_SLOT[0] = fn


def _grad_slot(w):
    fn = _GRAD_SLOT[0]
    if fn is None or w.grad is not None:          # a pass that ACCUMULATES adds into what is there (AccumulateGrad), not into a fresh slot
        return None
    return fn(w)


_WGRAD_SIDE = {"pixels": 0, "streams": {}}
_SIDE_TASK = [-1]           # autograd graph task the pending entries belong to
_SIDE_PENDING = []          # (weight, address of its gradient computed under the side stream) of the running backward pass -- the address
                            # only: a reference to the tensor would make autograd COPY it into .grad instead of moving it there


def set_wgrad_side_stream(max_pixels: int = 0):
    """Weight gradients beside the data-gradient chain (opt-in; parallel.GradientExchange switches it on for a single-rank job).  Nothing in a
    backward pass reads a layer's weight gradient, and from 1/4 resolution down neither gradient kernel of a layer fills the chip: with
    max_pixels > 0 the weight gradient of every layer whose top and bottom maps have at most that many pixels runs on a second HIP stream
    (FlowNetC train step, batch 8 @448x320: 9.46-9.52 -> 9.33 ms with 36000 = every layer but the stem).  The main stream waits for it when the
    backward pass ends (an autograd-engine callback; outside the engine -- the prototxt executor -- the gradient stays on the main stream).
    Constraints, checked where they can be: the weight has no gradient yet (a pass that ACCUMULATES stays on the main stream) and feeds ONE
    layer of the graph (the engine would add two gradients of a shared weight before the join); nothing reads `.grad` inside the pass
    (GradientExchange's hooks with a collective only COUNT the gradients and launch a bucket's all-reduce on the side stream itself, behind
    the kernels that produce it: side_stream_for_collective).  0 = off (default)."""
    _WGRAD_SIDE["pixels"] = int(max_pixels)


def side_stream_for_collective(device):
    """The stream a gradient bucket's all-reduce has to be ordered behind when weight gradients run on the second stream: that stream, made
    to wait for everything the main stream has been given so far (bias / flow-head / stem gradients are produced there).  The collective
    library orders its own stream behind the CURRENT stream at launch, so the caller launches under `torch.cuda.stream(side)`.  None when
    the second stream is not in use (the collective is then ordered behind the main stream as usual)."""
    if not _WGRAD_SIDE["pixels"]:
        return None
    st = _WGRAD_SIDE["streams"].get(device)
    if st is None:
        return None
    st.wait_stream(torch.cuda.current_stream(device))
    return st


def _wgrad_side_stream(d, x, w):
    if not _WGRAD_SIDE["pixels"] or not d.is_cuda or w.grad is not None or max(d.shape[2] * d.shape[3], x.shape[2] * x.shape[3]) > _WGRAD_SIDE["pixels"]:
        return None
    task = torch._C._current_graph_task_id()
    if task < 0:                # not inside the autograd engine (Net.Backward of the prototxt executor): nobody would join
        return None
    if task != _SIDE_TASK[0]:   # first one of this backward pass: the join rides on its end
        if _SIDE_PENDING:       # a pass that died on the way (an exception in some backward) left entries behind: settle them first
            _SIDE_PENDING.clear()
            for dev, st in _WGRAD_SIDE["streams"].items():
                torch.cuda.current_stream(dev).wait_stream(st)
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        _SIDE_TASK[0] = task
    st = _WGRAD_SIDE["streams"].get(d.device)
    if st is None:
        st = _WGRAD_SIDE["streams"][d.device] = torch.cuda.Stream(device=d.device)
    return st


def join_side_streams():
    """The current stream waits for the weight gradients computed beside it.  A gradient the engine COPIED or ADDED instead of moving it into
    `.grad` was read before it was complete (a weight shared by two layers, a retained graph ...): that is an error, not a silent race."""
    _SIDE_TASK[0] = -1
    if not _SIDE_PENDING:
        return
    for dev, st in _WGRAD_SIDE["streams"].items():
        torch.cuda.current_stream(dev).wait_stream(st)
    pending = list(_SIDE_PENDING)
    _SIDE_PENDING.clear()
    if _GRAD_SLOT[0] is not None:       # a gradient exchange owns `.grad` (it copies what was not produced in its bucket slot -- on the second stream)
        return
    for w, ptr in pending:
        if ptr and w.grad is not None and w.grad.data_ptr() != ptr:
            raise RuntimeError("functional.set_wgrad_side_stream: autograd did not move a weight gradient computed on the second stream into "
                               ".grad as it was (shared weight? retained graph?): the result may have been read early.  Switch it off "
                               "(set_wgrad_side_stream(0)) for this graph.")


def relu_chain_supported(x_shape, w, stride, pad) -> bool:
    """Can a Convolution with this weight on a bottom of this shape fold the ReLU derivative of the layer in front into its data gradient
    (fn2_conv_backward_data_masked: the transposed-convolution route)?"""
    desc = _layer_desc(w, stride, pad, False, x_shape=x_shape)
    if desc is None or not w.is_cuda:
        return False
    route = ops.conv_backward_data_route(desc, False)
    return route != 0 and ops.conv_backward_data_masked_supported(desc, False, route)


def conv_backward(x, w, y, g, stride, pad, slope, transposed, need_x, need_w, need_b, premasked=False, mask_bottom=None):
    """(bottom_diff, weight_diff, bias_diff) of a Convolution / Deconvolution (+ the leaky ReLU folded into it when `y`, the ACTIVATED
    output, is given) from top_diff g -- ConvolutionLayer / DeconvolutionLayer::Backward_gpu (conv_layer.cu:26-60, deconv_layer.cu:27-58)
    behind ReLULayer::Backward_gpu (relu_layer.cu:33-60).  One fused pass undoes the activation and reduces the bias gradient
    (csrc/bias_act.hip, from the saved output), the two convolution gradients run on the library's own routes (fn2_conv_backward_*); a
    geometry without an own kernel goes to the counted last resort (aten::convolution_backward).  Shared by the autograd function above
    and by the prototxt executor's Convolution / Deconvolution mirrors (stock_layers.py)."""
    if y is not None and premasked:
        # the ONE consumer of this layer's output folded this layer's ReLU derivative into its data gradient (relu_chain): g is top_diff of
        # the convolution itself.  The bias gradient comes out of the weight-gradient kernel where that has the fused form (the stem), else
        # from a read-only reduction pass
        d = g.contiguous()
        db = None
        if need_w and need_b and not transposed and d.is_cuda:
            desc = _layer_desc(w, stride, pad, False, x_shape=x.shape)
            if desc is not None and ops.conv_backward_weights_bias_fused(desc, False) and x.is_contiguous():
                gw, db = ops.conv_backward_weights_bias(x, d, desc, False, out=_grad_slot(w))
                gx = _own_bwd_data(d, w, stride, pad, transposed, x.shape) if need_x else None
                if not need_x or gx is not None:
                    return gx, gw, db
        if need_b:
            db = ops.conv_backward_bias(d, d.shape[1])
    elif y is not None:
        # the gradient of a Concat arrives as a channel-slice view of the Concat's top_diff: read in place (no .contiguous() copy)
        gb, g0 = _channel_slice(g)
        if not isinstance(y, tuple):                    # (a tuple: an output that lives in its consumer's Concat blob, read in place too)
            yb, y0 = _channel_slice(y)
            y = (yb, y0, y.shape[1])
        d, db = ops.bias_leaky_relu_backward(y, (gb, g0, g.shape[1]), slope, need_b)
    else:
        g = g.contiguous()
        d, db = g, (g.sum((0, 2, 3)) if need_b else None)
    gx = None
    if need_x and mask_bottom is not None:
        gx = _own_bwd_data_masked(d, w, stride, pad, x, mask_bottom)
    if need_x and gx is None:
        if mask_bottom is not None:
            raise RuntimeError("relu_chain: the data gradient of this layer has no masked form (relu_chain_supported was not asked?)")
        gx = _own_bwd_data(d, w, stride, pad, transposed, x.shape)
    gw = None
    if need_w:
        side = _wgrad_side_stream(d, x, w)
        if side is None:
            gw = _own_bwd_weight(d, x, w, stride, pad, transposed)
        else:
            # the weight gradient of a small map is off the critical path (nothing in the backward pass reads it) and does not fill the chip:
            # beside the data-gradient chain on a second stream; join_side_streams() before the optimizer / the gradient exchange reads it
            main = torch.cuda.current_stream(d.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gw = _own_bwd_weight(d, x, w, stride, pad, transposed)
            if gw is None:
                main.wait_stream(side)
            else:
                for t in (d, x):
                    t.record_stream(side)                 # read under the side stream, freed under the main one
                gw.record_stream(main)
                _SIDE_PENDING.append((w, gw.data_ptr()))
    lib_x, lib_w = need_x and gx is None, need_w and gw is None
    if (lib_x or lib_w) and os.environ.get("FN2_TRACE_BWD") == "1":
        print("bwd on the library: x %s w %s stride %d pad %d transposed %s -> %s%s" % (tuple(x.shape), tuple(w.shape), stride, pad, transposed,
                                                                                      "data " if lib_x else "", "weight" if lib_w else ""), flush=True)
    if lib_x or lib_w:
        if x.is_cuda:       # counted like the forward's last resort: `library_conv_fallbacks: 0` in the bench line covers backward too
            _note_fallback("%s backward{stride %d, pad %d}%s%s" % ("Deconvolution" if transposed else "Convolution", stride, pad,
                                                                   " data" if lib_x else "", " weight" if lib_w else ""), x, w)
        gxl, gwl, _ = torch.ops.aten.convolution_backward(d, x, w, None, [stride, stride], [pad, pad], [1, 1], transposed, [0, 0], 1,
                                                          [lib_x, lib_w, False])
        gx, gw = (gxl if lib_x else gx), (gwl if lib_w else gw)
    return gx, gw, db


def _layer_desc(w, stride, pad, transposed, x_shape=None, d_shape=None):
    """fn2_conv_desc of the layer that owns weight blob w ([Cout, Cin, k, k]; Deconvolution: [Cin, Cout, k, k]) from its bottom shape, or --
    where that is unambiguous (stride 1; Deconvolution{4, 2, 1}) -- from the shape of its top_diff."""
    k = int(w.shape[2])
    if w.shape[3] != k:
        return None
    Cin, Cout = (int(w.shape[0]), int(w.shape[1])) if transposed else (int(w.shape[1]), int(w.shape[0]))
    if x_shape is not None:
        N, H, W = int(x_shape[0]), int(x_shape[2]), int(x_shape[3])
    elif transposed and d_shape[2] % 2 == 0 and d_shape[3] % 2 == 0:
        N, H, W = int(d_shape[0]), int(d_shape[2]) // 2, int(d_shape[3]) // 2
    elif not transposed and stride == 1:
        N, H, W = int(d_shape[0]), int(d_shape[2]) - 1 + k - 2 * pad, int(d_shape[3]) - 1 + k - 2 * pad
    else:
        return None
    if H < 1 or W < 1:
        return None
    return ops.conv_desc(N, Cin, H, W, Cout, k, stride, pad)


def _own_bwd_weight(d, x, w, stride, pad, transposed):
    """weight_diff of a Convolution (ConvolutionLayer::Backward_gpu -> weight_gpu_gemm, conv_layer.cu:40-52) or Deconvolution
    (deconv_layer.cu:36-50, the roles of the two blobs swapped) on the own fp32 MFMA kernels: NCHW in, weight layout out, no layout
    transposes, deterministic.  Which kernel (the stem's taps-on-N kernel, csrc/conv_stem_wgrad.hip, or csrc/conv_wgrad.hip) is the library's
    decision (fn2_conv_backward_weights, csrc/conv_route.cpp -- the Caffe adapter's Backward_gpu calls the same function).  Returns None when
    no own kernel applies (tap classes 1/1, 3/1, 3/2, 4/2, 5/2 and the stem; layers with fewer than 16 channels on either side -- the
    2-channel flow heads -- have kernels of their own)."""
    if not d.is_cuda:
        return None
    desc = _layer_desc(w, stride, pad, transposed, x_shape=x.shape)
    if desc is None or not ops.conv_backward_weights_supported(desc, transposed):
        return None
    if (not transposed and desc.kernel == 7) and not (d.is_contiguous() and x.is_contiguous()):
        d, x = d.contiguous(), x.contiguous()           # the stem kernel reads whole blobs
    xb, x0 = _channel_slice(x)
    db, d0 = _channel_slice(d)
    return ops.conv_backward_weights(xb, db, desc, transposed, out=_grad_slot(w), bottom_c0=x0, top_c0=d0)


_PACKED_T = {}     # data-gradient packings and the deconvolution GEMM operand, keyed like _PACKED


def _cached_pack(cache, w, tag, make):
    return _cached(cache, (id(w), tag), w, make)


def _own_bwd_data(d, w, stride, pad, transposed, x_shape=None):
    """bottom_diff on the own kernels (ConvolutionLayer::Backward_gpu, conv_layer.cu:53-57: backward_gpu_gemm = weight^T x top_diff +
    col2im; DeconvolutionLayer::Backward_gpu, deconv_layer.cu:52-56: forward_gpu_gemm of top_diff).  The kernel is the library's choice
    (fn2_conv_backward_data_route, csrc/conv_route.cpp: Winograd on the rotated weights for 3x3 / 1, the transposed-convolution kernel for the
    stride-2 layers, the 4x4 / 2 convolution for a Deconvolution, the small-map kernels on 10x14 / 5x7 maps, the 1x1 kernel on the transposed
    weight); its operand is packed once per weight version.  Returns None when no own kernel applies."""
    if not d.is_cuda:
        return None
    desc = _layer_desc(w, stride, pad, transposed, x_shape=x_shape, d_shape=d.shape)
    if desc is None:
        return None
    route = ops.conv_backward_data_route(desc, transposed)
    if route == 0:
        return None
    key = (desc.N, desc.Hin, desc.Win)       # the route (and with it the operand's layout) is a function of the layer geometry
    packed = _cached_pack(_PACKED_T, w, ("dgrad", route, bool(transposed), stride, pad) + key,
                          lambda: ops.conv_backward_data_pack_weights(w.detach().contiguous(), desc, transposed, route))
    db, d0 = _channel_slice(d)
    return ops.conv_backward_data(db, packed, desc, transposed, route, top_c0=d0)


def _own_bwd_data_masked(d, w, stride, pad, x, slope):
    """_own_bwd_data with ReLUBackward of the layer in front folded in: x is this layer's bottom = that layer's activated output."""
    desc = _layer_desc(w, stride, pad, False, x_shape=x.shape)
    if desc is None or not d.is_cuda:
        return None
    route = ops.conv_backward_data_route(desc, False)
    if route == 0 or not ops.conv_backward_data_masked_supported(desc, False, route):
        return None
    key = (desc.N, desc.Hin, desc.Win)
    packed = _cached_pack(_PACKED_T, w, ("dgrad", route, False, stride, pad) + key,
                          lambda: ops.conv_backward_data_pack_weights(w.detach().contiguous(), desc, False, route))
    db, d0 = _channel_slice(d)
    xb, x0 = _channel_slice(x)
    return ops.conv_backward_data_masked(db, packed, desc, False, route, xb, slope, top_c0=d0, data_c0=x0)


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def conv_mfma_relu(x, weight, bias, stride, pad, negative_slope=0.1, act=True, out=None, out_c0=0, relu_chain=None):
    """Convolution + bias (+ leaky ReLU) as ONE MFMA kernel, NCHW in and out, optionally written into a channel slice of `out`:
    Winograd F(2x2, 3x3) for 3x3 / stride 1 / pad 1 (csrc/conv_wino.hip), the direct kernel otherwise (csrc/conv_mfma.hip).
    With autograd active the same forward runs inside an autograd function (library backward).  Returns None when neither kernel
    applies (unsupported geometry, too little work to fill the chip, disabled): the caller then runs the library convolution."""
    kind = _conv_mfma_pick(x, weight, stride, pad)
    if kind is None:
        return None
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        run = lambda xx, ww, bb, o=None, o0=0: _conv_mfma_run(kind, xx, ww, bb, stride, pad, negative_slope, act, o, o0)
        return _OwnForwardConv.apply(x, weight, bias, run, stride, pad, negative_slope, act, False, None if out is None else (out, out_c0), relu_chain)
    return _conv_mfma_run(kind, x, weight, bias, stride, pad, negative_slope, act, out, out_c0)


def _no_grad_needed(*ts):
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts))


_PACKED_D = {}


def _packed_deconv_weight(w):
    return _cached(_PACKED_D, id(w), w, lambda: ops.deconv_plane_pack_weights(w.detach().contiguous()))


def deconv_mfma_relu(x, weight, bias, negative_slope=0.1, act=True, out=None, out_c0=0):
    """Deconvolution{4, 2, 1} + bias (+ leaky ReLU) as ONE MFMA kernel (csrc/conv_plane.hip, one output parity class per wave), NCHW in
    and out, optionally written into a channel slice of `out` (the consumer's Concat blob).  weight: Caffe's [Cin, Cout, 4, 4] blob.
    The library routes (fn2_deconv_route): on the FlowNet shapes the GEMM (own 1x1 MFMA kernel) + our col2im pass is 5-25 % faster than this
    kernel (profiles/r02_deconv_bench_flownetc.txt), so it serves the planes the GEMM kernel does not take (5x7) .  Returns None when the
    library picks the GEMM route, the kernel does not apply, or a gradient is needed: the caller then takes the GEMM + col2im route."""
    if not x.is_cuda or _needs_grad(x, weight, bias):
        return None
    Cin, Cout = weight.shape[:2]
    if tuple(weight.shape[2:]) != (4, 4) or ops.deconv_route(x.shape[0], Cin, x.shape[2], x.shape[3], Cout) != "plane":
        return None
    blob, c0 = _channel_slice(x)
    return ops.deconv_plane_forward(blob, _packed_deconv_weight(weight), bias, Cout, act, negative_slope, out=out, out_c0=out_c0, in_c0=c0, Cin=Cin)


def _deconv_gemm_supported(x, cout, kernel):
    return x.is_cuda and (cout * kernel * kernel) % 32 == 0 and ops.conv_mfma_supported(x.shape[1], x.shape[2], x.shape[3], cout * kernel * kernel, 1, 1, 0)


def deconv_gemm_relu(x, weight_t, bias, cout, kernel=4, stride=2, pad=1, negative_slope=0.1, weight=None, out=None, out_c0=0):
    """Deconvolution + bias + leaky ReLU as the reference computes it -- weight^T x bottom (base_conv_layer.cpp:375-384: one GEMM, here the
    own 1x1 MFMA kernel), then col2im -- with the bias and activation folded into our col2im pass.
    weight_t = weight.view(Cin, Cout*k*k).t().contiguous() (cached by the caller; rebuilt from `weight` when autograd is active, inside
    _OwnForwardConv).  Returns None if the kernel does not apply, or autograd is needed and `weight` was not given."""
    N, Cin, H, W = x.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + kernel, (W - 1) * stride - 2 * pad + kernel
    gemm_ok = _deconv_gemm_supported(x, cout, kernel)
    # planes the 1x1 / GEMM kernel does not take (deconv5: 5x7) are the small-map deconvolution kernel's -- no column matrix at all
    plane_ok = bool(x.is_cuda and (kernel, stride, pad) == (4, 2, 1) and ops.deconv_plane_supported(N, Cin, H, W, cout))
    if not gemm_ok and not plane_ok:
        return None

    def run_t(xx, wt, bb, out=None, out_c0=0, ww=None):
        # weight^T x bottom as a 1x1 convolution with Cout * k * k output channels on the own MFMA kernel (csrc/conv_mfma.hip,
        # kernel_size 1): the column matrix [N, Cout*k*k, H*W].  Its operand is packed from the transposed copy wt, or (ww given: the
        # layer's own [Cin][Cout][k][k] blob) straight from the blob through the strided view -- no transposed copy at all
        blob, c0 = _channel_slice(xx)
        M = cout * kernel * kernel
        if ww is not None:
            pw = _cached_pack(_PACKED_T, ww, "deconv-gemm", lambda: ops.conv_mfma_pack_weights_view(ww.detach().contiguous(), M, Cin, 1, M, Cin, 1, M))
        else:
            pw = _cached_pack(_PACKED_T, wt, "deconv-gemm", lambda: ops.conv_mfma_pack_weights(wt.detach().reshape(M, Cin, 1, 1)))
        col = ops.conv_mfma_forward(blob, pw, None, cout * kernel * kernel, 1, 1, 0, False, 0.0, in_c0=c0, Cin=Cin).view(N, cout * kernel * kernel, H * W)
        return ops.col2im_bias_relu_forward(col, bb, N, cout, Ho, Wo, kernel, pad, stride, True, negative_slope, out=out, out_c0=out_c0)

    if _needs_grad(x, weight_t, bias, weight):
        if weight is None:
            return None
        def run(xx, ww, bb, o=None, o0=0):
            # ww is the parameter itself (autograd does not record inside _OwnForwardConv.forward): its transposed view and the packed
            # operands are cached on it until the optimizer writes it.  Planes whose size is no multiple of 4 (deconv5: 5x7) are not the
            # 1x1 / GEMM kernel's: the small-map deconvolution kernel computes them without a column matrix (and without a library GEMM)
            if plane_ok and ((H * W) % 4 != 0 or not gemm_ok):
                blob, c0 = _channel_slice(xx)
                return ops.deconv_plane_forward(blob, _packed_deconv_weight(ww), bb, cout, True, negative_slope, out=o, out_c0=o0, in_c0=c0, Cin=Cin)
            return run_t(xx, None, bb, o, o0, ww=ww)
        return _OwnForwardConv.apply(x, weight, bias, run, stride, pad, negative_slope, True, True, None if out is None else (out, out_c0))
    if out is not None:                # the col2im pass writes straight into the consumer's Concat blob
        return None if not gemm_ok else run_t(x, weight_t, bias, out, out_c0)
    if not gemm_ok:                    # (inference reaches the small-map kernel through deconv_mfma_relu before it comes here)
        if weight is None:
            return None
        blob, c0 = _channel_slice(x)
        return ops.deconv_plane_forward(blob, _packed_deconv_weight(weight), bias, cout, True, negative_slope, in_c0=c0, Cin=Cin)
    return run_t(x, weight_t, bias)
