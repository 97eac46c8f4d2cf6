"""fp64 evaluation of the FlowNetC TRAINING step's loss and parameter gradients -- the comparator of BASELINE config 4 at its own size.

TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's checker legs import it; the product path never does).

What it is: the graph of flownet2_amd.nets.flownet_c_core + multiscale_loss -- the very function the training step runs -- evaluated in
torch.float64 with autograd, through a backend that owns NONE of the product's kernels:
  * Convolution / Deconvolution / ReLU: torch's float64 conv2d / conv_transpose2d / leaky_relu (im2col + DGEMM: the reference's own
    algorithm, ConvolutionLayer::Forward_gpu / Backward_gpu, conv_layer.cu:8-60, base_conv_layer.cpp:325-393, at twice the precision);
  * Correlation: the sum over channels of shifted products, displacement by displacement, and its analytic gradient
    (CorrelateData / CorrelateDataBackward0/1, correlation_layer.cu:45-249) in float64;
  * L1Loss{l2_per_location, normalize_by_num_entries}: the formula of l1loss_layer.cu:67-190 (NaN mask, sqrt(sum d^2 + eps), masked
    pixels still contribute sqrt(eps), division by the number of valid entries / channels) with autograd;
  * Downsample of the ground truth (no gradient: downsample_layer.hpp:30): the C oracle's restatement (downsample_layer.cu:15-72).
fp64 rounding (1e-16) is eight orders below the fp32 path's, so this is a PINNED comparator: there is no kernel choice, no
non-determinism and no tolerance to tune on its side; whatever the product's fp32 gradients differ from it by is the product's error.
Runs on whatever device the tensors are on (float64 on the GPU box's MI355X: a few seconds at batch 8 @448x320; CPU for small cases)."""
from __future__ import annotations

import types

import numpy as np
import torch


class _Corr64(torch.autograd.Function):
    """Correlation{kernel_size 1, stride_1 1} in the tensors' own dtype: top[n, (q, o), y, x] = 1/C sum_c P0[n, c, y + md, x + md] *
    P1[n, c, y + md + q s2, x + md + o s2] on the zero-padded maps (correlation_layer.cu:45-114); gradients per displacement
    (correlation_layer.cu:117-249).  Nothing but the two bottoms is saved: the backward loop recomputes the slices."""

    @staticmethod
    def forward(ctx, b0, b1, pad, md, s2):
        N, C, H, W = b0.shape
        th, tw = H + 2 * pad - 2 * md, W + 2 * pad - 2 * md
        ngr = md // s2
        P0 = torch.nn.functional.pad(b0, (pad, pad, pad, pad))
        P1 = torch.nn.functional.pad(b1, (pad, pad, pad, pad))
        a = P0[:, :, md:md + th, md:md + tw]
        top = b0.new_empty((N, (2 * ngr + 1) ** 2, th, tw))
        k = 0
        for q in range(-ngr, ngr + 1):
            for o in range(-ngr, ngr + 1):
                top[:, k] = (a * P1[:, :, md + q * s2:md + q * s2 + th, md + o * s2:md + o * s2 + tw]).sum(1) / C
                k += 1
        ctx.cfg = (pad, md, s2, th, tw, ngr)
        ctx.save_for_backward(b0, b1)
        return top

    @staticmethod
    def backward(ctx, g):
        b0, b1 = ctx.saved_tensors
        pad, md, s2, th, tw, ngr = ctx.cfg
        N, C, H, W = b0.shape
        P0 = torch.nn.functional.pad(b0, (pad, pad, pad, pad))
        P1 = torch.nn.functional.pad(b1, (pad, pad, pad, pad))
        d0, d1 = torch.zeros_like(P0), torch.zeros_like(P1)
        a = P0[:, :, md:md + th, md:md + tw]
        k = 0
        for q in range(-ngr, ngr + 1):
            for o in range(-ngr, ngr + 1):
                gk = g[:, k:k + 1] / C
                ys, xs = md + q * s2, md + o * s2
                d0[:, :, md:md + th, md:md + tw] += gk * P1[:, :, ys:ys + th, xs:xs + tw]
                d1[:, :, ys:ys + th, xs:xs + tw] += gk * a
                k += 1
        crop = lambda t: t[:, :, pad:pad + H, pad:pad + W].contiguous()
        return crop(d0), crop(d1), None, None, None


def _l1_loss(b0, b1, l2_per_location=False, normalize_by_num_entries=False, epsilon=1e-2):
    d = b0 - b1
    mask = ~torch.isnan(d)                                                  # l1loss_layer.cu:11-18 (x == x)
    norm = mask.sum().to(d.dtype) / d.shape[1] if normalize_by_num_entries else torch.tensor(float(d.shape[0]), dtype=d.dtype, device=d.device)
    d = torch.where(mask, d, torch.zeros_like(d))
    if l2_per_location:
        return torch.sqrt((d * d).sum(1) + epsilon).sum() / norm            # masked pixels still add sqrt(eps): l1loss_layer.cu:93-119
    return d.abs().sum() / norm


def backend64():
    """The minimal backend nets.flownet_c_core / multiscale_loss need (everything else falls to torch's own ops in the tensors' dtype)."""
    import oracle

    def correlation(b0, b1, pad=0, kernel_size=1, max_displacement=0, stride_1=1, stride_2=1):
        assert kernel_size == 1 and stride_1 == 1, "fp64 comparator: FlowNetC's correlation (kernel_size 1, stride_1 1)"
        return _Corr64.apply(b0, b1, pad, max_displacement, stride_2)

    def downsample(x, top_height, top_width):
        # ground truth only (no gradient); fp32 in, the oracle's restatement, back in the caller's dtype
        out = oracle.downsample_forward(x.detach().cpu().numpy().astype(np.float32), top_height, top_width)
        return torch.from_numpy(out).to(device=x.device, dtype=x.dtype)

    be = types.SimpleNamespace()
    be.correlation = correlation
    be.downsample = downsample
    be.l1_loss = lambda b0, b1, l2_per_location=False, normalize_by_num_entries=False: _l1_loss(b0, b1, l2_per_location, normalize_by_num_entries)
    return be


class record_relu_branches:
    """Context manager around a forward pass of nets.flownet_c_core in ANY backend: records, in execution order, which side of the kink
    every leaky-ReLU output of the graph is on (the activated outputs of nets._conv / nets._deconv with act = True, and the plain
    F.leaky_relu calls of the graph itself: the correlation's ReLU).  `branches` is the list flownetc_train_reference(masks=...) takes.

    Why: a leaky ReLU is piecewise linear.  An fp32 and an fp64 evaluation of the same net disagree about the SIGN of the few
    pre-activations that are within rounding of zero (about one in 10^6: a handful among the 5 M outputs of conv3_1 at batch 8), and each
    such unit changes its whole upstream gradient by a factor of ten -- 10^-3 in the relative L2 of a layer's gradient, whatever the
    kernels' accuracy (measured in round 4: the library's fp32 kernels sit at 1.3e-3 from the plain fp64 graph on conv1's weights).
    Evaluating the fp64 comparator on the branch the fp32 run took removes that term: what is left is rounding."""

    def __enter__(self):
        import torch.nn.functional as F
        from flownet2_amd import nets
        self.branches = []
        self._oc, self._od, self._lr = nets._conv, nets._deconv, F.leaky_relu
        depth = [0]

        def conv(x, P, name, stride, pad, act=True, backend=None, **kw):
            depth[0] += 1
            try:
                y = self._oc(x, P, name, stride, pad, act, backend, **kw)
            finally:
                depth[0] -= 1
            if act:
                self.branches.append((name, (y.detach() > 0).cpu()))
            return y

        def deconv(x, P, name, act=True, backend=None):
            depth[0] += 1
            try:
                y = self._od(x, P, name, act, backend)
            finally:
                depth[0] -= 1
            if act:
                self.branches.append((name, (y.detach() > 0).cpu()))
            return y

        def leaky_relu(x, negative_slope=0.01, inplace=False):
            if depth[0] == 0:                                   # a ReLU of the graph itself (not one inside a convolution's fallback path)
                self.branches.append(("relu", (x.detach() > 0).cpu()))
            return self._lr(x, negative_slope, inplace)
        from flownet2_amd import functional as Fn
        self._cri = Fn.correlation_relu_into

        def correlation_relu_into(b0, b1, out, out_c0, negative_slope, **kw):
            r = self._cri(b0, b1, out, out_c0, negative_slope, **kw)
            if r is not None:                                   # inference-style fused route (frozen towers): the activated planes sit in the blob
                nch = (2 * (kw.get("max_displacement", 0) // kw.get("stride_2", 1)) + 1) ** 2
                self.branches.append(("relu", (out[:, out_c0:out_c0 + nch].detach() > 0).cpu()))
            return r
        Fn.correlation_relu_into = correlation_relu_into
        # the layers that write straight into a refinement stage's Concat blob (nets._conv_into_concat, nets._stage_deconv: inference, and
        # since round 5 the training graph too) do not pass through nets._conv / nets._deconv: record them where they run
        self._cic, self._sd = nets._conv_into_concat, nets._stage_deconv

        def conv_into_concat(x, P, name, stride, pad, extra_channels, backend, **kw):
            blob, y = self._cic(x, P, name, stride, pad, extra_channels, backend, **kw)
            if blob is not None:                                # (else the fallback went through nets._conv above)
                self.branches.append((name, (y.detach() > 0).cpu()))
            return blob, y

        def stage_deconv(P, x, dname, blob, cs, cd, backend, **kw):
            d = self._sd(P, x, dname, blob, cs, cd, backend, **kw)
            if d is not None:                                   # (else the copy fallback went through nets._deconv above)
                self.branches.append((dname, (blob[:, cs:cs + cd].detach() > 0).cpu()))
            return d
        self._ci = nets._conv_into

        def conv_into(x, P, name, stride, pad, blob, c0, backend):
            y = self._ci(x, P, name, stride, pad, blob, c0, backend)
            if y is not None:
                co = P[name + ".w"].shape[0]
                self.branches.append((name, (blob[:, c0:c0 + co].detach() > 0).cpu()))
            return y
        nets._conv_into = conv_into
        nets._conv_into_concat, nets._stage_deconv = conv_into_concat, stage_deconv
        nets._conv, nets._deconv, F.leaky_relu = conv, deconv, leaky_relu
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        from flownet2_amd import nets
        from flownet2_amd import functional as Fn
        nets._conv, nets._deconv, F.leaky_relu = self._oc, self._od, self._lr
        nets._conv_into_concat, nets._stage_deconv = self._cic, self._sd
        nets._conv_into = self._ci
        Fn.correlation_relu_into = self._cri
        return False


def flownetc_train_reference(P, img0, img1, gt, device=None, mean=0.43, masks=None, dtype=torch.float64):
    """loss (float) and {parameter name: float64 gradient on the CPU} of one FlowNetC training step exactly as bench.py --mode train
    states it: pre-processing im / 255 - mean, nets.flownet_c_core, nets.multiscale_loss against `gt` (NaN = no ground truth).
    P: {name: fp32 tensor}; img0 / img1: raw [N, 3, H, W]; gt: [N, 2, H, W].
    masks = record_relu_branches(...).branches of another run: every leaky ReLU takes the branch recorded there instead of the sign of
    its own input (y = x * (1 | slope) by the recorded mask): the fp64 value and gradient of the piecewise-linear function that run evaluated.
    dtype = torch.float32 turns the same torch graph into the LIBRARY-fp32 yardstick (torch's fp32 conv2d / conv_transpose2d = MIOpen on the
    GPU box, none of the product's kernels): how far a stock fp32 implementation sits from the fp64 graph on the same inputs."""
    import torch.nn.functional as F
    from flownet2_amd import nets
    dev = torch.device(device) if device is not None else img0.device
    P64 = {k: v.detach().to(device=dev, dtype=dtype).requires_grad_(True) for k, v in P.items()}
    i0, i1 = (im.detach().to(device=dev, dtype=dtype) for im in (img0, img1))
    g = gt.detach().to(device=dev, dtype=dtype)
    be = backend64()
    pre = [(im * (1.0 / 255.0)) - mean for im in (i0, i1)]
    lr, oc, od = F.leaky_relu, nets._conv, nets._deconv
    todo = dict(masks) if masks is not None else None          # by layer name ("relu": the correlation's): the two runs may order them differently
    if masks is not None:
        assert len(todo) == len(masks), "a layer name occurs twice among the recorded branches"
    current = [None]

    def named(fn):
        def wrapped(x, P_, name, *a, **k):
            prev, current[0] = current[0], name
            try:
                return fn(x, P_, name, *a, **k)
            finally:
                current[0] = prev
        return wrapped

    def pinned(x, negative_slope=0.01, inplace=False):
        name = current[0] or "relu"
        m = todo.pop(name)
        assert tuple(m.shape) == tuple(x.shape), (name, tuple(m.shape), tuple(x.shape))
        return x * torch.where(m.to(x.device), 1.0, float(negative_slope)).to(x.dtype)
    if todo is not None:
        F.leaky_relu, nets._conv, nets._deconv = pinned, named(oc), named(od)
    try:
        loss = nets.multiscale_loss(nets.flownet_c_core(P64, pre[0], pre[1], be), g, be)
    finally:
        F.leaky_relu, nets._conv, nets._deconv = lr, oc, od
    assert not todo, "recorded ReLU branches left over: the two graphs differ"
    loss.backward()
    return float(loss.detach()), {k: v.grad.detach().cpu() for k, v in P64.items() if v.grad is not None}


def relu_sign_flips(branches_a, branches_b):
    """(units whose leaky ReLU took different branches in two recorded runs, units in all) -- by layer name, as record_relu_branches lists them."""
    a, b = dict(branches_a), dict(branches_b)
    assert a.keys() == b.keys(), (sorted(a), sorted(b))
    return int(sum(int((a[k] != b[k]).sum()) for k in a)), int(sum(a[k].numel() for k in a))


def grad_agreement(grads, ref):
    """Per-parameter relative L2 error ||g - r|| / ||r|| (float64), the same over all parameters together, and the worst parameter."""
    rel, num, den = {}, 0.0, 0.0
    for k, r in ref.items():
        d = grads[k].detach().cpu().double() - r
        n2, d2 = float(r.pow(2).sum()), float(d.pow(2).sum())
        rel[k] = (d2 / max(n2, 1e-300)) ** 0.5
        num, den = num + d2, den + n2
    worst = max(rel, key=rel.get)
    return {"all": (num / den) ** 0.5, "median": float(np.median(list(rel.values()))), "worst": rel[worst], "worst_name": worst, "per_param": rel}
