"""Mirrors of the STOCK Caffe layers a FlowNet2 prototxt contains, on the fast paths of this repository: the prototxt executor
(flownet2_amd.net.Net) creates them by `type:` string through the same LayerRegistry as the custom layers (layer_factory.hpp:75-84).

    Convolution    conv_layer.cpp:8-40, base_conv_layer.cpp:15-254 (LayerSetUp / Reshape), conv_layer.cu:8-23
    Deconvolution  deconv_layer.cpp:8-45 (weight blob [Cin, Cout/g, kh, kw], base_conv_layer.cpp:125-139), deconv_layer.cu:8-26
    ReLU           relu_layer.cu:8-27 (negative_slope), in place
    Eltwise        eltwise_layer.cpp:12-65, eltwise_layer.cu:34-66 (SUM with coeff, PROD, MAX; ONE bottom allowed: eltwise_layer.hpp:29)
    Concat / Slice concat_layer.cpp:24-65 / slice_layer.cpp:24-88
    Silence, Input silence_layer.cu, input_layer.cpp

Backward_gpu of every class follows the reference's (conv_layer.cu:26-60, deconv_layer.cu:27-58, relu_layer.cu:33-60, eltwise_layer.cu:68-118,
concat_layer.cu:47-74, slice_layer.cu:49-75, split_layer.cu:18-33, silence_layer.cu:16-27): top diffs in, bottom diffs OVERWRITTEN, parameter
diffs ACCUMULATED -- Net.Backward (net.py) steps a TRAIN prototxt through them; the gradient kernels are the ones nets.py's autograd graph
uses (functional.conv_backward and the flow-head kernels).  Convolution / Deconvolution call the SAME routing functions as nets.py
(nets.conv_forward / deconv_forward and the flow-head kernels), so a prototxt-built net computes the same bits as nets.deploy_forward.
A ReLU that directly follows a Convolution / Deconvolution in place is folded into that layer by the executor (`fused_relu_`): one
kernel, as in nets.py.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import layers as L
from .layers import Blob, CHECK, CheckError, Layer


def _first(v, default):
    if v is None:
        return default
    if isinstance(v, list):
        return v[0] if v else default
    return v


def fill_blob(blob: Blob, filler: dict, seed_name: str = ""):
    """include/caffe/filler.hpp: constant (:25-43), diagonal (:264-290: blob[n][n][:, :] = diag_val[n] / kernel_area, else 0).  The random
    fillers (msra, xavier, gaussian, uniform) exist to be overwritten by CopyTrainedLayersFrom in a deploy net: they draw from a
    torch generator seeded by the layer name (boost's stream is not reproducible)."""
    t = str(filler.get("type", "constant"))
    shape = blob.shape()
    if t == "constant":
        blob.data = torch.full(shape, float(filler.get("value", 0.0)), dtype=torch.float32, device=blob.device)
    elif t == "diagonal":
        num, ch = shape[0], shape[1] if len(shape) > 1 else 1
        area = 1
        for s in shape[2:]:
            area *= s
        w = torch.zeros(shape, dtype=torch.float32)
        dv = list(filler.get("diag_val", []))
        for n in range(min(num, ch)):
            w[n, n] = (float(dv[n]) if n < len(dv) else 1.0) / float(area)
        blob.data = w.to(blob.device)
    elif t in ("msra", "xavier", "gaussian", "uniform", "positive_unitball", "bilinear"):
        import zlib
        g = torch.Generator().manual_seed(zlib.crc32(seed_name.encode()) & 0x7fffffff)
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        std = float(filler.get("std", 0.0)) if t == "gaussian" else (2.0 / max(1, fan_in)) ** 0.5
        blob.data = (torch.randn(shape, generator=g) * std).to(blob.device)
    else:
        raise CheckError("Unknown filler name: " + t)                                   # filler.hpp:319


class _ConvBase(Layer):
    """BaseConvolutionLayer::LayerSetUp / Reshape (base_conv_layer.cpp:15-254): square or h / w kernels, pad, stride; group 1 and
    dilation 1 only (all a FlowNet uses)."""
    transposed = False

    def MinBottomBlobs(self): return 1
    def MinTopBlobs(self): return 1
    def EqualNumBottomTopBlobs(self): return True                                        # base_conv_layer.hpp:29

    def _geom(self):
        cp = self.layer_param_.convolution_param
        CHECK("num_output" in cp, "num_output is required")
        kh = int(cp["kernel_h"]) if "kernel_h" in cp else int(_first(cp.get("kernel_size"), 0))
        kw = int(cp["kernel_w"]) if "kernel_w" in cp else int(_first(cp.get("kernel_size"), 0))
        CHECK(kh > 0 and kw > 0, "Filter dimensions cannot be zero.")                    # base_conv_layer.cpp:45
        ph = int(cp["pad_h"]) if "pad_h" in cp else int(_first(cp.get("pad"), 0))
        pw = int(cp["pad_w"]) if "pad_w" in cp else int(_first(cp.get("pad"), 0))
        sh = int(cp["stride_h"]) if "stride_h" in cp else int(_first(cp.get("stride"), 1))
        sw = int(cp["stride_w"]) if "stride_w" in cp else int(_first(cp.get("stride"), 1))
        CHECK(int(cp.get("group", 1)) == 1, "group > 1 is not supported by this mirror")
        CHECK(int(_first(cp.get("dilation"), 1)) == 1, "dilation > 1 is not supported by this mirror")
        CHECK(kh == kw and ph == pw and sh == sw, "only square kernels / pads / strides are supported by this mirror")
        return int(cp["num_output"]), kh, ph, sh, bool(cp.get("bias_term", True))

    def LayerSetUp(self, bottom, top):
        self.num_output_, self.kernel_, self.pad_, self.stride_, self.bias_term_ = self._geom()
        self.fused_relu_tops_ = {}              # top index -> negative slope of an in-place ReLU folded in by the executor
        cin = bottom[0].channels()
        cp = self.layer_param_.convolution_param
        if not self.blobs_:                                                               # base_conv_layer.cpp:125-152
            wshape = [cin, self.num_output_, self.kernel_, self.kernel_] if self.transposed else [self.num_output_, cin, self.kernel_, self.kernel_]
            w = Blob(*wshape, device=bottom[0].device)
            fill_blob(w, cp.get("weight_filler", {}), self.layer_param_.name + ".w")
            self.blobs_ = [w]
            if self.bias_term_:
                b = Blob(self.num_output_, device=bottom[0].device)
                fill_blob(b, cp.get("bias_filler", {}), self.layer_param_.name + ".b")
                self.blobs_.append(b)

    def _out_hw(self, h, w):
        k, p, s = self.kernel_, self.pad_, self.stride_
        if self.transposed:
            return s * (h - 1) + k - 2 * p, s * (w - 1) + k - 2 * p                       # deconv_layer.cpp:8-22
        return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1                        # conv_layer.cpp:8-23

    def Reshape(self, bottom, top):
        cin = self.blobs_[0].shape(0) if self.transposed else self.blobs_[0].shape(1)
        CHECK(bottom[0].channels() == cin, "Input size incompatible with convolution kernel.")   # base_conv_layer.cpp:191-192
        for b in bottom[1:]:                                                                     # base_conv_layer.cpp:194-197
            CHECK(b.shape() == bottom[0].shape(), "All inputs must have the same shape.")
        ho, wo = self._out_hw(bottom[0].height(), bottom[0].width())
        for t in top:                                                                            # one weight blob, applied to every bottom
            t.Reshape(bottom[0].num(), self.num_output_, ho, wo)

    @property
    def fused_relu_(self):
        """Negative slope of the folded ReLU when EVERY top has one with the same slope (the only case the single-top code paths and the
        batched siamese launch ask about), else None."""
        v = set(self.fused_relu_tops_.get(i) for i in range(len(self.layer_param_.top)))
        return next(iter(v)) if len(v) == 1 and None not in v else None

    def _slope(self, i):
        return self.fused_relu_tops_.get(i)

    def _forward_all(self, bottom, top, one):
        """`one(x, slope) -> y` on every (bottom, top) pair (base_conv_layer.cpp: the loop over bottom.size() in Forward_gpu,
        conv_layer.cu:12-22).  Several bottoms with one activation (the siamese towers of FlowNetC: one Convolution, two bottoms, two
        tops) run as ONE launch on the bottoms stacked along the batch axis -- the kernels are per-sample -- and the tops are the
        halves of its output."""
        self._stacked = None
        if len(bottom) > 1 and len({self._slope(i) for i in range(len(top))}) == 1 and bottom[0].data.is_cuda:
            n = bottom[0].num()
            x = torch.cat([b.data for b in bottom], 0)
            y = one(x, self._slope(0))
            for i, t in enumerate(top):
                t.data = y[i * n:(i + 1) * n]
            self._stacked = (x, y)              # Backward_gpu runs the stacked batch as ONE launch per gradient, like the forward
            return
        for i, (b, t) in enumerate(zip(bottom, top)):
            t.data = one(b.data, self._slope(i))

    def _accumulate(self, k, g):
        """Parameter diffs accumulate (weight_gpu_gemm / backward_gpu_bias with beta = 1); Net.ClearParamDiffs zeroes them per iteration."""
        if g is not None:
            self.blobs_[k].mutable_gpu_diff().add_(g.reshape(self.blobs_[k].shape()))

    def _backward_one(self, x, y, g, slope, need_x, head):
        from . import functional as Fn, ops
        w = self.blobs_[0].data
        need_w = self.param_propagate_down(0)
        need_b = self.bias_term_ and self.param_propagate_down(1)
        if head == "predict" and slope is None and ops.predict_flow_conv_backward_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3]):
            blob, c0 = Fn._channel_slice(x)
            return ops.predict_flow_conv_backward((blob, c0, x.shape[1]), w, g.contiguous(), need_x, need_w, need_b)
        if head == "upsample" and slope is None:
            return ops.upsample_flow_deconv_backward(x.contiguous(), w, g.contiguous(), need_x, need_w, need_b)
        return Fn.conv_backward(x, w, y if slope is not None else None, g, self.stride_, self.pad_, slope if slope is not None else 0.0,
                                self.transposed, need_x, need_w, need_b)

    def _head_kind(self):
        return None

    def Backward_gpu(self, top, propagate_down, bottom):
        CHECK(self.backend_ is None, "Backward runs on the default backend (flownet2_amd.functional) only")
        head = self._head_kind()
        if getattr(self, "_stacked", None) is not None:
            x, y = self._stacked
            n = bottom[0].num()
            g = torch.cat([t.mutable_gpu_diff() for t in top], 0)
            gx, gw, db = self._backward_one(x, y, g, self._slope(0), any(propagate_down), head)
            self._accumulate(0, gw)
            if self.bias_term_:
                self._accumulate(1, db)
            for i, b in enumerate(bottom):
                if propagate_down[i]:
                    b.diff = gx[i * n:(i + 1) * n]
            return
        for i, (b, t) in enumerate(zip(bottom, top)):
            gx, gw, db = self._backward_one(b.data, t.data, t.mutable_gpu_diff(), self._slope(i), bool(propagate_down[i]), head)
            self._accumulate(0, gw)
            if self.bias_term_:
                self._accumulate(1, db)
            if propagate_down[i]:
                b.diff = gx

    def _backend(self):
        if self.backend_ is not None:
            return self.backend_
        from . import functional
        return functional


class ConvolutionLayer(_ConvBase):
    def type(self): return "Convolution"

    def Forward_gpu(self, bottom, top):
        from . import nets
        be = self._backend()
        w, b = self.blobs_[0].data, (self.blobs_[1].data if self.bias_term_ else None)
        k, s, p = self.kernel_, self.stride_, self.pad_

        def one(x, slope):
            if k == 1 and s == 1 and p == 0 and slope is None and getattr(self, "diagonal_", None) is not None:
                # the deploy tail's SCALE convolution (weight_filler diagonal, run-flownet.py:47-48): y[c] = diag[c] x[c] + 0 x[other]
                y = x * self.diagonal_.view(1, -1, 1, 1)
                return y + b.view(1, -1, 1, 1) if (b is not None and bool((b != 0).any())) else y
            if k == 3 and s == 1 and p == 1 and self.num_output_ == 2 and slope is None and hasattr(be, "predict_flow_conv"):
                return be.predict_flow_conv(x, w, b)                                      # the predict_flow heads (csrc/flow_head.hip)
            return nets.conv_forward(x, w, b if b is not None else torch.zeros(self.num_output_, device=x.device), s, p,
                                     slope is not None, be, slope=slope)
        self._forward_all(bottom, top, one)

    def _head_kind(self):
        return "predict" if (self.kernel_, self.stride_, self.pad_, self.num_output_) == (3, 1, 1, 2) else None

    def note_weights_changed(self):
        """After the weights were (re)loaded: a 1x1 weight that is diagonal is applied as a per-channel scale (exactly what the zero
        off-diagonal products add up to for finite inputs)."""
        self.diagonal_ = None
        w = self.blobs_[0].data
        if self.kernel_ == 1 and w.shape[0] == w.shape[1]:
            m = w.view(w.shape[0], w.shape[1])
            d = torch.diagonal(m)
            if bool((m - torch.diag(d) == 0).all()):
                self.diagonal_ = d.contiguous()


class DeconvolutionLayer(_ConvBase):
    transposed = True

    def type(self): return "Deconvolution"

    def Forward_gpu(self, bottom, top):
        from . import nets
        be = self._backend()
        w, b = self.blobs_[0].data, (self.blobs_[1].data if self.bias_term_ else None)
        k, s, p = self.kernel_, self.stride_, self.pad_

        def one(x, slope):
            if (k, s, p) == (4, 2, 1) and w.shape[0] == 2 and w.shape[1] == 2 and slope is None and hasattr(be, "upsample_flow_deconv"):
                return be.upsample_flow_deconv(x, w, b)                                   # the upsample_flow heads (csrc/flow_head.hip)
            if (k, s, p) == (4, 2, 1):
                bb = b if b is not None else torch.zeros(self.num_output_, device=x.device)
                return nets.deconv_forward(x, w, bb, slope is not None, be, slope=slope)
            y = be.lib_conv_transpose2d(x, w, b, s, p) if hasattr(be, "lib_conv_transpose2d") else torch.nn.functional.conv_transpose2d(x, w, b, stride=s, padding=p)
            return torch.nn.functional.leaky_relu(y, slope) if slope is not None else y       # (a geometry outside FlowNet's: COUNTED in LIBRARY_FALLBACKS)
        self._forward_all(bottom, top, one)

    def _head_kind(self):
        w = self.blobs_[0].data
        return "upsample" if (self.kernel_, self.stride_, self.pad_) == (4, 2, 1) and w.shape[0] == 2 and w.shape[1] == 2 else None

    def note_weights_changed(self):
        pass


class ReLULayer(Layer):
    def type(self): return "ReLU"
    def ExactNumBottomBlobs(self): return 1
    def ExactNumTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        self.negative_slope_ = float(self.layer_param_.relu_param.get("negative_slope", 0.0))
        self.folded_ = False                    # True: the producing Convolution / Deconvolution applies it (executor peephole)

    def Reshape(self, bottom, top):
        top[0].ReshapeLike(bottom[0])

    def Forward_gpu(self, bottom, top):
        if self.folded_:
            if top[0] is not bottom[0]:
                top[0].data = bottom[0].data
            return
        x = bottom[0].data
        be = self.backend_
        if x.is_cuda and x.is_contiguous() and (be is None or hasattr(be, "conv_bias_leaky_relu")):
            from . import ops
            y = x if top[0] is bottom[0] else x.clone()
            top[0].data = ops.bias_leaky_relu_(y, None, self.negative_slope_)             # csrc/bias_act.hip, in place
        else:
            top[0].data = torch.nn.functional.leaky_relu(x, self.negative_slope_)

    def Backward_gpu(self, top, propagate_down, bottom):                                  # relu_layer.cu:33-60
        if not propagate_down[0]:
            return
        g = top[0].mutable_gpu_diff()
        if self.folded_:                        # the producing Convolution undoes the activation in its own backward pass
            bottom[0].diff = g
            return
        # in place the bottom data IS the top data; for a positive slope its sign is the input's (the reference reads bottom_data)
        y = top[0].data
        if y.is_cuda and y.is_contiguous():
            from . import ops
            bottom[0].diff = ops.bias_leaky_relu_backward(y, g.contiguous(), self.negative_slope_, False)[0]
        else:
            bottom[0].diff = g * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, self.negative_slope_))


class EltwiseLayer(Layer):
    def type(self): return "Eltwise"
    def MinBottomBlobs(self): return 1                                                   # eltwise_layer.hpp:29 (this fork)
    def ExactNumTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        ep = self.layer_param_.eltwise_param
        coeff = list(ep.get("coeff", []))
        CHECK(len(coeff) == 0 or len(coeff) == len(bottom), "Eltwise Layer takes one coefficient per bottom blob.")   # cpp:12-14
        self.op_ = str(ep.get("operation", "SUM"))
        CHECK(not (self.op_ == "PROD" and coeff), "Eltwise layer only takes coefficients for summation.")             # cpp:15-17
        self.coeffs_ = [float(c) for c in coeff] if coeff else [1.0] * len(bottom)
        self.stable_prod_grad_ = bool(ep.get("stable_prod_grad", True))                                               # caffe.proto: default true

    def Reshape(self, bottom, top):
        for b in bottom[1:]:
            CHECK(b.shape() == bottom[0].shape(), "bottom shapes differ")                 # cpp:29-31
        top[0].ReshapeLike(bottom[0])

    def Forward_gpu(self, bottom, top):
        xs = [b.data for b in bottom]
        if self.op_ == "PROD":
            y = xs[0] * xs[1] if len(xs) > 1 else xs[0].clone()
            for x in xs[2:]:
                y = y * x
        elif self.op_ == "MAX":
            y = xs[0].clone()
            for x in xs[1:]:
                y = torch.maximum(y, x)
        else:
            # eltwise_layer.cu:46-52: top = 0; top = coeff_i * bottom_i + top for every i.  0 + c x == c x, 1 a + (-1) b == a - b exactly
            c = self.coeffs_
            if len(xs) == 1:
                y = xs[0] * c[0]
            elif len(xs) == 2 and c == [1.0, -1.0]:
                y = xs[0] - xs[1]
            elif len(xs) == 2 and c == [1.0, 1.0]:
                y = xs[0] + xs[1]
            else:
                y = xs[0] * c[0]
                for x, ci in zip(xs[1:], c[1:]):
                    y = torch.add(y, x, alpha=ci)
        top[0].data = y

    def Backward_gpu(self, top, propagate_down, bottom):                                  # eltwise_layer.cu:68-118
        g = top[0].mutable_gpu_diff()
        for i, b in enumerate(bottom):
            if not propagate_down[i]:
                continue
            if self.op_ == "PROD" and not self.stable_prod_grad_:
                b.diff = (top[0].data / b.data) * g                    # eltwise_layer.cu:109-111: top / bottom (0 / 0 = NaN there too), then x top_diff
            elif self.op_ == "PROD":
                d = None                                               # :97-108: the product of the OTHER bottoms, in order, then x top_diff
                for j, o in enumerate(bottom):
                    if j != i:
                        d = o.data.clone() if d is None else o.data * d
                b.diff = g.clone() if d is None else d * g
            elif self.op_ == "MAX":
                # MaxForward (eltwise_layer.cu:10-31) keeps the running top only where it is STRICTLY greater (`a > b`, else b): on a tie the
                # LAST bottom that holds the maximum is recorded in the mask, and MaxBackward routes the gradient by that mask
                last = torch.ones_like(g, dtype=torch.bool)
                for j in range(i + 1, len(bottom)):
                    last &= bottom[j].data < top[0].data
                b.diff = g * ((b.data == top[0].data) & last).to(g.dtype)
            else:
                b.diff = g.clone() if self.coeffs_[i] == 1.0 else g * self.coeffs_[i]


class ConcatLayer(Layer):
    def type(self): return "Concat"
    def MinBottomBlobs(self): return 1
    def ExactNumTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        cp = self.layer_param_.concat_param
        CHECK(not ("axis" in cp and "concat_dim" in cp), "Either axis or concat_dim should be specified; not both.")   # cpp:13-14
        self.axis_ = int(cp.get("concat_dim", cp.get("axis", 1)))

    def Reshape(self, bottom, top):
        shape = bottom[0].shape()
        for b in bottom[1:]:
            s = b.shape()
            CHECK(len(s) == len(shape) and all(s[i] == shape[i] for i in range(len(s)) if i != self.axis_),
                  "All inputs must have the same shape, except at concat_axis.")          # cpp:42-49
            shape[self.axis_] += s[self.axis_]
        top[0].Reshape(*shape)

    def Forward_gpu(self, bottom, top):
        # one bottom: the top SHARES the bottom's storage, like the reference (concat_layer.cpp:50-53 ShareData / ShareDiff)
        top[0].data = bottom[0].data if len(bottom) == 1 else torch.cat([b.data for b in bottom], self.axis_)

    def Backward_gpu(self, top, propagate_down, bottom):                                  # concat_layer.cu:47-74
        g = top[0].mutable_gpu_diff()
        if len(bottom) == 1:
            bottom[0].diff = g
            return
        off = 0
        for i, b in enumerate(bottom):
            n = b.shape(self.axis_)
            if propagate_down[i]:
                b.diff = g.narrow(self.axis_, off, n)       # a channel-slice VIEW of the Concat's top_diff: the consumers read it in place
            off += n


class SliceLayer(Layer):
    def type(self): return "Slice"
    def ExactNumBottomBlobs(self): return 1
    def MinTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        sp = self.layer_param_.slice_param
        CHECK(not ("axis" in sp and "slice_dim" in sp), "Either axis or slice_dim should be specified; not both.")     # cpp:13-14
        self.axis_ = int(sp.get("slice_dim", sp.get("axis", 1)))
        self.slice_point_ = [int(v) for v in sp.get("slice_point", [])]

    def Reshape(self, bottom, top):
        n = bottom[0].shape(self.axis_)
        if self.slice_point_:
            CHECK(len(self.slice_point_) == len(top) - 1, "slice_point count must be top count - 1")                  # cpp:50
            cuts = [0] + self.slice_point_ + [n]
            CHECK(all(cuts[i] < cuts[i + 1] for i in range(len(cuts) - 1)), "slice points must be increasing")         # cpp:56
        else:
            CHECK(n % len(top) == 0, "Number of top blobs must evenly divide the slice axis")                          # cpp:67-70
            step = n // len(top)
            cuts = [i * step for i in range(len(top) + 1)]
        self.cuts_ = cuts
        for i, t in enumerate(top):
            s = bottom[0].shape()
            s[self.axis_] = cuts[i + 1] - cuts[i]
            t.Reshape(*s)

    def Forward_gpu(self, bottom, top):
        x = bottom[0].data
        if len(top) == 1:                                   # slice_layer.cpp:69-72: one top shares the bottom's storage
            top[0].data = x
            return
        for i, t in enumerate(top):
            # every top owns its storage (slice_layer.cpp:76-95 copies): a narrow() that is already contiguous (batch 1, or a slice along
            # axis 0) is a VIEW of the bottom, and an in-place layer behind it would write through into the bottom and the other tops
            y = x.narrow(self.axis_, self.cuts_[i], self.cuts_[i + 1] - self.cuts_[i])
            t.data = y.clone() if y.is_contiguous() else y.contiguous()

    def Backward_gpu(self, top, propagate_down, bottom):                                  # slice_layer.cu:49-75
        if not propagate_down[0]:
            return
        bottom[0].diff = top[0].mutable_gpu_diff() if len(top) == 1 else torch.cat([t.mutable_gpu_diff() for t in top], self.axis_)


class SilenceLayer(Layer):
    def type(self): return "Silence"
    def MinBottomBlobs(self): return 1
    def ExactNumTopBlobs(self): return 0
    def Reshape(self, bottom, top): pass
    def Forward_gpu(self, bottom, top): pass

    def Backward_gpu(self, top, propagate_down, bottom):                                  # silence_layer.cu:16-27
        for i, b in enumerate(bottom):
            if propagate_down[i]:
                b.diff = torch.zeros_like(b.data)


class SplitLayer(Layer):
    """split_layer.cpp:8-27, split_layer.cu:8-33: every top SHARES the bottom's data; the bottom's diff is the sum of the tops' diffs (the first
    two added, the rest accumulated in order).  Inserted by Net::Init (InsertSplits) wherever a blob has more than one consumer."""

    def type(self): return "Split"
    def ExactNumBottomBlobs(self): return 1
    def MinTopBlobs(self): return 1

    def Reshape(self, bottom, top):
        for t in top:
            CHECK(t is not bottom[0], "Split Layer does not allow in-place computation.")   # split_layer.cpp:13-18
            t.ReshapeLike(bottom[0])

    def Forward_gpu(self, bottom, top):
        for t in top:
            t.data = bottom[0].data

    def Backward_gpu(self, top, propagate_down, bottom):
        if not propagate_down[0]:
            return
        if len(top) == 1:
            bottom[0].diff = top[0].mutable_gpu_diff().clone()
            return
        d = top[0].mutable_gpu_diff() + top[1].mutable_gpu_diff()
        for t in top[2:]:
            d = d + t.mutable_gpu_diff()
        bottom[0].diff = d


class InputLayer(Layer):
    """input_layer.cpp:8-27: tops shaped by input_param.shape (one shape for all tops, or one per top)."""

    def type(self): return "Input"
    def ExactNumBottomBlobs(self): return 0
    def MinTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        shapes = self.layer_param_.input_param.get("shape", [])
        if isinstance(shapes, dict):
            shapes = [shapes]
        CHECK(len(shapes) in (0, 1, len(top)), "Must specify 'shape' once, once per top blob, or not at all")           # cpp:14-17
        for i, t in enumerate(top):
            if shapes:
                t.Reshape(*[int(d) for d in shapes[i if len(shapes) > 1 else 0].get("dim", [])])

    def Reshape(self, bottom, top): pass
    def Forward_gpu(self, bottom, top): pass


for _name, _cls in (("Convolution", ConvolutionLayer), ("Deconvolution", DeconvolutionLayer), ("ReLU", ReLULayer), ("Eltwise", EltwiseLayer),
                    ("Concat", ConcatLayer), ("Slice", SliceLayer), ("Silence", SilenceLayer), ("Split", SplitLayer), ("Input", InputLayer)):
    L.REGISTER_LAYER_CLASS(_name, _cls)
