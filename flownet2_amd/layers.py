"""Host-side mirror of the reference's operator/plugin interface for the hot-path layers.

Same names, argument meaning and error behaviour as the Caffe `Layer<Dtype>` API
(include/caffe/layer.hpp:42-53,69-76,94-152,484-537) and the layer registry
(include/caffe/layer_factory.hpp:67-84): a layer is created from a LayerParameter by its prototxt
`type:` string, `SetUp` = CheckBlobCounts -> LayerSetUp -> Reshape -> SetLossWeights, `Forward`
re-runs Reshape unless `reshape_every_iter` is false, `Backward(top, propagate_down, bottom)` writes
`bottom[i].diff`.  Blobs are NCHW fp32 with separate data / diff (include/caffe/blob.hpp:153-164,
269-270); here both live in HBM as torch tensors and the layers hand their device pointers to the C
ABI in include/flownet2_hip.h.  CHECK / LOG(FATAL) of the reference become exceptions.

The C++ twin of this file for a real Caffe tree is flownet2_amd/csrc/caffe_adapter/ (INTEGRATION.md).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import ops
from ._lib import Fn2Error


class CheckError(ValueError):
    """A glog CHECK / LOG(FATAL) of the reference."""


def CHECK(cond, msg):
    if not cond:
        raise CheckError(msg)


class Blob:
    """include/caffe/blob.hpp: NCHW tensor with data and diff, device-resident."""

    def __init__(self, *shape, device: Optional[torch.device] = None):
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.data: Optional[torch.Tensor] = None
        self.diff: Optional[torch.Tensor] = None
        self._shape: List[int] = []
        if shape:
            self.Reshape(*shape)

    @classmethod
    def from_tensor(cls, t: torch.Tensor) -> "Blob":
        b = cls(device=t.device)
        b._shape = list(t.shape)
        b.data = t.contiguous().float()
        return b

    def Reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            shape = tuple(shape[0])
        shape = [int(s) for s in shape]
        CHECK(all(s >= 0 for s in shape), "blob dims must be >= 0")
        if shape != self._shape or self.data is None:
            self._shape = shape
            self.data = torch.zeros(shape, dtype=torch.float32, device=self.device)
            self.diff = None

    def ReshapeLike(self, other: "Blob"):
        self.Reshape(*other.shape())

    def shape(self, i: Optional[int] = None):
        return list(self._shape) if i is None else self._shape[i]

    def num_axes(self):
        return len(self._shape)

    def shape_string(self) -> str:                        # blob.hpp:56-63
        return "".join("%d " % d for d in self._shape) + "(%d)" % self.count()

    def count(self):
        n = 1
        for s in self._shape:
            n *= s
        return n

    def _legacy(self, i):
        CHECK(len(self._shape) <= 4, "Cannot use legacy accessors on Blobs with > 4 axes.")
        return self._shape[i] if i < len(self._shape) else 1

    def num(self): return self._legacy(0)
    def channels(self): return self._legacy(1)
    def height(self): return self._legacy(2)
    def width(self): return self._legacy(3)

    # gpu_data()/mutable_gpu_data()/gpu_diff()/mutable_gpu_diff() of the reference
    def gpu_data(self) -> torch.Tensor:
        return self.data

    def mutable_gpu_diff(self) -> torch.Tensor:
        if self.diff is None or list(self.diff.shape) != self._shape:
            self.diff = torch.zeros(self._shape, dtype=torch.float32, device=self.device)
        return self.diff

    gpu_diff = mutable_gpu_diff

    def cpu_data(self):
        return self.data.detach().cpu().numpy()

    def cpu_diff(self):
        return self.mutable_gpu_diff().detach().cpu().numpy()


@dataclass
class LayerParameter:
    """The subset of caffe.proto's LayerParameter (:312-425) these layers read.  The *_param dicts
    use the proto field names (correlation_param: caffe.proto:628-644, flow_warp_param :553-560,
    resample_param :665-677, l1_loss_param :619-625, downsample_param :646-649, data_param :918-986)."""
    name: str = ""
    type: str = ""
    bottom: List[str] = field(default_factory=list)
    top: List[str] = field(default_factory=list)
    loss_weight: List[float] = field(default_factory=list)
    reshape_every_iter: bool = True          # caffe.proto:424
    correlation_param: Dict = field(default_factory=dict)
    flow_warp_param: Dict = field(default_factory=dict)
    resample_param: Dict = field(default_factory=dict)
    l1_loss_param: Dict = field(default_factory=dict)
    downsample_param: Dict = field(default_factory=dict)
    data_param: Dict = field(default_factory=dict)      # DataParameter, caffe.proto:918-986 (CustomData)
    augmentation_param: Dict = field(default_factory=dict)   # AugmentationParameter, caffe.proto:489-546 (crop sizes, mean, generators as dicts)
    coeff_schedule_param: Dict = field(default_factory=dict)  # CoeffScheduleParameter, caffe.proto:693-697
    phase: str = "TEST"                                       # TRAIN / TEST (caffe.proto:311); the augmentation layers draw only in TRAIN
    # the stock layers of a FlowNet prototxt (flownet2_amd/stock_layers.py)
    convolution_param: Dict = field(default_factory=dict)     # ConvolutionParameter, caffe.proto:568-617
    relu_param: Dict = field(default_factory=dict)            # ReLUParameter
    eltwise_param: Dict = field(default_factory=dict)         # EltwiseParameter
    concat_param: Dict = field(default_factory=dict)          # ConcatParameter
    slice_param: Dict = field(default_factory=dict)           # SliceParameter
    input_param: Dict = field(default_factory=dict)           # InputParameter
    param: List[Dict] = field(default_factory=list)           # ParamSpec (name: parameter sharing, net.cpp:451-540)
    propagate_down: List[bool] = field(default_factory=list)
    include: List[Dict] = field(default_factory=list)         # NetStateRule (phase only)
    exclude: List[Dict] = field(default_factory=list)

    @classmethod
    def from_dict(cls, d: Dict, phase: str = "TEST") -> "LayerParameter":
        """From flownet2_amd.prototxt.to_dict(layer message): the fields above by name; unknown fields are kept out (a layer type that
        needs them is not registered either)."""
        import dataclasses
        known = {f.name for f in dataclasses.fields(cls)}
        kw = {k: v for k, v in d.items() if k in known}
        kw.setdefault("phase", phase)
        for k in ("bottom", "top", "loss_weight", "param", "propagate_down", "include", "exclude"):
            if k in kw and not isinstance(kw[k], list):
                kw[k] = [kw[k]]
        kw["name"], kw["type"] = str(kw.get("name", "")), str(kw.get("type", ""))
        return cls(**kw)


class Layer:
    """include/caffe/layer.hpp.  Only the GPU mode exists (Forward == Forward_gpu)."""

    def __init__(self, param: LayerParameter):
        self.layer_param_ = param
        self.loss_: List[float] = []
        self.blobs_: List[Blob] = []          # learnable parameters (layer.hpp:288-291)
        self.param_propagate_down_: List[bool] = []       # layer.hpp:327-341, set by Net::Init (net.cpp:173-185)
        self.backend_ = None                  # flownet2_amd.functional unless the Net was given another one (stock layers only)

    def blobs(self):
        return self.blobs_

    def param_propagate_down(self, param_id: int) -> bool:                 # layer.hpp:327-331
        return self.param_propagate_down_[param_id] if param_id < len(self.param_propagate_down_) else False

    def set_param_propagate_down(self, param_id: int, value: bool):        # layer.hpp:336-341
        if len(self.param_propagate_down_) <= param_id:
            self.param_propagate_down_ += [True] * (param_id + 1 - len(self.param_propagate_down_))
        self.param_propagate_down_[param_id] = bool(value)

    def AllowForceBackward(self, bottom_index: int) -> bool: return True    # layer.hpp:314-316

    # --- interface subclasses implement -------------------------------------------------------
    def LayerSetUp(self, bottom: Sequence[Blob], top: Sequence[Blob]): pass
    def Reshape(self, bottom: Sequence[Blob], top: Sequence[Blob]): raise NotImplementedError
    def Forward_gpu(self, bottom, top): raise NotImplementedError
    def Backward_gpu(self, top, propagate_down, bottom): raise NotImplementedError
    def type(self) -> str: return ""
    def ExactNumBottomBlobs(self): return -1
    def MinBottomBlobs(self): return -1
    def MaxBottomBlobs(self): return -1
    def ExactNumTopBlobs(self): return -1
    def MinTopBlobs(self): return -1
    def MaxTopBlobs(self): return -1
    def EqualNumBottomTopBlobs(self) -> bool: return False   # layer.hpp:293
    def AutoTopBlobs(self) -> bool: return False              # layer.hpp:303 (loss layers: loss_layer.hpp:40)
    def AllowBackward(self) -> bool: return True      # layer.hpp:322-324

    # --- layer.hpp:69-76 -----------------------------------------------------------------------
    def SetUp(self, bottom, top):
        self.CheckBlobCounts(bottom, top)
        self.LayerSetUp(bottom, top)
        self.Reshape(bottom, top)
        self.SetLossWeights(top)

    def CheckBlobCounts(self, bottom, top):               # layer.hpp:397-435
        t = self.type()
        if self.ExactNumBottomBlobs() >= 0:
            CHECK(self.ExactNumBottomBlobs() == len(bottom), f"{t} Layer takes {self.ExactNumBottomBlobs()} bottom blob(s) as input.")
        if self.MinBottomBlobs() >= 0:
            CHECK(self.MinBottomBlobs() <= len(bottom), f"{t} Layer takes at least {self.MinBottomBlobs()} bottom blob(s) as input.")
        if self.MaxBottomBlobs() >= 0:
            CHECK(self.MaxBottomBlobs() >= len(bottom), f"{t} Layer takes at most {self.MaxBottomBlobs()} bottom blob(s) as input.")
        if self.ExactNumTopBlobs() >= 0:
            CHECK(self.ExactNumTopBlobs() == len(top), f"{t} Layer produces {self.ExactNumTopBlobs()} top blob(s) as output.")
        if self.MinTopBlobs() >= 0:
            CHECK(self.MinTopBlobs() <= len(top), f"{t} Layer produces at least {self.MinTopBlobs()} top blob(s) as output.")
        if self.MaxTopBlobs() >= 0:
            CHECK(self.MaxTopBlobs() >= len(top), f"{t} Layer produces at most {self.MaxTopBlobs()} top blob(s) as output.")
        if self.EqualNumBottomTopBlobs():
            CHECK(len(bottom) == len(top), f"{t} Layer produces one top blob as output for each bottom blob input.")   # layer.hpp:433-437

    def SetLossWeights(self, top):                         # layer.hpp:444-458
        lw = list(self.layer_param_.loss_weight)
        self.loss_ = [0.0] * len(top)
        if lw:
            CHECK(len(top) == len(lw), "loss_weight must be unspecified or specified once per top blob.")
            for i, w in enumerate(lw):
                if w == 0:
                    continue
                self.loss_[i] = float(w)
                top[i].mutable_gpu_diff().fill_(float(w))

    def loss(self, top_id):
        return self.loss_[top_id] if top_id < len(self.loss_) else 0.0

    def Forward(self, bottom, top):                        # layer.hpp:484-521
        if self.layer_param_.reshape_every_iter:
            self.Reshape(bottom, top)
        self.Forward_gpu(bottom, top)
        loss = 0.0
        for i, t in enumerate(top):
            if not self.loss(i):
                continue
            loss = loss + (t.data * t.mutable_gpu_diff()).sum()   # caffe_gpu_dot(data, loss_weights)
        return loss

    def Backward(self, top, propagate_down, bottom):       # layer.hpp:524-537
        self.Backward_gpu(top, propagate_down, bottom)


def _wrap(fn, *a, **k):
    try:
        return fn(*a, **k)
    except Fn2Error as e:      # reference: CHECK / LOG(FATAL)
        raise CheckError(str(e)) from e


class CorrelationLayer(Layer):
    """include/caffe/layers/correlation_layer.hpp:26-77; correlation_layer.cpp:13-84."""

    def type(self): return "Correlation"
    def ExactNumBottomBlobs(self): return 2
    def ExactNumTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        cp = self.layer_param_.correlation_param
        CHECK("kernel_size" in cp, "Filter kernel_size is not set")                 # cpp:18
        CHECK("max_displacement" in cp, "Max displacement is required.")            # cpp:19
        ctype = cp.get("correlation_type", 0)
        if isinstance(ctype, str):
            ctype = {"MULTIPLY": 0, "SUBTRACT": 1}[ctype]
        if int(cp["kernel_size"]) % 2 == 0:
            raise CheckError("Odd kernel size required")                             # cpp:22
        self.params_ = ops.corr_params(cp.get("pad", 0), cp["kernel_size"], cp["max_displacement"],
                                       cp.get("stride_1", 1), cp.get("stride_2", 1), ctype, cp.get("do_abs", False))

    def Reshape(self, bottom, top):
        CHECK(bottom[0].width() == bottom[1].width(), "Both bottom blobs must have same width")       # cpp:45
        CHECK(bottom[0].height() == bottom[1].height(), "Both bottom blobs must have same height")    # cpp:46
        CHECK(bottom[0].channels() == bottom[1].channels(), "Both bottom blobs must have same height")  # cpp:47 (sic)
        self.num_ = bottom[0].num()
        tc, th, tw = _wrap(ops.correlation_out_shape, self.params_, bottom[0].channels(), bottom[0].height(), bottom[0].width())
        self.top_channels_, self.top_height_, self.top_width_ = tc, th, tw
        top[0].Reshape(self.num_, tc, th, tw)                                        # cpp:73

    def Forward_gpu(self, bottom, top):
        _wrap(ops.correlation_forward, self.params_, bottom[0].data, bottom[1].data, out=top[0].data)

    def Backward_gpu(self, top, propagate_down, bottom):
        # like the reference (correlation_layer.cu:508-603) both diffs are always written
        d0, d1 = _wrap(ops.correlation_backward, self.params_, bottom[0].data, bottom[1].data, top[0].mutable_gpu_diff())
        bottom[0].diff, bottom[1].diff = d0, d1


class Correlation1DLayer(Layer):
    """include/caffe/layers/correlation_1d_layer.hpp; correlation_layer1d.cpp:12-92 (horizontal displacements only)."""

    def type(self): return "Correlation1D"
    def ExactNumBottomBlobs(self): return 2
    def ExactNumTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        cp = self.layer_param_.correlation_param
        CHECK("kernel_size" in cp, "Filter kernel_size is not set")                 # cpp:19
        CHECK("max_displacement" in cp, "Max displacement is required.")            # cpp:20
        ctype = cp.get("correlation_type", 0)
        if isinstance(ctype, str):
            ctype = {"MULTIPLY": 0, "SUBTRACT": 1}[ctype]
        if int(cp["kernel_size"]) % 2 == 0:
            raise CheckError("Odd kernel size required")                             # cpp:23
        sd = int(cp.get("single_direction", 0))
        if sd < -1 or sd > 1:
            raise CheckError("single_direction must be -1 (left), 0 (off), or 1 (right)")   # cpp:30
        self.params_ = ops.corr_params(cp.get("pad", 0), cp["kernel_size"], cp["max_displacement"], cp.get("stride_1", 1),
                                       cp.get("stride_2", 1), ctype, cp.get("do_abs", False), sd)

    def Reshape(self, bottom, top):
        CHECK(bottom[0].width() == bottom[1].width(), "Both bottom blobs must have same width")       # cpp:48
        CHECK(bottom[0].height() == bottom[1].height(), "Both bottom blobs must have same height")    # cpp:49
        CHECK(bottom[0].channels() == bottom[1].channels(), "Both bottom blobs must have same number of channels")  # cpp:50
        tc, th, tw = _wrap(ops.correlation1d_out_shape, self.params_, bottom[0].channels(), bottom[0].height(), bottom[0].width())
        top[0].Reshape(bottom[0].num(), tc, th, tw)                                  # cpp:81

    def Forward_gpu(self, bottom, top):
        top[0].data = _wrap(ops.correlation1d_forward, self.params_, bottom[0].data, bottom[1].data)

    def Backward_gpu(self, top, propagate_down, bottom):
        # like the reference (correlation_layer1d.cu:513-616) both diffs are always written
        d0, d1 = _wrap(ops.correlation1d_backward, self.params_, bottom[0].data, bottom[1].data, top[0].mutable_gpu_diff())
        bottom[0].diff, bottom[1].diff = d0, d1


class FlowWarpLayer(Layer):
    """include/caffe/layers/flow_warp_layer.hpp; flow_warp_layer.cpp:35-52."""

    def type(self): return "FlowWarp"

    def Reshape(self, bottom, top):
        CHECK(len(bottom) == 2, "FlowWarpLayer takes two input blobs: image and flow.")
        CHECK(len(top) == 1, "FlowWarpLayer outputs one blob.")
        CHECK(bottom[0].num() == bottom[1].num(), "Num of the inputs should be the same")
        CHECK(bottom[1].channels() == 2, "Flow should have 2 channels: x-flow and y-flow")
        CHECK(bottom[0].width() == bottom[1].width(), "Width of the inputs should be the same")
        CHECK(bottom[0].height() == bottom[1].height(), "Height of the inputs should be the same")
        top[0].Reshape(*bottom[0].shape())

    def _fill(self):
        fv = self.layer_param_.flow_warp_param.get("fill_value", "ZERO")
        if isinstance(fv, str):
            fv = {"ZERO": ops.FILL_ZERO, "NOT_A_NUMBER": ops.FILL_NAN}[fv]
        return int(fv)

    def Forward_gpu(self, bottom, top):
        top[0].data = _wrap(ops.flow_warp_forward, bottom[0].data, bottom[1].data, self._fill())

    def Backward_gpu(self, top, propagate_down, bottom):
        di, df = _wrap(ops.flow_warp_backward, bottom[0].data, bottom[1].data, top[0].mutable_gpu_diff(),
                       bool(propagate_down[0]), bool(propagate_down[1]))
        bottom[0].diff, bottom[1].diff = di, df


class ResampleLayer(Layer):
    """src/caffe/layers/resample_layer.hpp; resample_layer.cpp:14-55.  Forward only."""

    _TYPES = {"NEAREST": ops.NEAREST, "LINEAR": ops.LINEAR, "CUBIC": ops.CUBIC, "AREA": ops.AREA}

    def type(self): return "Resample"
    def AllowBackward(self): return False                                          # resample_layer.hpp:25

    def _rtype(self):
        t = self.layer_param_.resample_param.get("type", "LINEAR")
        return self._TYPES[t] if isinstance(t, str) else int(t)

    def LayerSetUp(self, bottom, top):
        if self._rtype() not in (ops.CUBIC, ops.LINEAR, ops.NEAREST):
            raise CheckError("ResampleLayer: only CUBIC, LINEAR and NEAREST interpolation is supported for now")   # cpp:17-20

    def Reshape(self, bottom, top):
        self.layer_param_.reshape_every_iter = False                                # cpp:27: Reshape only runs on setup
        CHECK(1 <= len(bottom) <= 2, "ResampleLayer takes one or two bottoms")
        CHECK(len(top) == 1, "ResampleLayer outputs one blob")
        rp = self.layer_param_.resample_param
        if len(bottom) == 1:
            th, tw = int(rp.get("height", 0)), int(rp.get("width", 0))             # cpp:41-42
        else:
            th, tw = bottom[1].height(), bottom[1].width()                          # cpp:44-45
        CHECK(th >= 1, "ResampleLayer must have top_height > 0")
        CHECK(tw >= 1, "ResampleLayer must have top_width > 0")
        self.top_height_, self.top_width_ = th, tw
        top[0].Reshape(bottom[0].num(), bottom[0].channels(), th, tw)

    def Forward_gpu(self, bottom, top):
        rp = self.layer_param_.resample_param
        top[0].data = _wrap(ops.resample_forward, bottom[0].data, self.top_height_, self.top_width_, self._rtype(),
                            bool(rp.get("antialias", True)))

    def Backward_gpu(self, top, propagate_down, bottom):
        if any(propagate_down):
            raise CheckError("ResampleLayer cannot do backward.")                    # resample_layer.cu:209-213


class L1LossLayer(Layer):
    """include/caffe/layers/l1_loss_layer.hpp; l1loss_layer.cpp:11-90, l1loss_layer.cu:67-188."""

    def type(self): return "L1Loss"
    def MinBottomBlobs(self): return 1
    def MaxBottomBlobs(self): return 2
    def ExactNumTopBlobs(self): return 1
    def AutoTopBlobs(self): return True                   # LossLayer, loss_layer.hpp:40: a prototxt may leave the top out

    def LayerSetUp(self, bottom, top):
        if not self.layer_param_.loss_weight:        # LossLayer::LayerSetUp, loss_layer.cpp:8-13
            self.layer_param_.loss_weight = [1.0]
        CHECK(len(bottom) in (1, 2), "L1LossLayer needs one or two input blobs.")
        lp = self.layer_param_.l1_loss_param
        self.params_ = ops.l1_params(lp.get("l2_per_location", False), lp.get("l2_prescale_by_channels", False),
                                     lp.get("normalize_by_num_entries", False), lp.get("epsilon", 1e-2), lp.get("plateau", 0.0))
        self.ws_ = None

    def Reshape(self, bottom, top):
        top[0].Reshape()                              # 0-axis scalar, cpp:68-69

    def Forward_gpu(self, bottom, top):
        b1 = bottom[1].data if len(bottom) > 1 else None
        loss, self.ws_ = _wrap(ops.l1loss_forward, self.params_, bottom[0].data, b1, self.ws_)
        top[0].data = loss

    def normalize_coeff(self) -> float:
        return float(self.ws_[:8].view(torch.float32)[1])

    def Backward_gpu(self, top, propagate_down, bottom):
        prop = bool(propagate_down[0]) or (len(bottom) > 1 and bool(propagate_down[1]))       # cu:150-151
        if not prop:
            return
        top_diff = float(top[0].mutable_gpu_diff())       # top[0]->cpu_diff()[0], cu:155
        b1 = bottom[1].data if len(bottom) > 1 else None
        d0, d1 = _wrap(ops.l1loss_backward, self.params_, bottom[0].data, b1, top_diff, self.ws_)
        if len(bottom) > 1:
            # Eltwise backward honours propagate_down per bottom (eltwise_layer.cu Backward)
            if propagate_down[0]: bottom[0].diff = d0
            if propagate_down[1]: bottom[1].diff = d1
        else:
            bottom[0].diff = d0


class ChannelNormLayer(Layer):
    """include/caffe/layers/channel_norm_layer.hpp; channel_norm_layer.cpp:27-40."""

    def type(self): return "ChannelNorm"

    def Reshape(self, bottom, top):
        CHECK(len(bottom) == 1, "ChannelNormLayer takes one input blob.")
        CHECK(len(top) == 1, "ChannelNormLayer outputs one blob.")
        top[0].Reshape(bottom[0].num(), 1, bottom[0].height(), bottom[0].width())

    def Forward_gpu(self, bottom, top):
        top[0].data = _wrap(ops.channel_norm_forward, bottom[0].data)

    def Backward_gpu(self, top, propagate_down, bottom):
        bottom[0].diff = _wrap(ops.channel_norm_backward, bottom[0].data, top[0].data, top[0].mutable_gpu_diff())


class DownsampleLayer(Layer):
    """include/caffe/layers/downsample_layer.hpp; downsample_layer.cpp:21-57.  Forward only."""

    def type(self): return "Downsample"
    def AllowBackward(self): return False                                            # downsample_layer.hpp:30

    def Reshape(self, bottom, top):
        self.layer_param_.reshape_every_iter = False                                 # cpp:24
        CHECK(1 <= len(bottom) <= 2, "DownsampleLayer takes one or two bottoms")
        CHECK(len(top) == 1, "DownsampleLayer outputs one blob")
        dp = self.layer_param_.downsample_param
        if len(bottom) == 1:
            th, tw = int(dp.get("top_height", 0)), int(dp.get("top_width", 0))
        else:
            th, tw = bottom[1].height(), bottom[1].width()
        CHECK(th >= 1, "DownsampleLayer must have top_height > 0")
        CHECK(tw >= 1, "DownsampleLayer must have top_width > 0")
        self.top_height_, self.top_width_ = th, tw
        top[0].Reshape(bottom[0].num(), bottom[0].channels(), th, tw)

    def Forward_gpu(self, bottom, top):
        top[0].data = _wrap(ops.downsample_forward, bottom[0].data, self.top_height_, self.top_width_)

    def Backward_gpu(self, top, propagate_down, bottom):
        if any(propagate_down):
            raise CheckError("DownsamplingLayer cannot do backward.")                 # downsample_layer.cu:132-138


class DataAugmentationLayer(Layer):
    """include/caffe/layers/data_augmentation_layer.hpp; data_augmentation_layer.cpp:32-160, .cu:320-637 -- for GIVEN coefficients.
    bottom = [images] or [images, coefficient blob] (`input_params_`); top = [augmented] or [augmented, coefficient blob].
    Without bottom[1]: outside the training phase every sample gets the default coefficients (.cu:375-387: centre crop, no colour
    change); in the training phase (or with `augment_during_test`) the layer draws them from the generator sub-messages of
    `augmentation_param` (dicts: rand_type / exp / mean / spread / prob ...) through flownet2_amd/augment.py -- the reference's control
    flow with numpy's random stream (`augmentation_param.seed`, ours), since boost's cannot be reproduced.
    Mean: `augmentation_param.mean` (3 values) with `mean_per_pixel: false` (.cpp:142-151), or the blobs a trained model carries
    (`adjust_blobs`, cpp:162-205: only a layer with `recompute_mean > 0` takes them); `set_mean` is an explicit override of ours.  The
    running re-computation of the mean over the first `recompute_mean` iterations (.cu:593-621) keeps blobs_[0..2] as num_iter_ /
    mean_pixel_ / mean_channel_."""

    def type(self): return "DataAugmentation"
    def AllowBackward(self): return False                                            # hpp:29

    def LayerSetUp(self, bottom, top):
        import numpy as np
        self.layer_param_.reshape_every_iter = False                                  # cpp:38
        self.mean_ = None
        self.mean_mode_ = ops.MEAN_NONE
        self.num_iter_ = 0                                                            # blobs_[0], .cu:349-351
        # ours: key of the counter-based streams.  caffe.proto has no seed field; the reference draws from the process-wide cuRAND / boost
        # generators, so every layer instance gets its OWN noise and coefficients (data_augmentation_layer.cu:578-587): the layer name is
        # mixed into the key, else img0s_aug and img1s_aug would add the same per-pixel noise field to both frames of a pair
        import zlib
        self.seed_ = (int(self.layer_param_.augmentation_param.get("seed", 0)) ^ zlib.crc32(str(self.layer_param_.name).encode())) & 0x7fffffff
        self.mean_pixel_ = None                                                       # blobs_[1] / blobs_[2]: the running means of recompute_mean
        self.mean_channel_ = None

    def _generators(self):
        return {k: v for k, v in self.layer_param_.augmentation_param.items() if isinstance(v, dict)}

    def _draw(self, bottom):
        """.cu:364-450: only with a crop size, only in TRAIN (or augment_during_test), only the groups that have generators."""
        from . import augment
        ap = self.layer_param_.augmentation_param
        if not self.do_cropping_ or not (self.layer_param_.phase == "TRAIN" or ap.get("augment_during_test", False)) or not self._generators():
            return None
        disc = augment.discount_coeff(self.num_iter_, self.layer_param_.coeff_schedule_param)
        # counter-based stream: the draws of iteration i are a function of (seed, i) alone (prefetchable, restartable)
        return augment.draw_batch(augment.make_rng(self.seed_, self.num_iter_), self._generators(), bottom[0].num(), bottom[0].width(),
                                  bottom[0].height(), self.cropped_width_, self.cropped_height_, discount=disc)

    def adjust_blobs(self, blobs):
        """CustomCopyBlobs of Net::CopyTrainedLayersFrom (net.cpp:769-781) -> adjust_blobs, data_augmentation_layer.cpp:162-205.  `blobs`: the
        source layer's blobs [iteration count, per-pixel mean, per-channel mean].  Nothing is taken unless this layer re-computes its mean
        (`recompute_mean > 0`) and the source carries at least two blobs: a layer with `recompute_mean: 0` subtracts the `mean` of its own
        proto (or nothing), whatever the .caffemodel holds.  Otherwise the iteration count comes back (so that a count beyond
        recompute_mean freezes the mean) and, with mean_per_pixel: false, blobs[2]; with mean_per_pixel: true, blobs[1] and its
        per-channel average -- or, when the source mean has another size, the source's per-channel average expanded over the plane."""
        ap = self.layer_param_.augmentation_param
        if not (int(ap.get("recompute_mean", 0)) > 0 and len(blobs) >= 2):
            return False
        import numpy as np
        dev = getattr(self, "device_", None) or torch.device("cpu")
        b1 = np.ascontiguousarray(blobs[1], np.float32)
        b1 = b1.reshape((1,) * (4 - b1.ndim) + b1.shape) if b1.ndim < 4 else b1
        channels, ch, cw = self.channels_, self.cropped_height_, self.cropped_width_
        CHECK(channels == b1.shape[1], "data augmentation mean: channel count of the source blob differs")          # cpp:166
        self.num_iter_ = int(np.asarray(blobs[0], np.float32).reshape(-1)[0])                                      # cpp:172
        if self.mean_pixel_ is None:
            self.mean_pixel_ = torch.zeros((channels, ch, cw), dtype=torch.float32, device=dev)
        if not ap.get("mean_per_pixel", True):
            CHECK(len(blobs) >= 3, "data augmentation mean: the source layer has no per-channel mean blob")
            self.mean_channel_ = torch.from_numpy(np.ascontiguousarray(blobs[2], np.float32).reshape(-1)[:channels].copy()).to(dev)   # cpp:175-183
        elif (b1.shape[2], b1.shape[3]) == (ch, cw):
            self.mean_pixel_ = torch.from_numpy(b1[0].copy()).to(dev)                                              # cpp:186-189
            self.mean_channel_ = (self.mean_pixel_ * (1.0 / (ch * cw))).sum(dim=(1, 2))
        else:
            src = torch.from_numpy(b1[0].copy()).to(dev)                                                           # cpp:191-199
            self.mean_channel_ = (src * (1.0 / (b1.shape[2] * b1.shape[3]))).sum(dim=(1, 2))
            self.mean_pixel_ = self.mean_channel_.view(-1, 1, 1).expand(channels, ch, cw).contiguous()
        return True

    def load_blobs(self, blobs):
        """The HDF5 weight route (Net::CopyTrainedLayersFromHDF5, net.cpp:853-872): hdf5_load_nd_dataset fills blobs_[0..2] with what the file
        stores -- no adjust_blobs, no `recompute_mean` condition, no re-derivation of one mean from the other.  A layer that does not
        re-compute its mean never reads those blobs (.cu:593-621 is the only reader), so the loaded values only matter with
        `recompute_mean > 0`.  The per-pixel mean must have the layer's own crop size (the reference would reshape the blob to the
        dataset's dims and index it with its own)."""
        import numpy as np
        dev = getattr(self, "device_", None) or torch.device("cpu")
        CHECK(len(blobs) <= 3, f"Incompatible number of blobs for layer {self.layer_param_.name}")                # net.cpp:849-850
        channels, ch, cw = self.channels_, self.cropped_height_, self.cropped_width_
        if len(blobs) >= 1:
            self.num_iter_ = int(np.asarray(blobs[0], np.float32).reshape(-1)[0])
        if len(blobs) >= 2:
            # hdf5_load_nd_dataset reshapes blobs_[1] to whatever the dataset holds (util/hdf5.cpp:49-52).  Only a layer with mean_per_pixel
            # ever indexes it (with its OWN crop size): there a stored mean of another size is refused; otherwise it is kept as stored
            b1 = np.ascontiguousarray(blobs[1], np.float32)
            if b1.size == channels * ch * cw:
                self.mean_pixel_ = torch.from_numpy(b1.reshape(channels, ch, cw).copy()).to(dev)
            else:
                CHECK(not self.layer_param_.augmentation_param.get("mean_per_pixel", True) or int(self.layer_param_.augmentation_param.get("recompute_mean", 0)) <= 0,
                      "data augmentation mean: the stored per-pixel mean has another size than this layer's crop")
                self.mean_pixel_ = torch.from_numpy(b1.copy()).to(dev)
        if len(blobs) >= 3:
            b2 = np.ascontiguousarray(blobs[2], np.float32).reshape(-1)
            CHECK(b2.size == channels, "data augmentation mean: the stored per-channel mean has another channel count")
            self.mean_channel_ = torch.from_numpy(b2.copy()).to(dev)
        return True

    def set_mean(self, per_channel=None, per_pixel=None):
        CHECK((per_channel is None) != (per_pixel is None), "give exactly one of per_channel / per_pixel")
        self.mean_ = (per_channel if per_channel is not None else per_pixel).contiguous().float()
        self.mean_mode_ = ops.MEAN_PER_CHANNEL if per_channel is not None else ops.MEAN_PER_PIXEL

    def Reshape(self, bottom, top):
        CHECK(1 <= len(bottom) <= 2, "Data augmentation layer takes one or two input blobs.")          # cpp:78-79
        CHECK(1 <= len(top) <= 2, "Data augmentation layer outputs one or two output blobs.")          # cpp:80-81
        ap = self.layer_param_.augmentation_param
        num, channels, height, width = bottom[0].num(), bottom[0].channels(), bottom[0].height(), bottom[0].width()
        self.output_params_, self.input_params_ = len(top) > 1, len(bottom) > 1
        self.do_cropping_ = "crop_width" in ap and "crop_height" in ap                                 # cpp:94
        if self.do_cropping_:
            self.cropped_width_, self.cropped_height_ = int(ap["crop_width"]), int(ap["crop_height"])
            CHECK(width >= self.cropped_width_, "crop width greater than original")                     # cpp:103
            CHECK(height >= self.cropped_height_, "crop height greater than original")                  # cpp:104
        else:
            self.cropped_width_, self.cropped_height_ = width, height
        self.channels_ = channels
        top[0].Reshape(num, channels, self.cropped_height_, self.cropped_width_)                        # cpp:108
        if self.output_params_:
            top[1].Reshape(num, ops.AUG_NUM_PARAMS, 1, 1)                                               # cpp:121-126
        mean = list(ap.get("mean", []))
        if len(mean) == 3 and not ap.get("mean_per_pixel", True):                                       # cpp:142-151
            self.mean_host_ = [float(v) for v in mean]
            self.mean_mode_ = ops.MEAN_PER_CHANNEL
        self.params_ = ops.data_aug_params(self.cropped_width_ if self.do_cropping_ else 0, self.cropped_height_ if self.do_cropping_ else 0,
                                           ap.get("max_multiplier", 255.0), ap.get("chromatic_eigvec") or None, self.mean_mode_)

    def Forward_gpu(self, bottom, top):
        if self.mean_ is None and getattr(self, "mean_host_", None) is not None:
            self.mean_ = torch.tensor(self.mean_host_, dtype=torch.float32, device=bottom[0].data.device)
        self.params_.mean_mode = self.mean_mode_
        self.num_iter_ += 1                                                                             # .cu:350
        self.params_.noise_seed, self.params_.noise_stream = self.seed_, self.num_iter_
        coeffs = bottom[1].data if self.input_params_ else self._draw(bottom)
        recompute = int(self.layer_param_.augmentation_param.get("recompute_mean", 0))
        if recompute > 0:                                                                               # .cu:593-621
            self.params_.mean_mode = ops.MEAN_NONE
            out = _wrap(ops.data_augmentation_forward, self.params_, bottom[0].data, coeffs, None)
            if self.mean_pixel_ is None:
                self.mean_pixel_ = torch.zeros(out.shape[1:], dtype=torch.float32, device=out.device)     # blobs_[1], cpp:113-116
            if self.mean_channel_ is None:
                self.mean_channel_ = torch.zeros(out.shape[1], dtype=torch.float32, device=out.device)    # blobs_[2]
            self.mean_pixel_, self.mean_channel_ = self.mean_pixel_.to(out.device), self.mean_channel_.to(out.device)
            if self.num_iter_ <= recompute:
                # scal(i - 1); axpy(1 / num) per sample; scal(1 / i); gemv(1 / area) -> per-channel mean   (:600-606)
                self.mean_pixel_.mul_(float(self.num_iter_ - 1))
                for n in range(out.shape[0]):
                    self.mean_pixel_.add_(out[n], alpha=1.0 / out.shape[0])
                self.mean_pixel_.mul_(1.0 / self.num_iter_)
                self.mean_channel_ = self.mean_pixel_.mean(dim=(1, 2))
            if self.layer_param_.augmentation_param.get("mean_per_pixel", True):
                out.sub_(self.mean_pixel_.unsqueeze(0))                                                   # :609-612
            else:
                out.sub_(self.mean_channel_.view(1, -1, 1, 1))                                            # :613-620
            top[0].data = out
            if self.output_params_:
                top[1].data = bottom[1].data if self.input_params_ else (torch.zeros(top[1].shape()) if coeffs is None else torch.from_numpy(coeffs).view(-1, ops.AUG_NUM_PARAMS, 1, 1))
            return
        if (coeffs is None and not self.do_cropping_ and not self.output_params_ and self.mean_mode_ == ops.MEAN_PER_CHANNEL
                and bottom[0].data.is_cuda):
            # the deploy nets' use of this layer (no crop, default coefficients: .cu:375-387, then :592-621): top = bottom - mean[c], one
            # streaming pass -- x * 1 is exact, so these are the bits of the subtraction
            self.neg_mean_ = getattr(self, "neg_mean_", None)
            if self.neg_mean_ is None or self.neg_mean_src_ is not self.mean_:
                self.neg_mean_, self.neg_mean_src_ = (-self.mean_).contiguous(), self.mean_
            top[0].data = _wrap(ops.scale_shift_forward, bottom[0].data, 1.0, self.neg_mean_)
            return
        top[0].data = _wrap(ops.data_augmentation_forward, self.params_, bottom[0].data, coeffs, self.mean_)
        if self.output_params_:                                                                         # .cu:346-347: the same blob
            if self.input_params_:
                top[1].data = bottom[1].data
            else:                                                                                       # coefficient blobs are read on the host
                top[1].data = torch.zeros(top[1].shape()) if coeffs is None else torch.from_numpy(coeffs).view(-1, ops.AUG_NUM_PARAMS, 1, 1)

    def Backward_gpu(self, top, propagate_down, bottom):
        CHECK(not any(propagate_down), "DataAugmentationLayer cannot do backward.")                    # hpp:38-41


class GenerateAugmentationParametersLayer(Layer):
    """include/caffe/layers/generate_augmentation_parameters_layer.hpp; generate_augmentation_parameters_layer.cpp:32-105, .cu:14-112.
    bottom = [any blob] (num and, if it is an image, the source size) or [coefficient blob, original images, augmented images];
    top = [coefficient blob].  Modes "add" / "replace" / "regenerate" (augmentation_param.mode, forced to "regenerate" when the single
    bottom is an image, cpp:56-60).  Host logic throughout (the reference's Forward_gpu runs on the host too); numpy's random stream."""

    def type(self): return "GenerateAugmentationParameters"
    def AllowBackward(self): return False

    def LayerSetUp(self, bottom, top):
        import numpy as np
        self.layer_param_.reshape_every_iter = False                                  # cpp:35
        self.seed_ = int(self.layer_param_.augmentation_param.get("seed", 0))

    def Reshape(self, bottom, top):
        ap = self.layer_param_.augmentation_param
        CHECK(len(bottom) in (1, 3), "Generate augmentation parameters layer takes one (any blob from which it can take num and potentially "
              "original image size) or three (aug params, orig data, augmented data) input blobs.")                                         # cpp:50
        CHECK(len(top) == 1, "Generate augmentation parameters layer outputs one output blob.")                                             # cpp:51
        self.mode_ = ap.get("mode", "add")                                                                                                   # caffe.proto:498
        if len(bottom) == 1 and (bottom[0].width() > 1 or bottom[0].height() > 1):
            self.mode_ = "regenerate"                                                                                                        # cpp:58-60
        self.num_ = bottom[0].num()
        if len(bottom) == 3:
            self.cropped_width_, self.cropped_height_ = bottom[2].width(), bottom[2].height()                                                # cpp:74-77
            self.bottomwidth_, self.bottomheight_ = bottom[1].width(), bottom[1].height()
        else:
            CHECK("crop_width" in ap and "crop_height" in ap, "Need crop_width and crop_height if there is no blob specifying these")        # cpp:79
            self.cropped_width_, self.cropped_height_ = int(ap["crop_width"]), int(ap["crop_height"])
            if bottom[0].width() > 1 or bottom[0].height() > 1:
                self.bottomwidth_, self.bottomheight_ = bottom[0].width(), bottom[0].height()
            else:
                CHECK("bottomwidth" in ap and "bottomheight" in ap, "Need bottomwidth and bottomheight if there is no blob specifying these")   # cpp:87
                self.bottomwidth_, self.bottomheight_ = int(ap["bottomwidth"]), int(ap["bottomheight"])
        CHECK(self.num_ >= 1, "Must provide num with a bottom blob or in the prototxt")                                                      # cpp:95
        top[0].Reshape(self.num_, ops.AUG_NUM_PARAMS, 1, 1)
        self.num_iter_ = 0

    def Forward_gpu(self, bottom, top):
        import numpy as np
        from . import augment
        ap = self.layer_param_.augmentation_param
        self.num_iter_ += 1                                                                                                                  # .cu:20
        gens = {k: v for k, v in ap.items() if isinstance(v, dict)}
        if not (self.layer_param_.phase == "TRAIN" or ap.get("augment_during_test", False)):
            gens = {}                                                                                                                        # .cu:35-36
        in_params = None
        if self.mode_ in ("add", "replace"):
            in_params = bottom[0].data.detach().cpu().numpy().reshape(self.num_, ops.AUG_NUM_PARAMS).astype(np.float32)
        disc = augment.discount_coeff(self.num_iter_, self.layer_param_.coeff_schedule_param)
        rng = augment.make_rng(self.seed_ ^ 0x9e3779b97f4a7c15, self.num_iter_)               # a stream of its own next to the DataAugmentation layers'
        if in_params is None:
            out = augment.draw_batch(rng, gens, self.num_, self.bottomwidth_, self.bottomheight_, self.cropped_width_, self.cropped_height_, disc,
                                     in_params=np.zeros((self.num_, ops.AUG_NUM_PARAMS), np.float32), mode="regenerate")
        else:
            out = augment.draw_batch(rng, gens, self.num_, self.bottomwidth_, self.bottomheight_, self.cropped_width_, self.cropped_height_, disc,
                                     in_params=in_params, mode=self.mode_)
        top[0].data = torch.from_numpy(out).view(self.num_, ops.AUG_NUM_PARAMS, 1, 1)          # read on the host by the consumers

    def Backward_gpu(self, top, propagate_down, bottom):
        pass


class FlowAugmentationLayer(Layer):
    """include/caffe/layers/flow_augmentation_layer.hpp; flow_augmentation_layer.cpp:30-72, .cu:92-160.
    bottom = [flow, coefficient blob of image 1, coefficient blob of image 2] (the `params` outputs of the two DataAugmentation layers)."""

    def type(self): return "FlowAugmentation"
    def AllowBackward(self): return False                                            # hpp:28

    def LayerSetUp(self, bottom, top):
        ap = self.layer_param_.augmentation_param
        CHECK(int(ap.get("crop_width", 0)) > 0, "Please enter crop width if you want to perform augmentation")     # cpp:33
        CHECK(int(ap.get("crop_height", 0)) > 0, "Please enter crop height if you want to perform augmentation")   # cpp:34
        self.layer_param_.reshape_every_iter = False                                  # cpp:35

    def Reshape(self, bottom, top):
        CHECK(len(bottom) == 3, "Flow augmentation layer takes three input blobs: FlowField, Img1TransfParams, Img2TransfParams")   # cpp:43
        CHECK(len(top) == 1, "Flow augmentation layer outputs one output blob: Augmented Flow")                                       # cpp:44
        CHECK(bottom[0].channels() == 2, "Flow data must have two channels")                                                          # cpp:52
        ap = self.layer_param_.augmentation_param
        self.cropped_width_, self.cropped_height_ = int(ap["crop_width"]), int(ap["crop_height"])
        top[0].Reshape(bottom[0].num(), 2, self.cropped_height_, self.cropped_width_)                                                 # cpp:57
        self.num_params_ = ops.AUG_NUM_PARAMS

    def Forward_gpu(self, bottom, top):
        top[0].data = _wrap(ops.flow_augmentation_forward, bottom[0].data, bottom[1].data, bottom[2].data,
                            self.cropped_height_, self.cropped_width_)

    def Backward_gpu(self, top, propagate_down, bottom):
        raise CheckError("FlowAugmentationLayer cannot do backward.")                 # hpp:38-41


class CustomDataLayer(Layer):
    """include/caffe/layers/custom_data_layer.hpp; custom_data_layer.cpp: LayerSetUp :326-633, the prefetch :138-303, Forward :664-699.

    `data_param.source` is, in the reference, the path of an LMDB environment.  The storage engine is out of scope here: `source`
    is the sequence of (key, value) records such an environment holds (any iterable of pairs or a dict); the layer sorts them by key
    and positions its cursor with the same "%08d" lower-bound lookup (MDB_SET_RANGE, :181-186).  The values are parsed on the host
    (Datum header only), their packed `data` bytes go to the GPU and fn2_custom_data_decode_forward produces the tops; the
    reference decodes on one prefetch thread and uploads fp32 blobs.  Data-parallel training: the reference SHARES this layer between the
    solver threads of one process (`ShareInParallel()`, custom_data_layer.hpp:37: successive batches go to successive solvers); with one
    process per GPU the same assignment is `data_param.world` / `data_param.rank` (ours, default 1 / 0): rank r takes batches
    r, r + world, r + 2 world, ... of the single cursor.  Not reproduced: rand_permute (std::random_shuffle seeded with
    std::srand, :29-42 -- the order depends on the C library), mean_file (a BlobProto on disk), preselection files."""

    def type(self): return "CustomData"
    def ExactNumBottomBlobs(self): return 0
    def MinTopBlobs(self): return 1

    def LayerSetUp(self, bottom, top):
        import bisect
        from . import sample_format as SF
        dp = self.layer_param_.data_param
        enc_names = {"UINT8": SF.UINT8, "UINT16FLOW": SF.UINT16FLOW, "BOOL1": SF.BOOL1}
        self.slice_point_ = [int(s) for s in dp.get("slice_point", [])]
        self.channel_encoding_ = [enc_names.get(e, e) for e in dp.get("encoding", [])]
        self.iter_ = 0
        n_slices = len(self.slice_point_) + 1
        if len(top) == n_slices:                                                               # :340-346
            self.output_labels_ = False
        elif len(top) == n_slices + 1:
            self.output_labels_ = True
        else:
            raise CheckError(f"CustomDataLayer has {len(top)} top blobs, but {n_slices} slices.")
        backend = dp.get("backend", "LEVELDB")                                                 # proto default, caffe.proto:945
        CHECK(backend in ("LMDB", 1), "LevelDB not supported by CustomData" if backend in ("LEVELDB", 0) else "Unknown database backend")
        for unsupported in ("rand_permute", "mean_file", "preselection_file"):
            CHECK(not dp.get(unsupported), f"CustomData: {unsupported} is not reproduced by flownet2_amd (see the class docstring)")
        src = dp.get("source")
        CHECK(src is not None and len(src) > 0, "mdb_env_open failed")                         # :359-361
        items = src.items() if isinstance(src, dict) else src
        recs = sorted(((k.encode() if isinstance(k, str) else bytes(k)), v) for k, v in items)
        self.keys_ = [k for k, _ in recs]
        self.values_ = [v for _, v in recs]
        self.bisect_ = bisect.bisect_left
        self.database_entries_ = len(recs)
        perm = list(range(self.database_entries_))                                             # :415-418
        self.range_start_ = int(dp.get("range_start", 0))
        self.range_end_ = int(dp.get("range_end", -1))
        if self.range_start_ < 0: self.range_start_ = 0                                        # :424-430
        if self.range_start_ >= len(perm): self.range_start_ = len(perm) - 1
        if self.range_end_ < 0 or self.range_end_ >= len(perm): self.range_end_ = len(perm) - 1
        CHECK(self.range_end_ >= self.range_start_, "Range end is before start.")
        self.range_size_ = self.range_end_ - self.range_start_ + 1
        self.permutation_vector_ = perm[self.range_start_: self.range_end_ + 1]                # :446-452
        CHECK(not dp.get("rand_skip"), "No rand_skip for CustomData layer")                    # :486-488
        self.datum_index_ = 0
        self.world_, self.rank_ = int(dp.get("world", 1)), int(dp.get("rank", 0))
        CHECK(self.world_ >= 1 and 0 <= self.rank_ < self.world_, "data_param.rank must be in [0, world)")
        d = SF.parse_datum(self.values_[0])                                                    # first record shapes the tops, :490-499
        CHECK(int(dp.get("crop_size", 0)) == 0, "Cropping currently not supported")            # :503-506
        batch = int(dp.get("batch_size", 1))
        self.batch_size_ = batch
        if self.slice_point_:                                                                  # :512-546
            CHECK(len(self.slice_point_) == len(top) - 1, f"Check failed: slice_point_.size() == top.size() - 1 ({len(self.slice_point_)} vs. {len(top) - 1})")
            CHECK(len(top) <= d.channels, "Check failed: top.size() <= datum.channels()")
            prev, slices = 0, []
            for sp in self.slice_point_:
                CHECK(sp > prev, f"Check failed: slice_point_[i] > prev ({sp} vs. {prev})")
                slices.append(sp - prev)
                prev = sp
            slices.append(d.channels - prev)
            for i in range(len(top)):
                top[i].Reshape(batch, slices[i], d.height, d.width)
        else:
            top[0].Reshape(batch, d.channels, d.height, d.width)                               # :558-560
        if self.output_labels_:
            top[len(self.slice_point_) + 1].Reshape(batch, 1, 1, 1)                            # :569-571
        self.datum_channels_, self.datum_height_, self.datum_width_ = d.channels, d.height, d.width
        CHECK(SF.sample_bytes(d.channels, d.height, d.width, self.slice_point_, self.channel_encoding_) > 0, "unreachable")
        sub = [float(x) for x in dp.get("subtract", [])]                                       # :591-610: one constant plane per listed channel
        self.mean_host_ = None
        if sub:
            CHECK(len(sub) <= d.channels, "more subtract values than channels")
            self.mean_host_ = torch.zeros(d.channels, d.height * d.width)
            for i, m in enumerate(sub):
                self.mean_host_[i] = m
        self.mean_ = None
        self.scale_ = float(dp.get("scale", 1.0))

    def Reshape(self, bottom, top):                                                            # :636-639: empty in the reference
        pass

    def _next_record(self):
        """CustomDataLayerPrefetch, :170-189: wrap around, permuted index -> "%08d" key -> first record with key >= it."""
        if self.datum_index_ >= self.range_size_:
            self.datum_index_ = 0
        db_index = self.permutation_vector_[self.datum_index_]
        pos = self.bisect_(self.keys_, b"%08d" % db_index)
        CHECK(pos < len(self.keys_), f"Internal data fetch error: Tried to fetch element {self.datum_index_} of {self.range_size_} which is in DB: {db_index}")
        self.datum_index_ += 1
        return self.values_[pos]

    def _next_batch(self):
        """The batch this rank owns: the shared cursor hands batch b to solver b % world (layer.hpp:484-490 with ShareInParallel)."""
        for _ in range(self.rank_ * self.batch_size_ if self.iter_ == 0 else (self.world_ - 1) * self.batch_size_):
            self._next_record()
        return [self._next_record() for _ in range(self.batch_size_)]

    def Forward_gpu(self, bottom, top):
        from . import sample_format as SF
        records = self._next_batch()
        # uint8 `data` records only (float_data Datums: sample_format.parse_datum + decode_batch(float_data=True)); the payload must
        # hold every byte the slicing decodes
        need = SF.sample_bytes(self.datum_channels_, self.datum_height_, self.datum_width_, self.slice_point_, self.channel_encoding_)
        samples, shape, labels = _wrap(SF.stage_records, records, top[0].device, need)
        CHECK(shape == (self.datum_channels_, self.datum_height_, self.datum_width_), "records of different shapes in one database")
        if self.mean_host_ is not None and self.mean_ is None:
            self.mean_ = self.mean_host_.to(top[0].device)
        tops = _wrap(SF.decode_batch, samples, *shape, self.slice_point_, self.channel_encoding_, self.mean_, self.scale_)
        for i, t in enumerate(tops):
            top[i].data = t
        if self.output_labels_:
            top[len(self.slice_point_) + 1].data = torch.tensor(labels, dtype=torch.float32, device=top[0].device).view(-1, 1, 1, 1)
        self.iter_ += 1

    def Backward_gpu(self, top, propagate_down, bottom):                                       # hpp:49-52: empty
        pass


class LayerRegistry:
    """include/caffe/layer_factory.hpp:53-114."""
    _registry: Dict[str, Callable[[LayerParameter], Layer]] = {}

    @classmethod
    def AddCreator(cls, type_: str, creator):
        CHECK(type_ not in cls._registry, f"Layer type {type_} already registered.")   # layer_factory.hpp:69-70
        cls._registry[type_] = creator

    @classmethod
    def CreateLayer(cls, param: LayerParameter) -> Layer:
        CHECK(param.type in cls._registry,
              f"Unknown layer type: {param.type} (known types: {', '.join(sorted(cls._registry))})")   # :79-80
        return cls._registry[param.type](param)

    @classmethod
    def LayerTypeList(cls):
        return sorted(cls._registry)


def REGISTER_LAYER_CLASS(type_: str, klass):
    LayerRegistry.AddCreator(type_, klass)


REGISTER_LAYER_CLASS("Correlation", CorrelationLayer)      # correlation_layer.cpp:102-103
REGISTER_LAYER_CLASS("Correlation1D", Correlation1DLayer)  # correlation_layer1d.cpp:108-109
REGISTER_LAYER_CLASS("FlowWarp", FlowWarpLayer)            # flow_warp_layer.cpp:259-260
REGISTER_LAYER_CLASS("Resample", ResampleLayer)            # resample_layer.cpp:71-72
REGISTER_LAYER_CLASS("L1Loss", L1LossLayer)                # l1loss_layer.cpp:108-109
REGISTER_LAYER_CLASS("ChannelNorm", ChannelNormLayer)      # channel_norm_layer.cpp:193-194
REGISTER_LAYER_CLASS("Downsample", DownsampleLayer)        # downsample_layer.cpp:78-79
REGISTER_LAYER_CLASS("DataAugmentation", DataAugmentationLayer)   # data_augmentation_layer.cpp:218-219
REGISTER_LAYER_CLASS("GenerateAugmentationParameters", GenerateAugmentationParametersLayer)   # generate_augmentation_parameters_layer.cpp:118-119
REGISTER_LAYER_CLASS("FlowAugmentation", FlowAugmentationLayer)   # flow_augmentation_layer.cpp:88-89
REGISTER_LAYER_CLASS("CustomData", CustomDataLayer)        # custom_data_layer.cpp:712-713

from . import stock_layers  # noqa: E402,F401  (registers Convolution / Deconvolution / ReLU / Eltwise / Concat / Slice / Silence / Input)
