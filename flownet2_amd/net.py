"""Prototxt-driven net: what `caffe.Net(deploy.prototxt, model.caffemodel, caffe.TEST)` + `net.forward(**inputs)` do in the reference's
runner (scripts/run-flownet.py:64-98), on the layer mirrors of this repository.

    Net::Init            src/caffe/net.cpp:40-270   layers in file order, blobs by name (an in-place layer re-uses its bottom blob,
                                                    :382-445 AppendTop), net inputs from `input:` / `input_shape` / `input_dim`
                                                    or an Input layer, phase filtering of include / exclude rules (:272-330
                                                    FilterNet: phase only), parameter sharing by ParamSpec name (:451-540 AppendParam),
                                                    every layer created by its `type:` string (layer_factory.hpp:75-84) and SetUp
    Net::ForwardFromTo   net.cpp:546-557            layer by layer, in order
    CopyTrainedLayersFrom net.cpp:752-819           source layers matched by NAME, blobs by index, shapes CHECKed, unknown source
                                                    layers ignored; DataAugmentation's mean blobs come the same way
                                                    (data_augmentation_layer.cpp:162-205)

Not reproduced: the automatic Split layers (net.cpp:46 InsertSplits -- only needed for gradient accumulation), backward, solver
state.  One executor-level optimisation: a ReLU that runs in place on the top of the Convolution / Deconvolution directly in front
of it is folded into that layer (one fused kernel, as in nets.py); results are unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import prototxt
from .layers import Blob, CHECK, CheckError, Layer, LayerParameter, LayerRegistry


def _phase_ok(lp: LayerParameter, phase: str) -> bool:
    """NetStateRule, phase only (net.cpp:272-330): any include rule must match; no exclude rule may match."""
    inc, exc = lp.include, lp.exclude
    if inc:
        if not any(str(r.get("phase", phase)) == phase for r in inc):
            return False
    return not any("phase" in r and str(r["phase"]) == phase for r in exc)


class Net:
    def __init__(self, proto_text: str, phase: str = "TEST", device=None, backend=None):
        self.phase_ = phase
        self.device_ = torch.device(device) if device is not None else torch.device("cuda")
        self.backend_ = backend
        unresolved = prototxt.unresolved(proto_text)
        CHECK(not unresolved, "prototxt still holds template variables: " + ", ".join("$%s$" % u for u in unresolved))
        self.param_ = prototxt.to_dict(prototxt.parse(proto_text))
        self.name_ = str(self.param_.get("name", ""))
        CHECK("layers" not in self.param_, "V1 `layers { }` prototxts are not supported: upgrade with the reference's upgrade_net_proto_text")
        self.blobs: "OrderedDict[str, Blob]" = OrderedDict()
        self.layers: List[Layer] = []
        self.layer_names: List[str] = []
        self.bottoms_: List[List[Blob]] = []
        self.tops_: List[List[Blob]] = []
        self.inputs: List[str] = []
        self._shared: Dict[str, Blob] = {}
        self._available: List[str] = []
        self._init_inputs()
        last_writer: Dict[str, int] = {}
        for ld in self.param_.get("layer", []):
            lp = LayerParameter.from_dict(ld, phase)
            if not _phase_ok(lp, phase):
                continue
            self._append_layer(lp, last_writer)
        self.outputs = list(self._available)            # blobs nobody consumed (net.cpp:221-230)

    # ---- construction --------------------------------------------------------------------------------------------------
    def _init_inputs(self):
        names = list(self.param_.get("input", []))
        shapes = self.param_.get("input_shape", [])
        dims = list(self.param_.get("input_dim", []))
        for i, n in enumerate(names):
            if shapes:
                CHECK(len(shapes) == len(names), "Must specify either input_shape OR deprecated input_dim, not both or neither")
                shape = [int(d) for d in shapes[i].get("dim", [])]
            else:
                CHECK(len(dims) == 4 * len(names), "Incorrect input blob dimension specifications.")         # net.cpp:111-113
                shape = [int(d) for d in dims[4 * i:4 * i + 4]]
            self.blobs[n] = Blob(*shape, device=self.device_)
            self.inputs.append(n)
            self._available.append(n)

    def _append_layer(self, lp: LayerParameter, last_writer: Dict[str, int]):
        bottom = []
        for b in lp.bottom:
            CHECK(b in self.blobs, f"Unknown bottom blob '{b}' (layer '{lp.name}', bottom index {len(bottom)})")   # net.cpp:433-434
            bottom.append(self.blobs[b])
            if b in self._available:
                self._available.remove(b)
        top = []
        for i, t in enumerate(lp.top):
            if i < len(lp.bottom) and lp.bottom[i] == t:
                top.append(self.blobs[t])                # in-place computation (net.cpp:394-398)
            else:
                CHECK(t not in self.blobs, f"Top blob '{t}' produced by multiple sources.")                   # net.cpp:399-402
                self.blobs[t] = Blob(device=self.device_)
                top.append(self.blobs[t])
            if t not in self._available:
                self._available.append(t)
        layer = LayerRegistry.CreateLayer(lp)
        layer.backend_ = self.backend_
        if lp.type == "Input":
            for t in lp.top:
                self.inputs.append(t)
        # parameter sharing: a ParamSpec name seen before hands over that owner's blob (net.cpp:451-540)
        layer.SetUp(bottom, top)
        for k, spec in enumerate(lp.param):
            pname = str(spec.get("name", "")) if isinstance(spec, dict) else ""
            if not pname or k >= len(layer.blobs_):
                continue
            if pname in self._shared:
                CHECK(self._shared[pname].shape() == layer.blobs_[k].shape(),
                      f"Cannot share param '{pname}' with layer '{lp.name}'; shape mismatch.")                 # net.cpp:503-530
                layer.blobs_[k] = self._shared[pname]
            else:
                self._shared[pname] = layer.blobs_[k]
        # executor peephole: ReLU in place on the top of the Convolution / Deconvolution directly in front of it
        if lp.type == "ReLU" and len(lp.bottom) == 1 and lp.top == lp.bottom:
            w = last_writer.get(lp.bottom[0])
            if w is not None and w == len(self.layers) - 1 and self.layers[w].layer_param_.type in ("Convolution", "Deconvolution") \
                    and len(self.layers[w].layer_param_.top) == 1:
                self.layers[w].fused_relu_ = layer.negative_slope_
                layer.folded_ = True
        for t in lp.top:
            last_writer[t] = len(self.layers)
        self.layers.append(layer)
        self.layer_names.append(lp.name)
        self.bottoms_.append(bottom)
        self.tops_.append(top)
        if hasattr(layer, "note_weights_changed"):
            layer.note_weights_changed()

    # ---- parameters ----------------------------------------------------------------------------------------------------
    def layer_by_name(self, name: str) -> Optional[Layer]:
        return self.layers[self.layer_names.index(name)] if name in self.layer_names else None

    def params(self) -> "OrderedDict[str, List[Blob]]":
        return OrderedDict((n, l.blobs_) for n, l in zip(self.layer_names, self.layers) if l.blobs_)

    def _copy_blob(self, layer_name: str, k: int, blob: Blob, src):
        src_t = src if torch.is_tensor(src) else torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32))
        want = blob.shape()
        ok = list(src_t.shape) == want or (src_t.dim() == 4 and len(want) <= 4 and list(src_t.shape) == [1] * (4 - len(want)) + want)
        CHECK(ok, f"Cannot copy param {k} weights from layer '{layer_name}'; shape mismatch.  Source param shape is "
                  f"{tuple(src_t.shape)}; target param shape is {tuple(want)}.")                               # net.cpp:783-793
        blob.data = src_t.reshape(want).to(device=blob.device, dtype=torch.float32).contiguous()

    def CopyTrainedLayersFrom(self, source) -> List[str]:
        """`source`: a .caffemodel path, or {layer name: {"blobs": [arrays]}} (flownet2_amd.caffemodel.read_caffemodel).  Returns the
        ignored source layer names."""
        if isinstance(source, str):
            from . import caffemodel
            source = caffemodel.load_file(source)
        ignored = []
        for name, src in source.items():
            layer = self.layer_by_name(name)
            if layer is None:
                ignored.append(name)                                                     # "Ignoring source layer", net.cpp:763
                continue
            blobs = src["blobs"]
            if layer.layer_param_.type == "DataAugmentation":                             # DoesUseCustomCopyBlobs: net.cpp:769-781
                layer.device_ = self.device_
                layer.adjust_blobs(blobs)                                                 # data_augmentation_layer.cpp:162-205
                continue
            CHECK(len(blobs) == len(layer.blobs_), f"Incompatible number of blobs for layer {name}")          # net.cpp:779-780
            for k, (dst, b) in enumerate(zip(layer.blobs_, blobs)):
                self._copy_blob(name, k, dst, b)
            if hasattr(layer, "note_weights_changed"):
                layer.note_weights_changed()
        return ignored

    def load_param_dict(self, P: Dict[str, "torch.Tensor"], prefix_map=None) -> List[str]:
        """Weights from a nets.py-style dict: layer `<name>` takes `<name>.w` / `<name>.b`; a layer whose ParamSpecs are named `<p>_w` /
        `<p>_b` (the siamese towers of FlowNetC) takes `<p>.w` / `<p>.b`.  Returns the layers that found nothing (they keep their fillers)."""
        missing = []
        for name, layer in zip(self.layer_names, self.layers):
            if not layer.blobs_ or layer.layer_param_.type not in ("Convolution", "Deconvolution"):
                continue
            keys = []
            specs = [str(s.get("name", "")) if isinstance(s, dict) else "" for s in layer.layer_param_.param]
            for k, suffix in enumerate((".w", ".b")[:len(layer.blobs_)]):
                cands = [name + suffix]
                if k < len(specs) and specs[k].endswith("_" + suffix[1:]):
                    cands.insert(0, specs[k][:-2] + suffix)
                keys.append(next((c for c in cands if c in P), None))
            if any(k is None for k in keys):
                missing.append(name)
                continue
            for k, key in enumerate(keys):
                self._copy_blob(name, k, layer.blobs_[k], P[key].detach())
            layer.note_weights_changed()
        return missing

    # ---- forward -------------------------------------------------------------------------------------------------------
    def forward(self, **inputs) -> "OrderedDict[str, torch.Tensor]":
        for n in self.inputs:
            if n in inputs:
                t = inputs[n]
                t = t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t, np.float32))
                b = self.blobs[n]
                CHECK(list(t.shape) == b.shape(), f"input '{n}' has shape {tuple(t.shape)}, the net expects {tuple(b.shape())}")
                d = t.to(device=self.device_, dtype=torch.float32).contiguous()
                b.data = d.clone() if d.data_ptr() == t.data_ptr() else d      # the net owns its input blobs: in-place layers must not write into the caller's tensor
        unknown = [k for k in inputs if k not in self.inputs]
        CHECK(not unknown, "Input blob arguments do not match net inputs: " + ", ".join(unknown))          # pycaffe.py:_Net_forward
        with torch.no_grad():
            for layer, bottom, top in zip(self.layers, self.bottoms_, self.tops_):
                layer.Forward(bottom, top)
        return OrderedDict((n, self.blobs[n].data) for n in self.outputs)


def from_template(template_text: str, width: int, height: int, batch: int = 1, **kw) -> Net:
    """The reference runner's recipe (run-flownet.py:38-68): substitute the six size variables, build the net."""
    text = prototxt.substitute(template_text, prototxt.deploy_vars(width, height))
    net = Net(text, **kw)
    if batch != 1:
        net.reshape_inputs(batch)
    return net


def _reshape_inputs(self: Net, batch: int):
    """A deploy template declares batch 1 (`dim: 1`); re-run SetUp with another batch size."""
    for n in self.inputs:
        s = self.blobs[n].shape()
        s[0] = batch
        self.blobs[n].Reshape(*s)
    for layer, bottom, top in zip(self.layers, self.bottoms_, self.tops_):
        layer.Reshape(bottom, top)


Net.reshape_inputs = _reshape_inputs
