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

    InsertSplits         util/insert_splits.cpp:12-86 a Split layer behind every top with more than one consumer (a loss weight counts as
                                                    one) -- for nets built with backward (TRAIN phase by default): the consumers' diffs are
                                                    summed by SplitLayer::Backward_gpu
    need-backward flags  net.cpp:95-104,164-256     blob / layer / bottom flags from lr_mult, AllowBackward, propagate_down, the blobs under a
                                                    loss, force_backward
    Net::BackwardFromTo  net.cpp:592-602            layers in reverse order, Layer::Backward with the bottom flags
    ClearParamDiffs / Update  net.cpp:949-967, 929-934   parameter diffs are ACCUMULATED by the layers and cleared per iteration

Not reproduced: the Solver (learning-rate policies, momentum / Adam history, snapshots): `Update()` applies the diffs as they are, a caller
scales them.  One executor-level optimisation: a ReLU that runs in place on the top of the Convolution / Deconvolution directly in front
of it is folded into that layer (one fused kernel, as in nets.py); results are unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import prototxt
from .layers import Blob, CHECK, CheckError, Layer, LayerParameter, LayerRegistry


def _state_meets_rule(state: Dict, rule: Dict) -> bool:
    """Net::StateMeetsRule, net.cpp:319-382: phase, min_level / max_level, every `stage` present, no `not_stage` present."""
    if "phase" in rule and str(rule["phase"]) != state["phase"]:
        return False
    if "min_level" in rule and state["level"] < int(rule["min_level"]):
        return False
    if "max_level" in rule and state["level"] > int(rule["max_level"]):
        return False
    if any(str(s) not in state["stage"] for s in rule.get("stage", [])):
        return False
    return not any(str(s) in state["stage"] for s in rule.get("not_stage", []))


def _layer_included(lp: LayerParameter, state: Dict) -> bool:
    """Net::FilterNet, net.cpp:290-317: no include rules -> included unless an exclude rule is met; else included iff an include rule is met."""
    CHECK(not (lp.include and lp.exclude), "Specify either include rules or exclude rules; not both.")      # net.cpp:297-298
    if not lp.include:
        return not any(_state_meets_rule(state, r) for r in lp.exclude)
    return any(_state_meets_rule(state, r) for r in lp.include)


def _split_layer_name(layer_name: str, blob_name: str, blob_idx: int) -> str:            # insert_splits.cpp:111-117
    return "%s_%s_%d_split" % (blob_name, layer_name, blob_idx)


def _split_blob_name(layer_name: str, blob_name: str, blob_idx: int, split_idx: int) -> str:   # insert_splits.cpp:119-126
    return "%s_%s_%d_split_%d" % (blob_name, layer_name, blob_idx, split_idx)


def _insert_splits(lps: List[LayerParameter], net_inputs: List[str], phase: str) -> List[LayerParameter]:
    """InsertSplits, util/insert_splits.cpp:12-86: every top (or legacy net input: pseudo-layer -1 named "input") that feeds more than one
    bottom -- a non-zero loss weight counts as a consumer -- gets a Split layer behind it, and the consumers read its tops
    `<blob>_<layer>_<top index>_split_<k>` in order of appearance."""
    import copy
    last_top: Dict[str, tuple] = {n: (-1, j) for j, n in enumerate(net_inputs)}
    source: Dict[tuple, tuple] = {}
    count: Dict[tuple, int] = {}
    loss_w: Dict[tuple, float] = {}
    names = {-1: "input"}
    for i, lp in enumerate(lps):
        names[i] = lp.name
        for j, b in enumerate(lp.bottom):
            CHECK(b in last_top, f"Unknown bottom blob '{b}' (layer '{lp.name}', bottom index {j})")
            source[(i, j)] = last_top[b]
            count[last_top[b]] = count.get(last_top[b], 0) + 1
        for j, t in enumerate(lp.top):
            last_top[t] = (i, j)
        for j in range(min(len(lp.loss_weight), len(lp.top))):
            idx = last_top[lp.top[j]]
            loss_w[idx] = float(lp.loss_weight[j])
            if loss_w[idx]:
                count[idx] = count.get(idx, 0) + 1

    def split_layer(layer_name, blob_name, j, n, lw):
        d = {"name": _split_layer_name(layer_name, blob_name, j), "type": "Split", "bottom": [blob_name],
             "top": [_split_blob_name(layer_name, blob_name, j, k) for k in range(n)]}
        if lw:
            d["loss_weight"] = [lw] + [0.0] * (n - 1)
        return LayerParameter.from_dict(d, phase)

    out: List[LayerParameter] = []
    next_split: Dict[tuple, int] = {}
    for j, n in enumerate(net_inputs):
        if count.get((-1, j), 0) > 1:
            out.append(split_layer("input", n, j, count[(-1, j)], 0.0))
    for i, lp in enumerate(lps):
        lp = copy.deepcopy(lp)
        for j in range(len(lp.bottom)):
            src = source[(i, j)]
            if count.get(src, 0) > 1:
                k = next_split.get(src, 0)
                next_split[src] = k + 1
                lp.bottom[j] = _split_blob_name(names[src[0]], lp.bottom[j], src[1], k)
        out.append(lp)
        for j, t in enumerate(lp.top):
            if count.get((i, j), 0) > 1:
                lw = loss_w.get((i, j), 0.0)
                out.append(split_layer(lp.name, t, j, count[(i, j)], lw))
                if lw:
                    lp.loss_weight = []
                    next_split[(i, j)] = next_split.get((i, j), 0) + 1
    return out


class Net:
    def __init__(self, proto_text: str, phase: str = "TEST", device=None, backend=None, level: int = 0, stages=(), with_backward=None):
        self.phase_ = phase
        self.with_backward_ = (phase == "TRAIN") if with_backward is None else bool(with_backward)
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
        # NetState: the phase the net is built for, plus the level / stages of the prototxt's own `state { }` and the caller's (the
        # reference merges them the same way: caffe.cpp / pycaffe pass level and stages, net.cpp:290-292 reads param.state())
        st = self.param_.get("state", {}) or {}
        self.state_ = {"phase": phase, "level": int(level or st.get("level", 0)), "stage": [str(x) for x in list(st.get("stage", [])) + list(stages)]}
        self.learnable_: List[Blob] = []                # learnable parameter blobs, owners only (net.cpp:484-505)
        self.params_lr_: List[float] = []
        self.params_decay_: List[float] = []
        self._owner: Dict[str, tuple] = {}
        last_writer: Dict[str, int] = {}
        lps = [lp for lp in (LayerParameter.from_dict(ld, phase) for ld in self.param_.get("layer", [])) if _layer_included(lp, self.state_)]
        if self.with_backward_:
            lps = _insert_splits(lps, list(self.inputs), phase)                                   # net.cpp:46
        self.blob_need_backward_: Dict[int, bool] = {}          # by id(Blob), net.cpp blob_need_backward_
        self.layer_need_backward_: List[bool] = []
        self.bottom_need_backward_: List[List[bool]] = []
        self.bottom_names_: List[List[str]] = []
        self.top_names_: List[List[str]] = []
        for lp in lps:
            self._append_layer(lp, last_writer)
        self.outputs = list(self._available)            # blobs nobody consumed (net.cpp:221-230)
        self._finish_backward_flags()
        self.loss_ = None

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
        if lp.propagate_down:
            CHECK(len(lp.propagate_down) == len(lp.bottom), "propagate_down param must be specified either 0 or bottom_size times ")   # net.cpp:77-82
        bottom = []
        need_backward = False
        bnb = []
        for j, b in enumerate(lp.bottom):
            CHECK(b in self.blobs, f"Unknown bottom blob '{b}' (layer '{lp.name}', bottom index {len(bottom)})")   # net.cpp:433-434
            bottom.append(self.blobs[b])
            if b in self._available:
                self._available.remove(b)
            nb = self.blob_need_backward_.get(id(self.blobs[b]), False)                          # AppendBottom, net.cpp:441-446
            need_backward |= nb                                                                   # net.cpp:102-103
            bnb.append(bool(lp.propagate_down[j]) if lp.propagate_down else nb)
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
        # AutoTopBlobs (loss layers): anonymous tops the prototxt did not name; nobody can consume them (net.cpp:116-130)
        if layer.AutoTopBlobs():
            while len(top) < max(layer.MinTopBlobs(), layer.ExactNumTopBlobs()):
                top.append(Blob(device=self.device_))
        layer.SetUp(bottom, top)
        for k in range(len(layer.blobs_)):                                                        # net.cpp:166-185
            spec = lp.param[k] if k < len(lp.param) and isinstance(lp.param[k], dict) else {}
            pnb = float(spec.get("lr_mult", 1.0)) != 0 and layer.AllowBackward()
            need_backward |= pnb
            layer.set_param_propagate_down(k, pnb)
        if not layer.AllowBackward():
            need_backward = False
        self.layer_need_backward_.append(need_backward)
        self.bottom_need_backward_.append(bnb)
        self.bottom_names_.append(list(lp.bottom))
        self.top_names_.append(list(lp.top))
        if need_backward:
            for t in top:
                self.blob_need_backward_[id(t)] = True
        # parameter sharing: a ParamSpec name seen before hands over that owner's blob (Net::AppendParam, net.cpp:451-540)
        CHECK(len(lp.param) <= len(layer.blobs_), f"Too many params specified for layer {lp.name}")           # net.cpp:163-165
        for k in range(len(layer.blobs_)):
            spec = lp.param[k] if k < len(lp.param) and isinstance(lp.param[k], dict) else {}
            pname = str(spec.get("name", ""))
            lr, decay = float(spec.get("lr_mult", 1.0)), float(spec.get("decay_mult", 1.0))
            if not pname or pname not in self._owner:
                if pname:
                    self._owner[pname] = (lp.name, len(self.learnable_))
                    self._shared[pname] = layer.blobs_[k]
                self.learnable_.append(layer.blobs_[k])
                self.params_lr_.append(lr)
                self.params_decay_.append(decay)
                continue
            owner_name, lid = self._owner[pname]
            mine, owner = layer.blobs_[k], self._shared[pname]
            if str(spec.get("share_mode", "STRICT")) == "PERMISSIVE":                                            # net.cpp:503-512
                CHECK(mine.count() == owner.count(),
                      f"Cannot share param '{pname}' owned by layer '{owner_name}' with layer '{lp.name}'; count mismatch.  Owner layer param "
                      f"shape is {owner.shape_string()}; sharing layer shape is {mine.shape_string()}")
            else:                                                                                               # net.cpp:513-520
                CHECK(mine.shape() == owner.shape(),
                      f"Cannot share param '{pname}' owned by layer '{owner_name}' with layer '{lp.name}'; shape mismatch.  Owner layer param "
                      f"shape is {owner.shape_string()}; sharing layer expects shape {mine.shape_string()}")
            if "lr_mult" in spec:                                                                               # net.cpp:524-531
                CHECK(lr == self.params_lr_[lid], f"Shared param '{pname}' has mismatched lr_mult.")
            if "decay_mult" in spec:                                                                            # net.cpp:532-540
                CHECK(decay == self.params_decay_[lid], f"Shared param '{pname}' has mismatched decay_mult.")
            layer.blobs_[k] = owner
        # executor peephole: a ReLU in place on a top of the Convolution / Deconvolution in front of it -- directly, or behind the ReLUs
        # that were folded for the other tops of the same layer (the siamese towers: one Convolution with two bottoms and two tops,
        # then ReLU on top a, ReLU on top b) -- is applied by that layer's own kernel
        if lp.type == "ReLU" and len(lp.bottom) == 1 and lp.top == lp.bottom:
            w = last_writer.get(lp.bottom[0])
            between = range(w + 1, len(self.layers)) if w is not None else []
            if w is not None and self.layers[w].layer_param_.type in ("Convolution", "Deconvolution") \
                    and self.layers[w].layer_param_.top.index(lp.bottom[0]) not in self.layers[w].fused_relu_tops_ \
                    and all(getattr(self.layers[j], "folded_", False) and last_writer.get(self.layers[j].layer_param_.top[0]) == w for j in between):
                self.layers[w].fused_relu_tops_[self.layers[w].layer_param_.top.index(lp.bottom[0])] = layer.negative_slope_
                layer.folded_ = True
                lp_is_folded = True
            else:
                lp_is_folded = False
        else:
            lp_is_folded = False
        if not lp_is_folded:                            # a folded ReLU leaves its blob to the convolution that now writes the activated values
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
        """`source`: a .caffemodel / .caffemodel.h5 path (a name ending in ".h5" is read as HDF5, net.cpp:804-811), or
        {layer name: {"blobs": [arrays]}} (flownet2_amd.caffemodel.read_caffemodel / read_caffemodel_h5).  Returns the ignored source layer
        names."""
        if isinstance(source, str):
            from . import caffemodel
            source = caffemodel.load_file(source)
        if getattr(source, "route", None) == "hdf5":
            return self._copy_trained_layers_from_hdf5(source)
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

    def _copy_trained_layers_from_hdf5(self, source) -> List[str]:
        """Net::CopyTrainedLayersFromHDF5, net.cpp:823-882.  What differs from the binaryproto route, all of it the reference's behaviour:
        * NO CustomCopyBlobs: a DataAugmentation layer's three blobs (iteration count, per-pixel mean, per-channel mean) are loaded as
          they are stored, whatever `recompute_mean` says (the binaryproto route only hands them to a layer that re-computes its mean and
          re-derives one mean from the other, data_augmentation_layer.cpp:162-205);
        * the source may hold FEWER blobs than the layer (CHECK_LE, :849-850); a missing blob is fine when that parameter is shared with
          an earlier layer (:859-862), otherwise "Incompatible number of blobs";
        * hdf5_load_nd_dataset reshapes the target blob to the dataset's dims without comparing shapes (util/hdf5.cpp:49-52).  Here the
          element COUNT must agree and the layer keeps its own shape (a [1,1,1,C] bias dataset fills a [C] blob): a count mismatch would
          leave the reference's layer computing on a blob of the wrong size."""
        ignored = []
        for name, src in source.items():
            layer = self.layer_by_name(name)
            if layer is None:
                ignored.append(name)                                                     # "Ignoring source layer", net.cpp:836
                continue
            blobs = src["blobs"]
            if layer.layer_param_.type == "DataAugmentation":
                layer.device_ = self.device_
                layer.load_blobs(blobs)
                continue
            CHECK(src.get("num_links", len(blobs)) <= len(layer.blobs_), f"Incompatible number of blobs for layer {name}")      # :849-850
            specs = [str(s.get("name", "")) if isinstance(s, dict) else "" for s in layer.layer_param_.param]
            for j, dst in enumerate(layer.blobs_):
                if j >= len(blobs):
                    shared = j < len(specs) and specs[j] and self._owner.get(specs[j], (name,))[0] != name
                    CHECK(shared, f"Incompatible number of blobs for layer {name}")                                              # :859-866
                    continue
                b = np.asarray(blobs[j])
                CHECK(b.size == dst.count(), f"Cannot copy param {j} weights from layer '{name}'; the HDF5 dataset holds {b.size} values "
                      f"(shape {tuple(b.shape)}), the target param {dst.count()} (shape {dst.shape_string()})")
                dst.data = torch.from_numpy(np.ascontiguousarray(b, np.float32)).reshape(dst.shape()).to(device=dst.device, dtype=torch.float32).contiguous()
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
        loss = None
        with torch.no_grad():
            for layer, bottom, top in zip(self.layers, self.bottoms_, self.tops_):
                l = layer.Forward(bottom, top)                                      # net.cpp:546-557: loss += layer_loss
                if torch.is_tensor(l):
                    loss = l if loss is None else loss + l
        self.loss_ = loss
        return OrderedDict((n, self.blobs[n].data) for n in self.outputs)

    # ---- backward ------------------------------------------------------------------------------------------------------
    def _finish_backward_flags(self):
        """net.cpp:187-256: walk the layers backwards -- a layer whose tops are under no loss needs no backward, a layer all of whose tops
        were marked to skip propagation needs none either; then force_backward."""
        under_loss, skip = set(), set()
        for i in range(len(self.layers) - 1, -1, -1):
            layer = self.layers[i]
            contributes, layer_skip = False, True
            for k, t in enumerate(self.tops_[i]):
                name = self.top_names_[i][k] if k < len(self.top_names_[i]) else None
                if layer.loss(k) or (name is not None and name in under_loss):
                    contributes = True
                if name is None or name not in skip:
                    layer_skip = False
                if contributes and not layer_skip:
                    break
            if self.layer_need_backward_[i] and layer_skip:
                self.layer_need_backward_[i] = False
                self.bottom_need_backward_[i] = [False] * len(self.bottom_need_backward_[i])
            if not contributes:
                self.layer_need_backward_[i] = False
            for j, name in enumerate(self.bottom_names_[i]):
                if contributes:
                    under_loss.add(name)
                else:
                    self.bottom_need_backward_[i][j] = False
                if not self.bottom_need_backward_[i][j]:
                    skip.add(name)
        if bool(self.param_.get("force_backward", False)):                          # net.cpp:237-256
            for i, layer in enumerate(self.layers):
                self.layer_need_backward_[i] = True
                for j in range(len(self.bottom_need_backward_[i])):
                    self.bottom_need_backward_[i][j] = self.bottom_need_backward_[i][j] or layer.AllowForceBackward(j)
                for k in range(len(layer.blobs_)):
                    layer.set_param_propagate_down(k, True)

    def Backward(self):
        """Net::Backward -> BackwardFromTo(layers - 1, 0), net.cpp:592-602,696-722."""
        CHECK(self.with_backward_, "this net was built without Split layers: construct it with phase TRAIN or with_backward=True")
        with torch.no_grad():
            for i in range(len(self.layers) - 1, -1, -1):
                if self.layer_need_backward_[i]:
                    self.layers[i].Backward(self.tops_[i], self.bottom_need_backward_[i], self.bottoms_[i])

    def ClearParamDiffs(self):                                                      # net.cpp:949-967
        for b in self.learnable_:
            b.mutable_gpu_diff().zero_()

    def ForwardBackward(self, **inputs):
        """Net::ForwardBackward (net.hpp:89-94): the loss of the forward pass (a device scalar, or None for a net without loss layers)."""
        self.forward(**inputs)
        self.Backward()
        return self.loss_

    def Update(self):
        """Net::Update -> Blob::Update, net.cpp:929-934, blob.cpp:166-188: data -= diff for every learnable parameter (the solver has
        scaled the diffs by its learning rate before; there is no solver here)."""
        with torch.no_grad():
            for b in self.learnable_:
                b.data.sub_(b.mutable_gpu_diff())
        for layer in self.layers:
            if hasattr(layer, "note_weights_changed"):
                layer.note_weights_changed()


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
