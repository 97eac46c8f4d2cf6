"""A prototxt this repository did NOT generate: tests/fixtures/FlowNet2-C_authors_style_deploy.prototxt.template, written by hand in the
dialect of the released FlowNet2 model archive ((memory): the archive itself is not in the reference tree) -- legacy input_dim, ONE
Convolution with two bottoms and two tops per siamese stage (base_conv_layer.cpp:186-200), mean subtraction in a DataAugmentation layer
that restores its mean from the .caffemodel (recompute_mean: 1000 -> adjust_blobs), param / filler / engine fields, propagate_down, Silence,
TRAIN-only loss layers with loss_weight -- goes through Net (net.cpp:40-557 mirror) and CopyTrainedLayersFrom with weights stored as V1
`layers { }` entries with legacy num / channels / height / width blob dims (what the FlowNet 1.0 era snapshots look like), and must compute
what the hand-wired graph of nets.py computes.  CPU: graph construction, shapes, CHECK messages, weight loading.  GPU: the flow."""
import os
import sys

import numpy as np
import pytest
import torch

from flownet2_amd import caffemodel, net as fnet, nets, prototxt
from flownet2_amd.layers import CheckError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEMPLATE = os.path.join(ROOT, "tests", "fixtures", "FlowNet2-C_authors_style_deploy.prototxt.template")
MEAN = np.array([0.411, 0.433, 0.45], np.float32)


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint(field << 3) + _varint(v)


def v1_caffemodel(P, aug_iters=2000.0):
    """NetParameter{ layers = 2: V1LayerParameter{ name = 4, type = 5, blobs = 6: BlobProto{ num, channels, height, width = 1..4, data = 5 } } }
    (caffe.proto:1531-1590, :10-22): every parameter of the nets.py dict as a legacy 4-D blob, plus the three blobs of the two
    DataAugmentation layers (iteration count, per-pixel mean, per-channel mean: data_augmentation_layer.cpp:41-62)."""
    def blob(a, dims4):
        a = np.ascontiguousarray(a, np.float32)
        return _vi(1, dims4[0]) + _vi(2, dims4[1]) + _vi(3, dims4[2]) + _vi(4, dims4[3]) + _ld(5, a.tobytes())
    raw = bytearray()
    names = []
    for k in P:
        if k.endswith(".w") and k[:-2] not in names:
            names.append(k[:-2])
    for n in names:
        w, b = P[n + ".w"].numpy(), P[n + ".b"].numpy()
        v1type = 39 if n.startswith(("deconv", "upsample")) else 4                    # V1LayerParameter.DECONVOLUTION / CONVOLUTION
        body = _ld(4, n.encode()) + _vi(5, v1type) + _ld(6, blob(w, w.shape)) + _ld(6, blob(b, (1, 1, 1, b.shape[0])))
        raw += _ld(2, body)
    for n in ("img0s_aug", "img1s_aug"):
        body = (_ld(4, n.encode()) + _vi(5, 0) + _ld(6, blob(np.array([aug_iters]), (1, 1, 1, 1)))
                + _ld(6, blob(np.zeros((3, 2, 2)), (1, 3, 2, 2))) + _ld(6, blob(MEAN, (1, 3, 1, 1))))
        raw += _ld(2, body)
    return bytes(raw)


def _build(w, h, device, phase="TEST", **kw):
    text = prototxt.substitute(open(TEMPLATE).read(), prototxt.deploy_vars(w, h))
    return fnet.Net(text, phase=phase, device=device, **kw)


def test_authors_style_template_builds_and_shapes():
    n = _build(448, 320, "cpu")
    assert n.inputs == ["img0", "img1"] and n.outputs == ["predict_flow_final"]             # Silence swallows conv2b / predict_flow6
    want = {"img0_nomean_resize": [1, 3, 320, 448], "conv1a": [1, 64, 160, 224], "conv1b": [1, 64, 160, 224], "conv3b": [1, 256, 40, 56],
            "corr": [1, 441, 40, 56], "blob20": [1, 473, 40, 56], "conv6_1": [1, 1024, 5, 7], "predict_flow6": [1, 2, 5, 7],
            "concat5": [1, 1026, 10, 14], "concat2": [1, 194, 80, 112], "predict_flow2": [1, 2, 80, 112], "predict_flow_final": [1, 2, 320, 448]}
    for k, s in want.items():
        assert n.blobs[k].shape() == s, (k, n.blobs[k].shape())
    # the TRAIN-only loss layers are filtered out of the TEST net (net.cpp:290-317); 48 layers remain of 53 + 5
    assert not any(l.layer_param_.type == "L1Loss" for l in n.layers) and "flow_loss2" not in n.blobs
    # one Convolution, two bottoms, two tops, ONE pair of parameter blobs; both in-place ReLUs folded into it
    c1 = n.layer_by_name("conv1")
    assert len(c1.blobs_) == 2 and c1.blobs_[0].shape() == [64, 3, 7, 7] and c1.fused_relu_tops_ == {0: 0.1, 1: 0.1} and c1.fused_relu_ == 0.1
    relus = [l for l in n.layers if l.layer_param_.type == "ReLU"]
    assert [l.layer_param_.bottom[0] for l in relus if not l.folded_] == ["corr"]
    # learnable parameters and their multipliers (net.cpp:484-540): 24 layers with 2 blobs + the bias-free scale layer
    assert len(n.learnable_) == 2 * 24 + 1 and n.params_lr_[:2] == [1.0, 1.0] and n.params_decay_[:2] == [1.0, 0.0] and n.params_lr_[-1] == 0.0
    # a non-multiple-of-64 size: Resample to the ADAPTED size, SCALE convolution off the diagonal filler
    m = _build(500, 300, "cpu")
    assert m.blobs["img0_nomean_resize"].shape() == [1, 3, 320, 512] and m.blobs["predict_flow_final"].shape() == [1, 2, 300, 500]
    np.testing.assert_allclose(m.layer_by_name("scale_conv1").blobs_[0].data.numpy().reshape(2, 2), np.diag([500 / 512.0, 300 / 320.0]), rtol=1e-7)


def test_authors_style_checks_are_the_references():
    text = prototxt.substitute(open(TEMPLATE).read(), prototxt.deploy_vars(448, 320))
    # TRAIN phase: the loss layers are included and name ground-truth blobs the deploy file does not have (net.cpp:430-436)
    with pytest.raises(CheckError, match=r"Unknown bottom blob 'blob_gt6' \(layer 'flow_loss6', bottom index 1\)"):
        fnet.Net(text, phase="TRAIN", device="cpu")
    # two bottoms, one top: base_conv_layer.hpp:29 EqualNumBottomTopBlobs (layer.hpp:433-437)
    with pytest.raises(CheckError, match="Convolution Layer produces one top blob as output for each bottom blob input."):
        fnet.Net(text.replace('  top: "conv1a"\n  top: "conv1b"\n', '  top: "conv1a"\n', 1), device="cpu")
    # bottoms of different shapes into one Convolution (base_conv_layer.cpp:194-197)
    bad = text.replace('  bottom: "conv3a"\n  top: "conv_redir"', '  bottom: "conv3a"\n  bottom: "conv2a"\n  top: "conv_redir"\n  top: "conv_redir_b"', 1)
    with pytest.raises(CheckError, match="All inputs must have the same shape."):
        fnet.Net(bad, device="cpu")
    # propagate_down must come once per bottom (net.cpp:77-82)
    with pytest.raises(CheckError, match="propagate_down param must be specified either 0 or bottom_size times"):
        fnet.Net(text.replace('  top: "img0_nomean"\n  propagate_down: false\n', '  top: "img0_nomean"\n  propagate_down: false\n  propagate_down: false\n', 1), device="cpu")
    # more ParamSpecs than parameter blobs (net.cpp:163-165)
    with pytest.raises(CheckError, match="Too many params specified for layer scale_conv1"):
        fnet.Net(text.replace('  param {\n    lr_mult: 0\n    decay_mult: 0\n  }\n  convolution_param {\n    num_output: 2\n    bias_term: false',
                              '  param {\n    lr_mult: 0\n    decay_mult: 0\n  }\n  param {\n    lr_mult: 0\n  }\n  convolution_param {\n    num_output: 2\n    bias_term: false', 1), device="cpu")
    # include and exclude rules on one layer (net.cpp:297-298)
    with pytest.raises(CheckError, match="Specify either include rules or exclude rules; not both."):
        fnet.Net(text.replace("  include {\n    phase: TRAIN\n  }\n", "  include {\n    phase: TRAIN\n  }\n  exclude {\n    phase: TEST\n  }\n", 1), device="cpu")
    # legacy input_dim needs four values per input (net.cpp input handling / upgrade_proto)
    with pytest.raises(CheckError, match="Incorrect input blob dimension specifications."):
        fnet.Net(text.replace("input_dim: 1\ninput_dim: 3\n", "input_dim: 3\n", 1), device="cpu")


def test_state_rules_levels_and_stages():
    """Net::StateMeetsRule (net.cpp:319-382) beyond the phase: min_level / max_level / stage / not_stage, with the NetState of the
    prototxt's own `state { }` merged with the caller's."""
    proto = ('input: "a" input_dim: 1 input_dim: 2 input_dim: 4 input_dim: 4 state { level: 1 stage: "deploy" } '
             'layer { name: "lo" type: "Eltwise" bottom: "a" top: "lo" eltwise_param { coeff: 2 } include { max_level: 0 } } '
             'layer { name: "hi" type: "Eltwise" bottom: "a" top: "hi" eltwise_param { coeff: 3 } include { min_level: 1 stage: "deploy" } } '
             'layer { name: "dbg" type: "Eltwise" bottom: "a" top: "dbg" eltwise_param { coeff: 4 } include { stage: "deploy" stage: "debug" } } '
             'layer { name: "nd" type: "Eltwise" bottom: "a" top: "nd" eltwise_param { coeff: 5 } exclude { not_stage: "debug" phase: TEST } }')
    assert fnet.Net(proto, device="cpu").layer_names == ["hi"]                            # level 1, stages {deploy}: "nd" is excluded (no debug stage)
    assert fnet.Net(proto, device="cpu", stages=["debug"]).layer_names == ["hi", "dbg", "nd"]
    assert fnet.Net(proto.replace("state { level: 1 stage: \"deploy\" }", ""), device="cpu").layer_names == ["lo"]


def test_shared_params_and_auto_top_loss():
    """AppendParam's checks (net.cpp:451-540) and a loss layer without a `top:` (AutoTopBlobs, net.cpp:116-130) whose loss_weight turns
    the forward pass's return value into the weighted loss (layer.hpp:484-521, net.cpp:546-557)."""
    base = ('input: "a" input_dim: 1 input_dim: 2 input_dim: 6 input_dim: 6 input: "b" input_dim: 1 input_dim: 2 input_dim: 6 input_dim: 6 '
            'layer { name: "ca" type: "Convolution" bottom: "a" top: "ca" param { name: "w" lr_mult: 1 } param { name: "bias" lr_mult: 2 } '
            'convolution_param { num_output: 2 kernel_size: 3 pad: 1 weight_filler { type: "constant" value: 0.5 } } } '
            'layer { name: "cb" type: "Convolution" bottom: "b" top: "cb" param { name: "w" %s } param { name: "bias" lr_mult: 2 } '
            'convolution_param { num_output: 2 kernel_size: %d pad: 1 } } '
            'layer { name: "loss" type: "L1Loss" bottom: "ca" bottom: "cb" loss_weight: 0.25 l1_loss_param { l2_per_location: false } }')
    n = fnet.Net(base % ("lr_mult: 1", 3), device="cpu")
    assert n.layer_by_name("cb").blobs_[0] is n.layer_by_name("ca").blobs_[0] and len(n.learnable_) == 2 and n.params_lr_ == [1.0, 2.0]
    assert n.outputs == [] and len(n.tops_[-1]) == 1                                      # the anonymous loss top is nobody's input or output
    with pytest.raises(CheckError, match="Shared param 'w' has mismatched lr_mult."):
        fnet.Net(base % ("lr_mult: 3", 3), device="cpu")
    with pytest.raises(CheckError, match=r"Cannot share param 'w' owned by layer 'ca' with layer 'cb'; shape mismatch.  Owner layer param shape is 2 2 3 3 \(36\); "
                                         r"sharing layer expects shape 2 2 5 5 \(100\)"):
        fnet.Net(base % ("lr_mult: 1", 5), device="cpu")
    with pytest.raises(CheckError, match="count mismatch"):
        fnet.Net(base % ("share_mode: PERMISSIVE", 5), device="cpu")


def test_v1_caffemodel_loads_into_the_authors_style_net():
    P = nets.init_params("C", seed=2)
    raw = v1_caffemodel(P)
    layers = caffemodel.read_caffemodel(raw)
    assert layers["conv1"]["type"] == 4 and layers["deconv5"]["type"] == 39 and len(layers) == 26
    assert layers["conv1"]["blobs"][1].shape == (1, 1, 1, 64)                             # legacy 4-D bias blob
    n = _build(448, 320, "cpu")
    assert n.CopyTrainedLayersFrom(layers) == []
    for name in ("conv1", "conv3", "conv_redir", "deconv2", "Convolution5", "upsample_flow3to2"):
        l = n.layer_by_name(name)
        assert torch.equal(l.blobs_[0].data, P[name + ".w"]) and torch.equal(l.blobs_[1].data, P[name + ".b"])       # [1,1,1,C] matches [C] (ShapeEquals)
    for a in ("img0s_aug", "img1s_aug"):                                                   # adjust_blobs: iteration count beyond recompute_mean, mean from blobs[2]
        l = n.layer_by_name(a)
        assert l.num_iter_ == 2000 and np.array_equal(l.mean_channel_.numpy(), MEAN)
    with pytest.raises(CheckError, match="Incompatible number of blobs for layer conv2"):
        n.CopyTrainedLayersFrom({"conv2": {"blobs": [P["conv2.w"].numpy()]}})


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,batch", [(192, 128, 2), (448, 320, 1), (200, 150, 1)])
def test_authors_style_net_computes_the_flow_of_nets_py(w, h, batch):
    """Same weights through the V1 .caffemodel, same images: at sizes that are multiples of 64 (Resample = identity, so the order of
    mean subtraction and Resample does not matter) the flow has the BITS of nets.deploy_forward; at other sizes the authors' order
    (mean first, then Resample) differs from nets.py's (Resample first) by fp32 rounding only."""
    from flownet2_amd import functional as Fn
    P = nets.init_params("C", seed=2)
    Pd = {k: v.cuda() for k, v in P.items()}
    rng = np.random.default_rng(3)
    i0 = torch.from_numpy(rng.integers(0, 256, (batch, 3, h, w)).astype(np.float32)).cuda()
    i1 = torch.from_numpy(np.roll(i0.cpu().numpy(), (1, -2), (2, 3)).copy()).cuda()
    Fn.set_batch_invariant(True)
    fallbacks = Fn.LIBRARY_FALLBACKS[0]
    try:
        n = _build(w, h, "cuda")
        if batch != 1:
            n.reshape_inputs(batch)
        assert n.CopyTrainedLayersFrom(caffemodel.read_caffemodel(v1_caffemodel(P))) == []
        got = n.forward(img0=i0, img1=i1)["predict_flow_final"]
        assert Fn.LIBRARY_FALLBACKS[0] == fallbacks, "a layer of the authors'-style net left the own kernels"
        with torch.no_grad():
            want = nets.deploy_forward("C", Pd, i0, i1, Fn, mean=torch.from_numpy(MEAN).cuda())
        again = n.forward(img0=i0, img1=i1)["predict_flow_final"]
    finally:
        Fn.set_batch_invariant(False)
    assert tuple(got.shape) == (batch, 2, h, w) and bool(torch.isfinite(got).all()) and torch.equal(got, again)
    if w % 64 == 0 and h % 64 == 0:
        assert torch.equal(got, want), float((got - want).abs().max())
    else:
        epe = float(((got - want) ** 2).sum(1).sqrt().mean())
        assert epe <= 1e-4, epe
    for a in ("img0s_aug", "img1s_aug"):                                                   # frozen: the forward passes did not touch the restored mean
        assert np.array_equal(n.layer_by_name(a).mean_channel_.cpu().numpy(), MEAN) and n.layer_by_name(a).num_iter_ == 2002
