"""Prototxt-driven nets (row N1: "existing FlowNet2 prototxts ... load unmodified"): the text-format reader, the template substitution
of scripts/run-flownet.py:38-58, the layer-by-layer executor over the layer registry, and the templates authored from nets.py.
CPU: parsing, substitution, graph construction and blob shapes (SetUp / Reshape run without a GPU), weight copying by name.
GPU: the nets built from the templates compute the SAME BITS as nets.deploy_forward / flownet2_deploy_forward."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from flownet2_amd import net as fnet
from flownet2_amd import nets, prototxt, templates
from flownet2_amd.layers import CheckError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_text_format_reader():
    m = prototxt.parse("""
        name: "n"  # comment
        input: "a" input: "b"
        input_shape { dim: 1 dim: 3 dim: 8 dim: 8 }
        input_shape: { dim: [1, 3, 8, 8] }
        layer { name: 'l1' type: "Convolution" bottom: "a" top: "c"
                convolution_param { num_output: 4 kernel_size: 3 pad: 1 weight_filler { type: "constant" value: -1.5e-1 } bias_term: false }
                param { lr_mult: 1 } param { lr_mult: 2 decay_mult: 0 } }
        layer { name: "corr" type: "Correlation" bottom: "c" bottom: "c" top: "d"
                correlation_param { pad: 2 kernel_size: 1 max_displacement: 2 stride_1: 1 stride_2: 2 correlation_type: MULTIPLY } }
        str: "a\\"b\\n\\x41\\101" "cat"  hex: 0x1F oct: 017 f: 1.f neg: -.5 e: [ MAX, SUM ] inf_v: inf nan_v: nan ninf: -inf
    """)
    d = prototxt.to_dict(m)
    assert d["name"] == "n" and d["input"] == ["a", "b"] and [s["dim"] for s in d["input_shape"]] == [[1, 3, 8, 8]] * 2
    l1, corr = d["layer"]
    assert l1["convolution_param"]["kernel_size"] == [3] and l1["convolution_param"]["pad"] == [1]          # repeated in ConvolutionParameter
    assert l1["convolution_param"]["weight_filler"] == {"type": "constant", "value": -0.15} and l1["convolution_param"]["bias_term"] is False
    assert l1["param"] == [{"lr_mult": 1}, {"lr_mult": 2, "decay_mult": 0}]
    assert corr["correlation_param"]["kernel_size"] == 1 and corr["correlation_param"]["correlation_type"] == "MULTIPLY"   # scalars there
    assert d["str"] == 'a"b\nAAcat' and d["hex"] == 31 and d["oct"] == 15 and d["f"] == 1.0 and d["neg"] == -0.5 and d["e"] == ["MAX", "SUM"]
    assert d["inf_v"] == math.inf and math.isnan(d["nan_v"]) and d["ninf"] == -math.inf
    for bad in ['layer { name: "x" ', "a: }", "a b", 'x: "unterminated']:
        with pytest.raises(prototxt.ParseError):
            prototxt.parse(bad)


def test_template_substitution_is_the_reference_runner():
    v = prototxt.deploy_vars(1024, 436)                      # run-flownet.py:38-48 on a Sintel frame
    assert v == {"TARGET_WIDTH": 1024, "TARGET_HEIGHT": 436, "ADAPTED_WIDTH": 1024, "ADAPTED_HEIGHT": 448, "SCALE_WIDTH": 1.0,
                 "SCALE_HEIGHT": 436 / 448.0}
    t = "dim: $TARGET_HEIGHT$ dim: $TARGET_WIDTH$\nwidth: $ADAPTED_WIDTH$ height: $ADAPTED_HEIGHT$ diag_val: $SCALE_WIDTH$ diag_val: $SCALE_HEIGHT$\n"
    s = prototxt.substitute(t, v)
    assert s == "dim: 436 dim: 1024\nwidth: 1024 height: 448 diag_val: 1.0 diag_val: %s\n" % str(436 / 448.0)
    assert prototxt.unresolved(s) == [] and prototxt.unresolved(t) == sorted(v)
    with pytest.raises(CheckError, match="template variables"):
        fnet.Net(t, device="cpu")


def test_committed_templates_are_what_nets_py_generates():
    for kind, (fname, fn) in templates.TEMPLATES.items():
        assert open(templates.template_path(kind)).read() == fn(), fname + " is stale: python -m flownet2_amd.templates"


@pytest.mark.parametrize("kind,nlayers", [("S", 51), ("C", 61), ("2", 244)])
def test_nets_build_from_the_templates_and_take_the_weights(kind, nlayers):
    n = fnet.from_template(open(templates.template_path(kind)).read(), 436, 200, device="cpu")
    assert len(n.layers) == nlayers and n.inputs == ["img0", "img1"] and n.outputs == ["predict_flow_final"]
    assert n.blobs["predict_flow_final"].shape() == [1, 2, 200, 436]
    assert n.blobs["img0_resized"].shape() == [1, 3, 256, 448]                         # ADAPTED size: multiples of 64
    if kind != "S":
        assert n.blobs["corr"].shape() == [1, 441, 32, 56] and n.blobs["blob_redir_corr"].shape() == [1, 473, 32, 56]
        assert n.layer_by_name("conv1a").blobs_[0] is n.layer_by_name("conv1b").blobs_[0]            # siamese towers share their blobs
    if kind == "2":
        assert n.blobs["net_input1"].shape() == [1, 12, 256, 448] and n.blobs["fuse_input"].shape() == [1, 11, 256, 448]
    P = nets.init_params_flownet2(0) if kind == "2" else nets.init_params(kind, 0)
    assert n.load_param_dict(P) == ["scale_conv1"]                                     # every learnable layer found its weights
    sc = n.layer_by_name("scale_conv1")                                                # DiagonalFiller: SCALE_WIDTH, SCALE_HEIGHT
    np.testing.assert_array_equal(sc.blobs_[0].data.numpy().reshape(2, 2), np.diag(np.array([436 / 448.0, 200 / 256.0], np.float32)))
    assert sc.diagonal_ is not None
    # every ReLU behind a convolution is folded into it; only the ReLU behind the Correlation layer runs on its own
    relus = [l for l in n.layers if l.layer_param_.type == "ReLU"]
    assert sum(1 for l in relus if not l.folded_) == (0 if kind == "S" else 1)
    n.reshape_inputs(3)
    assert n.blobs["predict_flow_final"].shape() == [3, 2, 200, 436]
    # CopyTrainedLayersFrom: by layer name, shapes CHECKed, unknown source layers ignored, DataAugmentation through adjust_blobs
    src = {"conv1" if kind == "S" else "conv1a": {"blobs": [np.full((64, 6 if kind == "S" else 3, 7, 7), 2.0, np.float32), np.ones((1, 1, 1, 64), np.float32)]},
           "img0s_aug": {"blobs": [np.zeros(1, np.float32), np.zeros((1, 3, 2, 2), np.float32), np.array([.1, .2, .3], np.float32).reshape(1, 3, 1, 1)]},
           "not_in_this_net": {"blobs": [np.zeros(3, np.float32)]}}
    assert n.CopyTrainedLayersFrom(src) == ["not_in_this_net"]
    first = n.layer_by_name("conv1" if kind == "S" else "conv1a")
    assert float(first.blobs_[0].data[3, 1, 2, 2]) == 2.0 and float(first.blobs_[1].data[5]) == 1.0
    # recompute_mean: 0 in the deploy templates -> adjust_blobs takes NOTHING from the model (data_augmentation_layer.cpp:165): the proto mean stays
    aug = n.layer_by_name("img0s_aug")
    assert aug.mean_host_ == [0.411, 0.433, 0.45] and aug.mean_ is None and aug.num_iter_ == 0 and aug.mean_channel_ is None
    with pytest.raises(CheckError, match="shape mismatch"):
        n.CopyTrainedLayersFrom({"conv2" if kind == "S" else "conv2a": {"blobs": [np.zeros((128, 64, 3, 3), np.float32), np.zeros(128, np.float32)]}})


def test_train_style_augmentation_param_parses_and_draws():
    """`mean` is a repeated float in AugmentationParameter (caffe.proto:498) and an optional float in RandomGeneratorParameter (:610): a
    TRAIN-phase layer with generator sub-messages must come out of the reader in a form augment.draw_batch can use."""
    from flownet2_amd import augment
    from flownet2_amd.layers import LayerParameter
    text = """
      layer { name: "img0s_aug" type: "DataAugmentation" bottom: "img0" top: "img0_aug" top: "img0_aug_params"
        augmentation_param {
          max_multiplier: 1 augment_during_test: false recompute_mean: 1000 mean_per_pixel: false
          translate { rand_type: "uniform_bernoulli" exp: false mean: 0 spread: 0.4 prob: 1.0 }
          rotate { rand_type: "uniform_bernoulli" exp: false mean: 0 spread: 0.4 prob: 1.0 }
          zoom { rand_type: "uniform_bernoulli" exp: true mean: 0.2 spread: 0.4 prob: 1.0 }
          squeeze { rand_type: "uniform_bernoulli" exp: true mean: 0 spread: 0.3 prob: 1.0 }
          lmult_pow { rand_type: "uniform_bernoulli" exp: true mean: -0.2 spread: 0.4 prob: 1.0 }
          lmult_mult { rand_type: "uniform_bernoulli" exp: true mean: 0.0 spread: 0.4 prob: 1.0 }
          lmult_add { rand_type: "uniform_bernoulli" exp: false mean: 0 spread: 0.03 prob: 1.0 }
          noise { rand_type: "uniform_bernoulli" exp: false mean: 0.03 spread: 0.03 prob: 1.0 }
          crop_width: 448 crop_height: 320
          chromatic_eigvec: 0.51 chromatic_eigvec: 0.56 chromatic_eigvec: 0.65 chromatic_eigvec: 0.79 chromatic_eigvec: 0.01
          chromatic_eigvec: -0.62 chromatic_eigvec: 0.35 chromatic_eigvec: -0.83 chromatic_eigvec: 0.44
        } }
      layer { name: "img1s_aug" type: "DataAugmentation" bottom: "img1" bottom: "aug_params1" top: "img1_aug"
        augmentation_param { max_multiplier: 1 mean_per_pixel: false mean: [0.4, 0.41] mean: 0.42 crop_width: 448 crop_height: 320 } }
    """
    d = prototxt.to_dict(prototxt.parse(text))
    ap0, ap1 = (l["augmentation_param"] for l in d["layer"])
    assert ap0["translate"]["mean"] == 0 and ap0["zoom"]["mean"] == 0.2 and isinstance(ap0["noise"]["mean"], float)      # scalars in the generators
    assert len(ap0["chromatic_eigvec"]) == 9 and ap1["mean"] == [0.4, 0.41, 0.42]                                        # lists in the parameter
    gens = {k: v for k, v in LayerParameter.from_dict(d["layer"][0]).augmentation_param.items() if isinstance(v, dict)}
    co = augment.draw_batch(augment.make_rng(3, 1), gens, 4, 512, 384, 448, 320)
    assert co.shape == (4, 42) and np.isfinite(co).all() and len({tuple(r) for r in co.round(6)}) == 4
    for c in co:                                                                    # the draws respect the generators' ranges
        k = augment.array_to_coeff(c)
        assert abs(k["angle"]) <= 0.4 + 1e-6 and math.exp(0.2 - 0.4) - 1e-5 <= k["zoom_x"] * k["zoom_y"] and 0.0 <= k["noise"] <= 0.06 + 1e-6


def test_data_augmentation_adjust_blobs_follows_the_reference():
    """data_augmentation_layer.cpp:162-205 through Net.CopyTrainedLayersFrom (net.cpp:769-781)."""
    def build(recompute, per_pixel, crop=(6, 8)):
        return fnet.Net('input: "a" input_shape { dim: 2 dim: 3 dim: 6 dim: 8 } layer { name: "aug" type: "DataAugmentation" bottom: "a" top: "b" '
                        'augmentation_param { crop_width: %d crop_height: %d recompute_mean: %d mean_per_pixel: %s mean: 0.1 mean: 0.2 mean: 0.3 } }'
                        % (crop[1], crop[0], recompute, "true" if per_pixel else "false"), device="cpu")
    rng = np.random.default_rng(0)
    pix = rng.random((1, 3, 6, 8)).astype(np.float32)
    src = {"aug": {"blobs": [np.array([1234.0], np.float32), pix, np.array([.5, .6, .7], np.float32).reshape(1, 3, 1, 1)]}}
    a = build(0, False)
    a.CopyTrainedLayersFrom(src)                                   # recompute_mean: 0 -> nothing is taken, the proto mean is what the layer subtracts
    l = a.layer_by_name("aug")
    assert l.num_iter_ == 0 and l.mean_channel_ is None and l.mean_host_ == [0.1, 0.2, 0.3]
    b = build(5, False)
    b.CopyTrainedLayersFrom(src)                                   # per-channel: iteration count + blobs[2]
    l = b.layer_by_name("aug")
    assert l.num_iter_ == 1234
    np.testing.assert_array_equal(l.mean_channel_.numpy(), np.array([.5, .6, .7], np.float32))
    c = build(5, True)
    c.CopyTrainedLayersFrom(src)                                   # per-pixel, same size: blobs[1] and its per-channel average
    l = c.layer_by_name("aug")
    np.testing.assert_array_equal(l.mean_pixel_.numpy(), pix[0])
    np.testing.assert_allclose(l.mean_channel_.numpy(), pix[0].mean((1, 2)), rtol=1e-6)
    e = build(5, True, crop=(4, 4))
    e.CopyTrainedLayersFrom(src)                                   # per-pixel, another size: the source's channel average expanded over the plane
    l = e.layer_by_name("aug")
    assert tuple(l.mean_pixel_.shape) == (3, 4, 4)
    np.testing.assert_allclose(l.mean_pixel_.numpy(), np.broadcast_to(pix[0].mean((1, 2)).reshape(3, 1, 1), (3, 4, 4)), rtol=1e-6)
    f = build(5, True)
    f.CopyTrainedLayersFrom({"aug": {"blobs": [np.array([7.0], np.float32)]}})      # "no blobs to copy"
    assert f.layer_by_name("aug").num_iter_ == 0
    with pytest.raises(CheckError, match="channel count"):
        build(5, True).CopyTrainedLayersFrom({"aug": {"blobs": [np.zeros(1, np.float32), np.zeros((1, 2, 6, 8), np.float32)]}})


def test_slice_tops_own_their_storage_and_inputs_are_copied():
    """A top must not alias its bottom where the reference copies (slice_layer.cpp:76-95): an in-place layer behind a Slice of a batch-1
    blob would otherwise write through into the bottom.  One top / one bottom share storage like the reference (ShareData)."""
    n = fnet.Net('input: "a" input_shape { dim: 1 dim: 4 dim: 2 dim: 2 } '
                 'layer { name: "s" type: "Slice" bottom: "a" top: "lo" top: "hi" slice_param { axis: 1 slice_point: 2 } } '
                 'layer { name: "c" type: "Concat" bottom: "hi" top: "hi_c" }', device="cpu")
    x = torch.arange(16, dtype=torch.float32).reshape(1, 4, 2, 2)
    n.forward(a=x)
    a, lo, hi, hic = (n.blobs[k].data for k in ("a", "lo", "hi", "hi_c"))
    assert a.data_ptr() != x.data_ptr()                                                    # the net owns its input blob
    assert lo.data_ptr() != a.data_ptr() and hi.untyped_storage().data_ptr() != a.untyped_storage().data_ptr()
    assert hic.data_ptr() == hi.data_ptr()                                                 # concat_layer.cpp:50-53
    lo.mul_(-1.0)
    assert torch.equal(a, x) and torch.equal(hi, x[:, 2:])


def test_net_errors_are_the_references():
    with pytest.raises(CheckError, match="Unknown bottom blob 'nope'"):
        fnet.Net('input: "a" input_shape { dim: 1 dim: 1 dim: 4 dim: 4 } layer { name: "r" type: "ReLU" bottom: "nope" top: "x" }', device="cpu")
    with pytest.raises(CheckError, match="produced by multiple sources"):
        fnet.Net('input: "a" input_shape { dim: 1 dim: 1 dim: 4 dim: 4 } layer { name: "r" type: "ReLU" bottom: "a" top: "b" } '
                 'layer { name: "q" type: "ReLU" bottom: "a" top: "b" }', device="cpu")
    with pytest.raises(CheckError, match="Unknown layer type: Nope"):
        fnet.Net('input: "a" input_shape { dim: 1 dim: 1 dim: 4 dim: 4 } layer { name: "r" type: "Nope" bottom: "a" top: "b" }', device="cpu")
    # phase rules: a TRAIN-only layer is filtered out of a TEST net
    n = fnet.Net('input: "a" input_shape { dim: 1 dim: 2 dim: 4 dim: 4 } layer { name: "s" type: "Silence" bottom: "a" include { phase: TRAIN } } '
                 'layer { name: "e" type: "Eltwise" bottom: "a" top: "b" eltwise_param { coeff: 2 } }', device="cpu")
    assert n.layer_names == ["e"]
    out = n.forward(a=np.ones((1, 2, 4, 4), np.float32))
    assert float(out["b"].sum()) == 64.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind,batch,h,w", [("S", 2, 128, 192), ("C", 2, 128, 192), ("2", 1, 128, 192), ("C", 1, 100, 150)])
def test_prototxt_net_computes_the_bits_of_nets_py(kind, batch, h, w):
    """The graph built from the template, executed layer by layer through the registry, against the hand-wired graph of nets.py: the
    same kernels in the same order -> the same bits (batch-invariant mode: the routing must not depend on how the two paths batch the
    siamese towers)."""
    from flownet2_amd import functional as Fn
    P = nets.init_params_flownet2(0) if kind == "2" else nets.init_params(kind, 0)
    Pd = {k: v.cuda() for k, v in P.items()}
    rng = np.random.default_rng(7)
    i0 = torch.from_numpy(rng.integers(0, 256, (batch, 3, h, w)).astype(np.float32)).cuda()
    i1 = torch.from_numpy(np.roll(i0.cpu().numpy(), (2, -3), (2, 3)).copy()).cuda()
    Fn.set_batch_invariant(True)
    try:
        with torch.no_grad():
            want = nets.flownet2_deploy_forward(Pd, i0, i1, Fn) if kind == "2" else nets.deploy_forward(kind, Pd, i0, i1, Fn)
        n = fnet.from_template(open(templates.template_path(kind)).read(), w, h, batch=batch, device="cuda")
        assert n.load_param_dict(Pd) == ["scale_conv1"]
        got = n.forward(img0=i0, img1=i1)["predict_flow_final"]
    finally:
        Fn.set_batch_invariant(False)
    assert tuple(got.shape) == (batch, 2, h, w) and bool(torch.isfinite(got).all())
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.gpu
def test_runner_with_the_references_argument_order(tmp_path):
    """scripts/run_flownet.py model deploy.prototxt.template img0 img1 out.flo == the built-in graph's .flo, byte for byte."""
    import subprocess
    from PIL import Image
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (96, 136, 3), dtype=np.uint8)
    pa, pb = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    Image.fromarray(a).save(pa); Image.fromarray(np.roll(a, (1, 2), (0, 1))).save(pb)
    run = os.path.join(ROOT, "scripts", "run_flownet.py")
    subprocess.check_call([sys.executable, run, "seed:S", templates.template_path("S"), pa, pb, str(tmp_path / "proto.flo")])
    subprocess.check_call([sys.executable, run, "--net", "S", pa, pb, str(tmp_path / "builtin.flo")])
    assert open(tmp_path / "proto.flo", "rb").read() == open(tmp_path / "builtin.flo", "rb").read()


def test_eltwise_backward_follows_the_reference_mask_and_prod_forms():
    """EltwiseLayer::Backward (eltwise_layer.cu:68-130) through a TRAIN net: MAX hands the gradient to the LAST bottom that holds the maximum
    (MaxForward keeps the running top only where `a > b`, :16-29), PROD is the product of the other bottoms (stable_prod_grad, the default) or
    top / bottom."""
    import torch
    from flownet2_amd import net as fnet
    from flownet2_amd.layers import LayerRegistry, LayerParameter, Blob

    def run(op, vals, extra=""):
        lp = LayerParameter(name="e", type="Eltwise", bottom=["a", "b", "c"], top=["t"], eltwise_param=dict(operation=op, **({"stable_prod_grad": False} if extra else {})))
        layer = LayerRegistry.CreateLayer(lp)
        bottom = [Blob.from_tensor(torch.tensor(v, dtype=torch.float32).view(1, 1, 1, -1)) for v in vals]
        top = [Blob(device="cpu")]
        layer.SetUp(bottom, top)
        layer.Forward(bottom, top)
        top[0].diff = torch.tensor([1.0, 10.0, 100.0, 1000.0]).view(1, 1, 1, -1)
        layer.Backward(top, [True, True, True], bottom)
        return top[0].data.view(-1).tolist(), [b.diff.view(-1).tolist() for b in bottom]

    t, d = run("MAX", [[1, 5, 2, 0], [1, 5, 3, 0], [0, 5, 3, -1]])
    assert t == [1, 5, 3, 0]
    assert d[0] == [0, 0, 0, 0] and d[1] == [1, 0, 0, 1000] and d[2] == [0, 10, 100, 0]     # ties go to the last maximal bottom
    t, d = run("PROD", [[1, 2, 0, 4], [2, 3, 5, 0.5], [3, 4, 7, 2]])
    assert t == [6, 24, 0, 4] and d[0] == [6, 120, 3500, 1000] and d[1] == [3, 80, 0, 8000] and d[2] == [2, 60, 0, 2000]
    t, d = run("PROD", [[1, 2, 2, 4], [2, 3, 5, 0.5], [3, 4, 7, 2]], extra="unstable")
    assert d[0] == [6, 120, 3500, 1000] and d[2] == [2, 60, 1000, 2000]
