"""The C++ Caffe adapter (flownet2_amd/csrc/caffe_adapter/fn2_caffe_layers.cpp), compiled against the stand-in
Caffe headers and driven through LayerRegistry<float>::CreateLayer by prototxt type string -- the same way, through
the same C shim, as the reference's own layer classes in oracle/_ref.  Checks: the plug-in builds, registers the
eight type strings, enforces the reference's CHECKs, and computes what the oracle (and the reference) computes."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def test_adapter_builds_and_exports_layer_driver():
    subprocess.check_call(["bash", os.path.join(ROOT, "flownet2_amd", "csrc", "caffe_adapter", "build_adapter.sh")], stdout=subprocess.DEVNULL)
    assert ref.adapter_available()
    out = subprocess.check_output(["nm", "-D", ref.ADAPTER_SO]).decode()
    for sym in ("fn2ref_correlation", "fn2ref_correlation1d", "fn2ref_flow_augmentation", "fn2ref_flow_warp", "fn2ref_resample", "fn2ref_channel_norm", "fn2ref_downsample", "fn2ref_l1loss"):
        assert sym in out
    # the adapter must call INTO libflownet2_hip.so (undefined symbols resolved at load time), not re-implement it
    assert " U fn2_correlation_forward" in out and " U fn2_flow_warp_backward" in out


@pytest.fixture()
def adapter():
    if not ref.adapter_available():
        pytest.skip("adapter test library not built")
    ref.use("adapter")
    yield ref
    ref.use("ref")


@pytest.mark.gpu
def test_adapter_layers_match_oracle(adapter):
    b0, b1 = rnd((2, 32, 12, 20), 1), rnd((2, 32, 12, 20), 2)
    top = adapter.correlation(b0, b1, 20, 1, 20, 1, 2, 0)
    td = rnd(top.shape, 3)
    _, d0, d1 = adapter.correlation(b0, b1, 20, 1, 20, 1, 2, 0, td)
    po = oracle.corr_params(20, 1, 20, 1, 2)
    np.testing.assert_allclose(top, oracle.correlation_forward(po, b0, b1), rtol=0, atol=2e-6)
    o0, o1 = oracle.correlation_backward(po, b0, b1, td)
    np.testing.assert_allclose(d0, o0, rtol=0, atol=3e-6)
    np.testing.assert_allclose(d1, o1, rtol=0, atol=3e-6)
    # Correlation1D, DispNetCorr1D-style (left only) and both directions
    for sd in (-1, 0):
        top = adapter.correlation1d(b0, b1, 8, 1, 8, 1, 1, 0, sd)
        td = rnd(top.shape, 30)
        _, d0, d1 = adapter.correlation1d(b0, b1, 8, 1, 8, 1, 1, 0, sd, td)
        po = oracle.corr_params(8, 1, 8, 1, 1, 0, 0, sd)
        np.testing.assert_allclose(top, oracle.correlation1d_forward(po, b0, b1), rtol=0, atol=2e-6)
        o0, o1 = oracle.correlation1d_backward(po, b0, b1, td)
        np.testing.assert_allclose(d0, o0, rtol=0, atol=3e-6)
        np.testing.assert_allclose(d1, o1, rtol=0, atol=3e-6)
    # FlowAugmentation: coefficient blobs in coeff_to_array layout (mirror, dx, dy, angle, log zoom_x, log zoom_y, defaults)
    fl = rnd((2, 2, 24, 40), 31, 3.0)
    c1, c2 = np.zeros((2, 42), np.float32), np.zeros((2, 42), np.float32)
    c1[:, :6] = [[0, 0.02, -0.03, 0.1, 0.1, 0.05], [1, 0, 0.01, -0.05, 0, 0.08]]
    c2[:, :6] = [[0, -0.01, 0.02, 0.05, 0.05, 0.1], [1, 0.03, 0, 0, 0.1, 0.1]]
    got = adapter.flow_augmentation(fl, c1, c2, 16, 28)
    err = np.abs(got - oracle.flow_augmentation_forward(fl, c1, c2, 16, 28))
    assert np.quantile(err, 0.99) < 1e-4       # an i.i.d. flow: a rounding flip of the sampled pixel changes an output completely
    img, flow, g = rnd((2, 3, 24, 40), 4), rnd((2, 2, 24, 40), 5, 5.0), rnd((2, 3, 24, 40), 6)
    out, di, df = adapter.flow_warp(img, flow, 1, g)
    np.testing.assert_allclose(out, oracle.flow_warp_forward(img, flow), rtol=0, atol=1e-6)
    odi, odf = oracle.flow_warp_backward(img, flow, g)
    np.testing.assert_allclose(di, odi, rtol=0, atol=1e-5)
    np.testing.assert_allclose(df, odf, rtol=0, atol=1e-5)
    x = rnd((2, 2, 20, 28), 7)
    np.testing.assert_allclose(adapter.resample(x, 80, 112, 2, True), oracle.resample_forward(x, 80, 112, 2, True), rtol=0, atol=2e-6)
    np.testing.assert_allclose(adapter.channel_norm(x), oracle.channel_norm_forward(x), rtol=0, atol=1e-6)
    np.testing.assert_allclose(adapter.downsample(x, 5, 7), oracle.downsample_forward(x, 5, 7), rtol=0, atol=1e-6)
    pred, gt = rnd((4, 2, 10, 14), 8), rnd((4, 2, 10, 14), 9)
    gt[0, :, 2, 2] = np.nan
    loss, weighted, l0, l1 = adapter.l1loss(pred, gt, l2_per_location=True, normalize_by_num_entries=True, loss_weight=0.32)
    po = oracle.l1_params(l2_per_location=True, normalize_by_num_entries=True)
    rloss, rnorm = oracle.l1loss_forward(po, pred, gt)
    assert abs(loss - rloss) <= 1e-6 and abs(weighted - 0.32 * rloss) <= 1e-6       # Layer::Forward adds loss_weight * top
    r0, r1 = oracle.l1loss_backward(po, pred, gt, 0.32, rnorm)
    np.testing.assert_allclose(l0, r0, rtol=0, atol=1e-6)
    np.testing.assert_allclose(l1, r1, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_adapter_enforces_reference_checks(adapter):
    b = rnd((1, 4, 8, 8), 10)
    with pytest.raises(RuntimeError, match="Odd kernel size"):
        adapter.correlation(b, b, 4, 2, 4, 1, 1)
    with pytest.raises(RuntimeError, match="pad .* < max_displacement"):     # C-ABI error surfaces as LOG(FATAL)
        adapter.correlation(b, b, 1, 1, 4, 1, 1)
    with pytest.raises(RuntimeError, match="single_direction must be"):
        adapter.correlation1d(b, b, 4, 1, 4, 1, 1, 0, 2)
    with pytest.raises(RuntimeError, match="only CUBIC, LINEAR and NEAREST"):
        adapter.resample(b, 4, 4, 4, True)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_adapter_is_a_drop_in_for_the_reference_layers(adapter):
    """Same shim, same calls, reference layers vs adapter layers."""
    b0, b1 = rnd((1, 64, 16, 24), 12), rnd((1, 64, 16, 24), 13)
    td = rnd((1, 441, 16, 24), 14)
    a_top, a0, a1 = adapter.correlation(b0, b1, 20, 1, 20, 1, 2, 0, td)
    ref.use("ref")
    r_top, r0, r1 = ref.correlation(b0, b1, 20, 1, 20, 1, 2, 0, td)
    np.testing.assert_allclose(a_top, r_top, rtol=0, atol=2e-6)
    np.testing.assert_allclose(a0, r0, rtol=0, atol=3e-6)
    np.testing.assert_allclose(a1, r1, rtol=0, atol=3e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/caffe"), reason="reference tree not present (GPU box)")
def test_adapter_compiles_against_the_reference_headers(tmp_path):
    """fn2_caffe_layers.cpp against include/caffe/layer.hpp:42-53,236-324, blob.hpp, layer_factory.hpp:67-84, common.hpp of the
    reference itself (its include/ first on the path; only boost / glog / gflags / CUDA / CBLAS and the protoc output are
    stand-ins): a drift between the runnable stand-in headers (compat/) and the real plug-in interface fails here."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "flownet2_amd", "csrc", "caffe_adapter", "real_headers_check", "check.sh")
    out = subprocess.run(["bash", script, str(tmp_path / "adapter.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert os.path.getsize(str(tmp_path / "adapter.o")) > 10000


CONV_CASES = [
    # (N, Cin, H, W, Cout, kernel, stride, pad, deconv): every Convolution / Deconvolution class of the FlowNet graphs, at sizes the reference's
    # im2col + SGEMM stand-in finishes quickly -- one per kernel family the library routes to
    (2, 64, 24, 32, 128, 5, 2, 2, False),      # conv2 / conv3 class: direct kernel
    (2, 64, 16, 24, 64, 3, 1, 1, False),       # conv3_1 class: Winograd or the small-map kernel (the library decides)
    (2, 64, 16, 24, 128, 3, 2, 1, False),      # conv4 class
    (1, 256, 8, 12, 32, 1, 1, 0, False),       # conv_redir: 1x1
    (2, 128, 8, 12, 64, 4, 2, 1, True),        # deconv class: GEMM + col2im
    (2, 128, 5, 7, 64, 4, 2, 1, True),         # deconv5: a 5x7 plane, the parity-class kernel
    (2, 72, 16, 24, 64, 3, 1, 1, False),       # bottom channels that are no multiple of the kernels' channel groups: the data gradient goes through the scratch
    (2, 40, 10, 14, 64, 3, 1, 1, False),       # ... on a small map (the small-map kernel's route)
    (2, 96, 16, 24, 2, 3, 1, 1, False),        # predict_flow*: the 2-channel head (round 6: reachable by descriptor, FN2_CONV_ROUTE_HEAD)
    (2, 2, 8, 12, 2, 4, 2, 1, True),           # upsample_flow*: Deconvolution 2 -> 2 (FN2_DECONV_ROUTE_HEAD)
]
STEM_CASE = (2, 3, 64, 96, 64, 7, 2, 3, False)   # conv1: the 7x7 / 2 stem (FN2_CONV_ROUTE_STEM); forward only -- its bottom is the image (no data gradient kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES + [STEM_CASE])
def test_adapter_convolution_plugins_match_the_reference_layers(case):
    """Round 5: "Convolution" / "Deconvolution" created BY TYPE STRING through LayerRegistry are the adapter's plug-ins (the kernel family is
    picked inside libflownet2_hip.so, fn2_conv_route / fn2_deconv_route); the same blobs through the reference's own ConvolutionLayer /
    DeconvolutionLayer (oracle/_ref: conv_layer.cpp, deconv_layer.cpp, base_conv_layer.cpp, im2col.cu compiled in place)."""
    if not (ref.available() and ref.adapter_available()):
        pytest.skip("reference / adapter libraries not built")
    N, Cin, H, W, Cout, k, s, p, deconv = case
    x = rnd((N, Cin, H, W), 11)
    w = rnd((Cin, Cout, k, k) if deconv else (Cout, Cin, k, k), 12, 0.1)
    b = rnd((Cout,), 13)
    ref.use("ref")
    want = ref.convolution(x, w, b, kernel=k, stride=s, pad=p, deconv=deconv)
    ref.use("adapter")
    try:
        got = ref.convolution_by_registry(x, w, b, kernel=k, stride=s, pad=p, deconv=deconv)
        got_nb = ref.convolution_by_registry(x, w, None, kernel=k, stride=s, pad=p, deconv=deconv)
    finally:
        ref.use("ref")
    assert got.shape == want.shape
    # fp32 sums of Cin k k products in different orders (and Winograd's transforms): the tolerance of the reference's own conv tests (1e-4)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(got_nb, want - b.reshape(1, -1, 1, 1), rtol=0, atol=1e-4 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES)
def test_adapter_convolution_plugins_backward_matches_the_reference_layers(case):
    """Backward_gpu of the same plug-ins (bias, weight and data gradients on the library's own routes: fn2_conv_backward_bias / _weights / _data,
    csrc/conv_route.cpp) against ConvolutionLayer / DeconvolutionLayer::Backward_gpu of the reference (conv_layer.cu:26-60, deconv_layer.cu:27-58,
    base_conv_layer.cpp:352-393 compiled in place).  The parameter diffs start from non-zero values: both sides ACCUMULATE into them (beta = 1),
    bottom_diff is overwritten."""
    if not (ref.available() and ref.adapter_available()):
        pytest.skip("reference / adapter libraries not built")
    N, Cin, H, W, Cout, k, s, p, deconv = case
    x = rnd((N, Cin, H, W), 21)
    w = rnd((Cin, Cout, k, k) if deconv else (Cout, Cin, k, k), 22, 0.1)
    b = rnd((Cout,), 23)
    Ho, Wo = (s * (H - 1) + k - 2 * p, s * (W - 1) + k - 2 * p) if deconv else ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    g = rnd((N, Cout, Ho, Wo), 24)
    wd0, bd0 = rnd(w.shape, 25), rnd(b.shape, 26)
    ref.use("ref")
    dx, dw, db = ref.convolution_backward(x, w, b, g, wd0, bd0, kernel=k, stride=s, pad=p, deconv=deconv)
    ref.use("adapter")
    try:
        gx, gw, gb = ref.convolution_backward(x, w, b, g, wd0, bd0, kernel=k, stride=s, pad=p, deconv=deconv, by_registry=True)
    finally:
        ref.use("ref")
    # fp32 sums of N Ho Wo (weights, bias) or Cout k k (data) products in different orders: 1e-4 of the result's scale, the tolerance of the
    # reference's own convolution gradient tests (test_convolution_layer.cpp: 1e-3 relative through the gradient checker)
    for name, got, want in (("bottom_diff", gx, dx), ("weight_diff", gw, dw), ("bias_diff", gb, db)):
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * scale, err_msg=name)
    # accumulation really happened on both sides (the start values are O(1), the gradients O(10..1000))
    assert float(np.abs(dw - wd0).max()) > 1e-2 and float(np.abs(gw - wd0).max()) > 1e-2


@pytest.mark.gpu
def test_flownetc_forward_chained_from_registry_created_layers_and_the_weight_cache():
    """Round 6: every Convolution / Deconvolution of a FlowNetC graph -- stem, 5x5 / 2, Winograd, small maps, 1x1, the 2-channel heads -- is
    created BY TYPE STRING (the adapter's plug-ins) and chained with the reference's own in-place ReLU and Concat layers
    (oracle/ref_shim.cpp: fn2ref_flownetc_time, `caffe time` style, tools/caffe.cpp:346-366).  The packed weight operand is built once:
    with untouched TEST-phase weights every forward after the first reuses it; touching the weight blobs' mutable pointers between passes
    (what a solver's update does) makes every forward repack."""
    if not ref.adapter_available():
        pytest.skip("adapter library not built")
    ref.use("adapter")
    try:
        if not hasattr(ref.lib(), "fn2ref_flownetc_time"):
            pytest.skip("adapter shim built without the reference's ReLU / Concat sources")
        cached = ref.flownetc_time(1, 128, 192, warmup=2, iterations=3, use_cache=True)
        fresh = ref.flownetc_time(1, 128, 192, warmup=2, iterations=3, use_cache=False)
    finally:
        ref.use("ref")
    names = [n for n, _ in cached["layers"]]
    convs = [n for n in names if not n.endswith("_relu") and not n.startswith(("concat", "blob20", "corr"))]
    assert len(convs) == 6 + 8 + 4 * 3 + 1 == 27 and "corr" in names and "blob20" in names and names[-1] == "predict_flow2"
    assert cached["output_finite"] and fresh["output_finite"]
    assert cached["packs"] == 0 and cached["pack_reuses"] == 3 * len(convs)          # the operands of the warm-up passes serve the timed ones
    assert fresh["packs"] == 3 * len(convs) and fresh["pack_reuses"] == 0
    assert all(ms > 0 for _, ms in cached["layers"]) and cached["total_ms"] > 0
