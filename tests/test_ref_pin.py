"""Direct pins against oracle/_ref (the reference's own layer code compiled in place, oracle/ref_build.sh).
CPU part: the reference's CPU implementations of FlowWarp / ChannelNorm (flow_warp_layer.cpp:58-199,
channel_norm_layer.cpp:43-124) against the C oracle on randomised shapes.  GPU part: the reference's kernels
executed on the MI355X against the oracle and against our HIP kernels, at sizes beyond the golden file."""
import numpy as np
import pytest

import oracle
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("shape", [(2, 3, 13, 17), (1, 1, 5, 9), (1, 8, 24, 32)])
def test_reference_cpu_flow_warp_equals_oracle(shape):
    N, C, H, W = shape
    img, flow, g = rnd(shape, 1), rnd((N, 2, H, W), 2, 5.0), rnd(shape, 3)
    out, di, df = ref.flow_warp(img, flow, 1, g, cpu=True)
    np.testing.assert_allclose(oracle.flow_warp_forward(img, flow), out, rtol=0, atol=1e-6)
    odi, odf = oracle.flow_warp_backward(img, flow, g)
    np.testing.assert_allclose(odi, di, rtol=0, atol=1e-6)
    np.testing.assert_allclose(odf, df, rtol=0, atol=1e-6)
    nan_out = ref.flow_warp(img, flow + 1000, 2, cpu=True)
    assert np.isnan(nan_out).all() and np.isnan(oracle.flow_warp_forward(img, flow + 1000, 2)).all()


def test_reference_cpu_channel_norm_equals_oracle():
    x, g = rnd((2, 3, 7, 9), 4), rnd((2, 1, 7, 9), 5)
    top, d = ref.channel_norm(x, g, cpu=True)
    np.testing.assert_allclose(oracle.channel_norm_forward(x), top, rtol=0, atol=1e-6)
    np.testing.assert_allclose(oracle.channel_norm_backward(x, top, g), d, rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", [(12, 20, 7, 3, 5), (5, 7, 3, 2, 4), (1, 1, 2, 1, 3), (9, 1, 4, 4, 2), (24, 32, 3, 2, 1)])
def test_reference_custom_data_layer_equals_oracle(case):
    """The reference's CustomDataLayer runs on the CPU (Forward_gpu = Forward_cpu, custom_data_layer.cu:19-23): records packed by the
    oracle's writer restatement, served by the in-memory LMDB stand-in, decoded by the reference, against the oracle's decode."""
    H, W, n, batch, fw = case
    rng = np.random.default_rng(H * 1000 + W)
    recs, datas = [], []
    for r in range(n):
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        b = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        f = (rng.standard_normal((2, H, W)) * 30).astype(np.float32)
        f[rng.random((2, H, W)) < 0.1] = np.nan
        o = rng.random((H, W)) < 0.5
        datas.append(oracle.custom_data_encode_sample(a, b, f, o))
        recs.append(("%08d_x" % r, oracle.datum_serialize(9, H, W, datas[-1], r)))
    sp, enc = (3, 6, 8), (1, 1, 2, 3)
    sub = (10.5, 20.25, 30, 40, 50, 60, 0.5, -0.5, 0.25)
    order = [k % n for k in range(batch * fw)]
    samples = np.stack([np.frombuffer(datas[k], np.uint8) for k in order])
    mean = np.repeat(np.asarray(sub, np.float32), H * W)
    for scale, subtract, m in [(1.0, (), None), (0.00390625, sub, mean), (3.0, sub[:3], np.concatenate([mean[:3 * H * W], np.zeros(6 * H * W, np.float32)]))]:
        tops, _ = ref.custom_data(recs, batch, sp, enc, scale, subtract, 0, -1, fw)
        want = oracle.custom_data_decode(samples, 9, H, W, sp, enc, mean=m, scale=scale)
        for t, w in zip(tops, want):
            assert np.array_equal(t.view(np.uint32), w.view(np.uint32))
    # labels only without slicing (with slice points the reference CHECK-fails when a label top is requested, :513)
    tops, labels = ref.custom_data(recs, batch, (), (), 1.0, (), 0, -1, fw, with_labels=True)
    assert labels.tolist() == [float(k) for k in order]
    want = oracle.custom_data_decode(np.stack([np.frombuffer(datas[k], np.uint8)[:9 * H * W] for k in order]), 9, H, W)
    assert np.array_equal(tops[0], want[0])          # default encoding: every channel UINT8 (only the first 9*H*W bytes are read)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 64, 24, 40, 20, 1, 20, 1, 2, 0), (1, 256, 16, 24, 20, 1, 20, 1, 2, 0),
                                  (1, 8, 11, 13, 4, 3, 2, 1, 2, 1), (1, 12, 10, 12, 6, 1, 6, 2, 3, 0)])
def test_reference_gpu_correlation_equals_oracle_and_hip(case):
    import torch
    from flownet2_amd import ops
    N, C, H, W, pad, K, md, s1, s2, t = case
    b0, b1 = rnd((N, C, H, W), 6), rnd((N, C, H, W), 7)
    top = ref.correlation(b0, b1, pad, K, md, s1, s2, t)
    td = rnd(top.shape, 8)
    _, d0, d1 = ref.correlation(b0, b1, pad, K, md, s1, s2, t, td)
    po = oracle.corr_params(pad, K, md, s1, s2, t)
    np.testing.assert_allclose(oracle.correlation_forward(po, b0, b1), top, rtol=0, atol=2e-6)
    o0, o1 = oracle.correlation_backward(po, b0, b1, td)
    np.testing.assert_allclose(o0, d0, rtol=0, atol=3e-6)
    np.testing.assert_allclose(o1, d1, rtol=0, atol=3e-6)
    p = ops.corr_params(pad, K, md, s1, s2, t)
    dv = lambda a: torch.from_numpy(a).cuda()
    np.testing.assert_allclose(ops.correlation_forward(p, dv(b0), dv(b1)).cpu().numpy(), top, rtol=0, atol=2e-6)
    h0, h1 = ops.correlation_backward(p, dv(b0), dv(b1), dv(td))
    np.testing.assert_allclose(h0.cpu().numpy(), d0, rtol=0, atol=3e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), d1, rtol=0, atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 32, 12, 48, 16, 1, 16, 1, 1, 0, 1), (1, 64, 8, 40, 20, 1, 20, 1, 2, 0, 0),
                                  (1, 8, 11, 21, 4, 3, 2, 1, 2, 1, 0), (1, 12, 10, 25, 6, 1, 6, 2, 3, 0, 1)])
def test_reference_gpu_correlation1d_equals_oracle_and_hip(case):
    import torch
    from flownet2_amd import ops
    N, C, H, W, pad, K, md, s1, s2, t, sd = case
    b0, b1 = rnd((N, C, H, W), 16), rnd((N, C, H, W), 17)
    top = ref.correlation1d(b0, b1, pad, K, md, s1, s2, t, sd)
    td = rnd(top.shape, 18)
    _, d0, d1 = ref.correlation1d(b0, b1, pad, K, md, s1, s2, t, sd, td)
    po = oracle.corr_params(pad, K, md, s1, s2, t, 0, sd)
    np.testing.assert_allclose(oracle.correlation1d_forward(po, b0, b1), top, rtol=0, atol=2e-6)
    o0, o1 = oracle.correlation1d_backward(po, b0, b1, td)
    np.testing.assert_allclose(o0, d0, rtol=0, atol=3e-6)
    np.testing.assert_allclose(o1, d1, rtol=0, atol=3e-6)
    p = ops.corr_params(pad, K, md, s1, s2, t, False, sd)
    dv = lambda a: torch.from_numpy(a).cuda()
    np.testing.assert_allclose(ops.correlation1d_forward(p, dv(b0), dv(b1)).cpu().numpy(), top, rtol=0, atol=2e-6)
    h0, h1 = ops.correlation1d_backward(p, dv(b0), dv(b1), dv(td))
    np.testing.assert_allclose(h0.cpu().numpy(), d0, rtol=0, atol=3e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), d1, rtol=0, atol=3e-6)


@pytest.mark.gpu
def test_reference_gpu_warp_resample_downsample_equal_oracle():
    img, flow, g = rnd((2, 16, 24, 40), 9), rnd((2, 2, 24, 40), 10, 6.0), rnd((2, 16, 24, 40), 11)
    out, di, df = ref.flow_warp(img, flow, 1, g)
    np.testing.assert_allclose(oracle.flow_warp_forward(img, flow), out, rtol=0, atol=1e-6)
    odi, odf = oracle.flow_warp_backward(img, flow, g)
    np.testing.assert_allclose(odi, di, rtol=0, atol=2e-5)
    np.testing.assert_allclose(odf, df, rtol=0, atol=1e-5)
    x = rnd((2, 2, 20, 28), 12)
    for t in (1, 2, 3):
        for (ho, wo) in [(80, 112), (10, 14), (20, 28), (13, 17)]:
            np.testing.assert_allclose(oracle.resample_forward(x, ho, wo, t, True), ref.resample(x, ho, wo, t, True), rtol=0, atol=5e-6)
    x = rnd((1, 2, 40, 56), 13)
    x[0, 1, 10:30, :] = np.nan
    a, b = oracle.downsample_forward(x, 10, 14), ref.downsample(x, 10, 14)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(l2_per_location=True, normalize_by_num_entries=True), dict(l2_per_location=True, l2_prescale_by_channels=True),
                                 dict(normalize_by_num_entries=True), dict(), dict(l2_per_location=True, normalize_by_num_entries=True, plateau=0.5)])
@pytest.mark.parametrize("two", [True, False])
def test_reference_gpu_l1loss_equals_oracle_and_hip(cfg, two):
    """The reference's L1LossLayer -- its own .cpp/.cu plus the stock Eltwise / Power / Convolution layers it instantiates,
    compiled in place; only the cuBLAS calls underneath are stand-ins -- against the oracle and the fused HIP kernels."""
    import torch
    from flownet2_amd import ops
    shape = (3, 2, 17, 23)
    b0 = rnd(shape, 20, 3.0)
    b1 = rnd(shape, 21, 3.0) if two else None
    if two:
        m = np.random.default_rng(22).random((3, 1, 17, 23)) < 0.1
        b1[np.broadcast_to(m, shape)] = np.nan
    loss, weighted, d0, d1 = ref.l1loss(b0, b1, loss_weight=0.32, **cfg)
    po = oracle.l1_params(**cfg)
    ol, ncoef = oracle.l1loss_forward(po, b0, b1)
    assert abs(ol - loss) <= 2e-6 * max(1.0, abs(loss)) and abs(ol * 0.32 - weighted) <= 2e-6 * max(1.0, abs(weighted))
    o0, o1 = oracle.l1loss_backward(po, b0, b1, 0.32, ncoef)
    np.testing.assert_allclose(o0, d0, rtol=0, atol=2e-6)
    if two:
        np.testing.assert_allclose(o1, d1, rtol=0, atol=2e-6)
    dv = lambda a: torch.from_numpy(a).cuda() if a is not None else None
    p = ops.l1_params(**cfg)
    hl, ws = ops.l1loss_forward(p, dv(b0), dv(b1))
    assert abs(float(hl) - loss) <= 2e-6 * max(1.0, abs(loss))
    h0, h1 = ops.l1loss_backward(p, dv(b0), dv(b1), 0.32, ws)
    np.testing.assert_allclose(h0.cpu().numpy(), d0, rtol=0, atol=2e-6)
    if two:
        np.testing.assert_allclose(h1.cpu().numpy(), d1, rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_reference_stock_layers_pin_the_fast_paths():
    """conv1 + ReLU1, predict_flow, upsample_flow, the GEMM route and bias + ReLU against the reference's stock
    Convolution / Deconvolution / ReLU layers (their own sources; plain fp32 SGEMM stand-in underneath)."""
    import torch
    from flownet2_amd import functional as Fn, ops
    dv = lambda a: torch.from_numpy(a).cuda()
    hv = lambda t: t.cpu().numpy()

    def close(a, b, atol, what):
        s = max(1.0, float(np.abs(b).max()))
        err = float(np.abs(a - b).max())
        assert a.shape == b.shape and err <= atol * s, f"{what}: {err:.3e} > {atol * s:.3e}"

    # stem: Convolution{7,2,3} + ReLU{0.1}
    for cin in (3, 6):
        x, w, b = rnd((2, cin, 32, 64), 30, 1.0), rnd((64, cin, 7, 7), 31, 0.1), rnd((64,), 32)
        want = ref.convolution(x, w, b, kernel=7, stride=2, pad=3, relu=True)
        close(hv(ops.conv_k7s2_relu_forward(dv(x), dv(w), dv(b), 0.1)), want, 3e-6, "stem")
        close(oracle.conv_k7s2_relu_forward(x, w, b, 0.1), want, 3e-6, "stem oracle")
    # predict_flow: Convolution{3,1,1} -> 2 channels, no ReLU
    x, w, b = rnd((2, 194, 20, 28), 33), rnd((2, 194, 3, 3), 34, 0.05), rnd((2,), 35)
    want = ref.convolution(x, w, b, kernel=3, stride=1, pad=1)
    close(hv(ops.predict_flow_conv_forward(dv(x), dv(w), dv(b))), want, 3e-6, "predict_flow")
    close(oracle.predict_flow_conv_forward(x, w, b), want, 3e-6, "predict_flow oracle")
    # upsample_flow: Deconvolution{4,2,1} 2 -> 2
    x, w, b = rnd((2, 2, 10, 14), 36), rnd((2, 2, 4, 4), 37), rnd((2,), 38)
    want = ref.convolution(x, w, b, kernel=4, stride=2, pad=1, deconv=True)
    close(hv(ops.upsample_flow_deconv_forward(dv(x), dv(w), dv(b))), want, 2e-6, "upsample_flow")
    close(oracle.upsample_flow_deconv_forward(x, w, b), want, 2e-6, "upsample_flow oracle")
    # small maps: 3x3 stride 1 / 2 + ReLU (small-map / Winograd kernels), 4x4/2 deconvolution + ReLU (1x1 MFMA kernel + col2im)
    x, w, b = rnd((2, 64, 10, 14), 39), rnd((128, 64, 3, 3), 40, 0.05), rnd((128,), 41)
    for stride in (1, 2):
        want = ref.convolution(x, w, b, kernel=3, stride=stride, pad=1, relu=True)
        close(hv(Fn.conv_mfma_relu(dv(x), dv(w), dv(b), stride, 1, 0.1, True)), want, 1e-5, "conv on the small-map kernels")
        col = oracle.im2col_forward(x, 3, 1, stride)
        np.testing.assert_array_equal(hv(ops.im2col_forward(dv(x), 3, 1, stride)), col)
    wd, bd = rnd((64, 32, 4, 4), 42, 0.05), rnd((32,), 43)
    want = ref.convolution(x, wd, bd, kernel=4, stride=2, pad=1, deconv=True, relu=True)
    close(hv(Fn.deconv_gemm_relu(dv(x), dv(wd).reshape(64, 512).t().contiguous(), dv(bd), 32)), want, 1e-5, "deconv via GEMM + col2im")
    # bias + ReLU alone: the reference's bias-free convolution, then our pass
    nob = ref.convolution(x, w, None, kernel=3, stride=1, pad=1)
    close(hv(ops.bias_leaky_relu_(dv(nob), dv(b), 0.1)), ref.convolution(x, w, b, kernel=3, stride=1, pad=1, relu=True), 1e-6, "bias + ReLU")
    close(oracle.bias_leaky_relu_forward(nob, b, 0.1), ref.convolution(x, w, b, kernel=3, stride=1, pad=1, relu=True), 1e-6, "bias + ReLU oracle")


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json sizes, element by element against the reference's own kernels (oracle/_ref runs them on the same GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 256, 40, 56), (4, 256, 48, 96), (1, 256, 56, 128)])
def test_reference_gpu_correlation_at_baseline_shapes(shape):
    """All 441 displacement channels and both bottom diffs at the conv3 shapes of BASELINE.json's configs (FlowNetC batch 8
    @448x320, FlowNet2 batch 4 @768x384, batch 1 @1024x448) -- correlation_layer.cu:431-603 executed here vs the MFMA kernels."""
    import torch
    from flownet2_amd import ops
    N, C, H, W = shape
    b0, b1 = rnd(shape, 50), rnd(shape, 51)
    top = ref.correlation(b0, b1, 20, 1, 20, 1, 2, 0)
    assert top.shape == (N, 441, H, W)
    td = rnd(top.shape, 52)
    _, d0, d1 = ref.correlation(b0, b1, 20, 1, 20, 1, 2, 0, td)
    p = ops.corr_params(20, 1, 20, 1, 2)
    dv = lambda a: torch.from_numpy(a).cuda()
    mine = ops.correlation_forward(p, dv(b0), dv(b1)).cpu().numpy()
    s = max(1.0, float(np.abs(top).max()))
    assert _maxerr(mine, top) <= 2e-6 * s
    h0, h1 = ops.correlation_backward(p, dv(b0), dv(b1), dv(td))
    for mine_d, want in ((h0, d0), (h1, d1)):
        sd = max(1.0, float(np.abs(want).max()))
        assert _maxerr(mine_d.cpu().numpy(), want) <= 3e-6 * sd


@pytest.mark.gpu
def test_reference_gpu_flow_warp_resample_at_baseline_shapes():
    """FlowWarp at the FlowNet2 refinement size [4,3,384,768] (smooth + rough flow, forward and both diffs) and Resample
    [4,2,96,192] -> [384,768] (LINEAR x4 and NEAREST), flow_warp_layer.cu:357-514 / resample_layer.cu:128-206 executed here."""
    import torch
    from flownet2_amd import ops
    dv = lambda a: torch.from_numpy(a).cuda()
    N, C, H, W = 4, 3, 384, 768
    img, g = rnd((N, C, H, W), 60), rnd((N, C, H, W), 61)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    smooth = np.stack([6 * np.sin(yy / 37) + 3 * np.cos(xx / 53), 5 * np.cos(yy / 41) - 4 * np.sin(xx / 29)])[None].repeat(N, 0).astype(np.float32)
    for flow in (smooth, rnd((N, 2, H, W), 62, 8.0)):
        out, di, df = ref.flow_warp(img, flow, 1, g)
        assert _maxerr(ops.flow_warp_forward(dv(img), dv(flow), 1).cpu().numpy(), out) <= 1e-6 * max(1.0, float(np.abs(out).max()))
        hi, hf = ops.flow_warp_backward(dv(img), dv(flow), dv(g))
        # the reference accumulates the image gradient with float atomics in arrival order: compare at the summation-order tolerance
        assert _maxerr(hi.cpu().numpy(), di) <= 2e-5 * max(1.0, float(np.abs(di).max()))
        assert _maxerr(hf.cpu().numpy(), df) <= 1e-5 * max(1.0, float(np.abs(df).max()))
    x = rnd((4, 2, 96, 192), 63, 4.0)
    for t in (1, 2):            # NEAREST, LINEAR
        want = ref.resample(x, 384, 768, t, True)
        assert _maxerr(ops.resample_forward(dv(x), 384, 768, t, True).cpu().numpy(), want) <= 1e-6 * max(1.0, float(np.abs(want).max()))
    x = rnd((8, 2, 80, 112), 64, 4.0)                       # FlowNetC deploy tail: [8,2,80,112] -> [320,448]
    want = ref.resample(x, 320, 448, 2, True)
    assert _maxerr(ops.resample_forward(dv(x), 320, 448, 2, True).cpu().numpy(), want) <= 1e-6 * max(1.0, float(np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)])
def test_reference_gpu_l1loss_at_training_scales(hw):
    """L1Loss{l2_per_location, normalize_by_num_entries} at the five prediction scales of a FlowNetC training step (batch 8
    @448x320), NaN-masked ground truth: loss and both diffs vs the reference's L1LossLayer (l1loss_layer.cu:67-190)."""
    import torch
    from flownet2_amd import ops
    shape = (8, 2) + hw
    b0, b1 = rnd(shape, 70, 2.0), rnd(shape, 71, 2.0)
    m = np.random.default_rng(72).random((8, 1) + hw) < 0.07
    b1[np.broadcast_to(m, shape)] = np.nan
    cfg = dict(l2_per_location=True, normalize_by_num_entries=True)
    loss, weighted, d0, d1 = ref.l1loss(b0, b1, loss_weight=0.32, **cfg)
    dv = lambda a: torch.from_numpy(a).cuda()
    p = ops.l1_params(**cfg)
    hl, ws = ops.l1loss_forward(p, dv(b0), dv(b1))
    assert abs(float(hl) - loss) <= 2e-6 * max(1.0, abs(loss))
    h0, h1 = ops.l1loss_backward(p, dv(b0), dv(b1), 0.32, ws)
    assert _maxerr(h0.cpu().numpy(), d0) <= 2e-6
    assert _maxerr(h1.cpu().numpy(), d1) <= 2e-6
