"""Pins against tests/golden/ref_golden.npz = outputs of the reference's OWN kernels (its CUDA layer sources,
compiled unchanged as HIP in oracle/_ref and run on an MI355X by tests/golden/make_golden.py).
CPU part: the C oracle reproduces them.  GPU part: the HIP kernels reproduce them through the C ABI."""
import os
import sys

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

GOLD = os.path.join(HERE, "golden", "ref_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(GOLD), reason="golden vectors not generated yet")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def close(a, b, atol):
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    s = max(1.0, float(np.nanmax(np.abs(b))) if np.isfinite(b).any() else 1.0)
    err = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max()
    assert err <= atol * s, f"max err {err:.3e} > {atol * s:.3e}"


def _corr_inputs(i):
    N, C, H, W, pad, K, md, s1, s2, t = MG.CORR[i]
    b0, b1 = MG.rnd((N, C, H, W), 100 + i), MG.rnd((N, C, H, W), 200 + i)
    return b0, b1, (pad, K, md, s1, s2, t)


@pytest.mark.parametrize("i", range(len(MG.CORR)))
def test_oracle_correlation_matches_reference_kernels(gold, i):
    b0, b1, (pad, K, md, s1, s2, t) = _corr_inputs(i)
    p = oracle.corr_params(pad, K, md, s1, s2, t)
    top = oracle.correlation_forward(p, b0, b1)
    close(top, gold[f"corr{i}_top"], 1e-6)
    td = MG.rnd(top.shape, 300 + i)
    d0, d1 = oracle.correlation_backward(p, b0, b1, td)
    close(d0, gold[f"corr{i}_d0"], 2e-6)
    close(d1, gold[f"corr{i}_d1"], 2e-6)


def _corr1d_cases():
    return [(False, i) for i in range(len(MG.CORR1D))] + [(True, i) for i in range(len(MG.CORR1D_LEFT))]


def _corr1d_check(gold, left, i, fwd, bwd, tol):
    """fwd / bwd are callables (params tuple, b0, b1[, td]) -> numpy; compared with the reference kernels' outputs outside
    the elements for which the reference reads in front of its scratch blob (left mode, sample 0, row 0)."""
    tag = f"corr1dL{i}" if left else f"corr1d{i}"
    if f"{tag}_top" not in gold:
        pytest.skip("golden arrays for this Correlation1D case not generated")
    case = (MG.CORR1D_LEFT if left else MG.CORR1D)[i]
    b0, b1, prm = MG.corr1d_inputs(i, left)
    want = gold[f"{tag}_top"]
    mt, m0 = MG.corr1d_undefined_mask(case, want.shape)
    assert mt.any() == (left and True)
    top = fwd(prm, b0, b1)
    close(np.where(mt, 0, top), np.where(mt, 0, want), tol)
    td = MG.rnd(want.shape, (1400 if left else 1200) + i)
    d0, d1 = bwd(prm, b0, b1, td)
    close(np.where(m0, 0, d0), np.where(m0, 0, gold[f"{tag}_d0"]), 2 * tol)
    close(d1, gold[f"{tag}_d1"], 2 * tol)


@pytest.mark.parametrize("left,i", _corr1d_cases())
def test_oracle_correlation1d_matches_reference_kernels(gold, left, i):
    def P(prm):
        pad, K, md, s1, s2, t, sd = prm
        return oracle.corr_params(pad, K, md, s1, s2, t, 0, sd)
    _corr1d_check(gold, left, i, lambda prm, a, b: oracle.correlation1d_forward(P(prm), a, b),
                  lambda prm, a, b, td: oracle.correlation1d_backward(P(prm), a, b, td), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("left,i", _corr1d_cases())
def test_hip_correlation1d_matches_reference_kernels(gold, left, i):
    import torch
    from flownet2_amd import ops

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    def P(prm):
        pad, K, md, s1, s2, t, sd = prm
        return ops.corr_params(pad, K, md, s1, s2, t, False, sd)

    def bwd(prm, a, b, td):
        d0, d1 = ops.correlation1d_backward(P(prm), dev(a), dev(b), dev(td))
        return d0.cpu().numpy(), d1.cpu().numpy()
    _corr1d_check(gold, left, i, lambda prm, a, b: ops.correlation1d_forward(P(prm), dev(a), dev(b)).cpu().numpy(), bwd, 2e-6)


def _cdata_expected_order(i):
    """Record index of every decoded sample: the cursor walks [range_start, range_end] and wraps (custom_data_layer.cpp:170-189)."""
    H, W, n, batch, sp, enc, scale, sub, r0, r1, fw = MG.CUSTOM_DATA[i]
    r1 = n - 1 if (r1 < 0 or r1 >= n) else r1
    span = list(range(max(r0, 0), r1 + 1))
    return [span[k % len(span)] for k in range(batch * fw)]


def _cdata_inputs(i):
    H, W, n, batch, sp, enc, scale, sub, r0, r1, fw = MG.CUSTOM_DATA[i]
    recs = MG.custom_data_records(i)
    datums = [oracle.datum_parse(v) for _, v in recs]
    channels = datums[0]["channels"]
    samples = np.stack([np.frombuffer(datums[k]["data"], np.uint8) for k in _cdata_expected_order(i)])
    mean = None
    if sub:
        mean = np.zeros((channels, H * W), np.float32)
        mean[:len(sub)] = np.asarray(sub, np.float32)[:, None]
    return recs, samples, channels, mean


@pytest.mark.parametrize("i", range(len(MG.CUSTOM_DATA)))
def test_oracle_custom_data_decode_matches_the_reference_layer(gold, i):
    """Golden = tops of the reference's CustomDataLayer (compiled in place, in-memory LMDB stand-in) over n_forward batches."""
    H, W, n, batch, sp, enc, scale, sub, r0, r1, fw = MG.CUSTOM_DATA[i]
    recs, samples, channels, mean = _cdata_inputs(i)
    tops = oracle.custom_data_decode(samples, channels, H, W, sp, enc, mean=mean, scale=scale)
    assert len(tops) == len(sp) + 1
    for s_, t in enumerate(tops):
        assert np.array_equal(t.view(np.uint32), gold[f"cdata{i}_top{s_}"]), (i, s_)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(MG.CUSTOM_DATA)))
def test_hip_custom_data_layer_matches_the_reference_layer(gold, i):
    import torch
    from flownet2_amd import sample_format as SF
    from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
    H, W, n, batch, sp, enc, scale, sub, r0, r1, fw = MG.CUSTOM_DATA[i]
    recs, samples, channels, mean = _cdata_inputs(i)
    got = SF.decode_batch(torch.from_numpy(samples).cuda(), channels, H, W, sp, enc,
                          mean=torch.from_numpy(mean).cuda() if mean is not None else None, scale=scale)
    for s_, t in enumerate(got):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), gold[f"cdata{i}_top{s_}"]), (i, s_)
    # the same through the Layer mirror, batch by batch, the way Net::Forward drives the reference layer
    lp = LayerParameter(name="data", type="CustomData",
                        data_param=dict(source=recs, backend="LMDB", batch_size=batch, slice_point=list(sp), encoding=list(enc), scale=scale,
                                        subtract=list(sub), range_start=r0, range_end=r1))
    layer = LayerRegistry.CreateLayer(lp)
    top = [Blob() for _ in range(len(sp) + 1)]
    layer.SetUp([], top)
    for f in range(fw):
        layer.Forward([], top)
        for s_, t in enumerate(top):
            want = gold[f"cdata{i}_top{s_}"][f * batch:(f + 1) * batch]
            assert t.shape() == list(want.shape)
            assert np.array_equal(t.cpu_data().view(np.uint32), want), (i, f, s_)


def test_oracle_flow_warp_matches_reference_gpu_and_cpu_code(gold):
    img, flow, wd = MG.rnd((2, 3, 13, 17), 400), MG.rnd((2, 2, 13, 17), 401, 4.0), MG.rnd((2, 3, 13, 17), 402)
    flow[0, :, 0, 0] = 0
    for fill in (1, 2):
        close(oracle.flow_warp_forward(img, flow, fill), gold[f"warp_gpu_fill{fill}"], 1e-6)
    close(oracle.flow_warp_forward(img, flow, 1), gold["warp_cpu"], 1e-6)
    di, df = oracle.flow_warp_backward(img, flow, wd)
    close(di, gold["warp_gpu_di"], 1e-5)      # reference GPU uses float atomics
    close(df, gold["warp_gpu_df"], 1e-6)
    close(di, gold["warp_cpu_di"], 1e-6)
    close(df, gold["warp_cpu_df"], 1e-6)


@pytest.mark.parametrize("i", range(len(MG.RESAMPLE)))
def test_oracle_resample_matches_reference_kernels(gold, i):
    (hi, wi), (ho, wo) = MG.RESAMPLE[i]
    x = MG.rnd((2, 2, hi, wi), 500 + i)
    for t in (1, 2, 3):
        for aa in (0, 1):
            close(oracle.resample_forward(x, ho, wo, t, bool(aa)), gold[f"resample{i}_t{t}_aa{aa}"], 2e-6)


def test_oracle_channel_norm_and_downsample_match_reference(gold):
    x = MG.rnd((2, 3, 9, 10), 600)
    top = oracle.channel_norm_forward(x)
    close(top, gold["cnorm_gpu"], 1e-6)
    close(top, gold["cnorm_cpu"], 1e-6)
    close(oracle.channel_norm_backward(x, top, MG.rnd((2, 1, 9, 10), 601)), gold["cnorm_cpu_diff"], 1e-6)
    for i, ((hi, wi), (ho, wo)) in enumerate(MG.DOWN):
        x = MG.rnd((1, 2, hi, wi), 700 + i)
        x[0, 0, :5, :7] = np.nan
        close(oracle.downsample_forward(x, ho, wo), gold[f"down{i}"], 1e-6)


def _l1_oracle(i):
    shape, two, l2, pre, norm, eps, plateau, nans, lw = MG.L1[i]
    b0, b1 = MG.l1_inputs(i)
    po = oracle.l1_params(l2_per_location=l2, l2_prescale_by_channels=pre, normalize_by_num_entries=norm, epsilon=eps, plateau=plateau)
    loss, ncoef = oracle.l1loss_forward(po, b0, b1)
    d0, d1 = oracle.l1loss_backward(po, b0, b1, lw, ncoef)
    return loss, d0, d1, lw


@pytest.mark.parametrize("i", range(len(MG.L1)))
def test_oracle_l1loss_matches_reference_layer(gold, i):
    """The reference's L1LossLayer (with its Eltwise / Power / Convolution sub-layers) executed on the MI355X."""
    if f"l1_{i}_loss" not in gold:
        pytest.skip("golden file predates the L1Loss vectors")
    loss, d0, d1, lw = _l1_oracle(i)
    ref_loss, ref_weighted = gold[f"l1_{i}_loss"]
    assert abs(loss - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    assert abs(loss * lw - ref_weighted) <= 2e-6 * max(1.0, abs(ref_weighted))
    close(d0, gold[f"l1_{i}_d0"], 2e-6)
    if d1 is not None:
        close(d1, gold[f"l1_{i}_d1"], 2e-6)


def test_oracle_stock_layer_twins_match_reference_layers(gold):
    """The oracle twins of the stock-layer fast paths against the reference's Convolution / Deconvolution / ReLU layers."""
    if "stock_stem" not in gold:
        pytest.skip("golden file predates the stock-layer vectors")
    x, w, b = MG.stock_inputs("stem")
    close(oracle.conv_k7s2_relu_forward(x, w, b, 0.1), gold["stock_stem"], 3e-6)
    x, w, b = MG.stock_inputs("predict_flow")
    close(oracle.predict_flow_conv_forward(x, w, b), gold["stock_predict_flow"], 3e-6)
    x, w, b = MG.stock_inputs("upsample_flow")
    close(oracle.upsample_flow_deconv_forward(x, w, b), gold["stock_upsample_flow"], 2e-6)
    x, w, b = MG.stock_inputs("deconv")          # weight^T x bottom in fp64 here, then the oracle's col2im + bias + ReLU
    N, Cin, H, W = x.shape
    col = np.matmul(w.reshape(Cin, -1).T.astype(np.float64), x.reshape(N, Cin, H * W).astype(np.float64)).astype(np.float32)
    close(oracle.col2im_bias_relu_forward(col, b, N, w.shape[1], 2 * H, 2 * W, 4, 1, 2, True, 0.1), gold["stock_deconv_relu"], 3e-6)
    x, w, b = MG.stock_inputs("conv3x3")         # the oracle's im2col, fp64 GEMM, then the oracle's bias + ReLU
    col = oracle.im2col_forward(x, 3, 1, 2)
    y = np.matmul(w.reshape(16, -1).astype(np.float64), col.astype(np.float64)).astype(np.float32).reshape(gold["stock_conv3x3s2_nobias"].shape)
    close(y, gold["stock_conv3x3s2_nobias"], 3e-6)
    close(oracle.bias_leaky_relu_forward(gold["stock_conv3x3s2_nobias"], b, 0.1), gold["stock_conv3x3s2_relu"], 1e-6)


@pytest.mark.gpu
def test_hip_kernels_match_reference_kernels(gold):
    import torch
    from flownet2_amd import ops

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    for i in range(len(MG.CORR)):
        b0, b1, (pad, K, md, s1, s2, t) = _corr_inputs(i)
        p = ops.corr_params(pad, K, md, s1, s2, t)
        top = ops.correlation_forward(p, dev(b0), dev(b1))
        close(top.cpu().numpy(), gold[f"corr{i}_top"], 2e-6)
        td = MG.rnd(tuple(top.shape), 300 + i)
        d0, d1 = ops.correlation_backward(p, dev(b0), dev(b1), dev(td))
        close(d0.cpu().numpy(), gold[f"corr{i}_d0"], 3e-6)
        close(d1.cpu().numpy(), gold[f"corr{i}_d1"], 3e-6)
    img, flow, wd = MG.rnd((2, 3, 13, 17), 400), MG.rnd((2, 2, 13, 17), 401, 4.0), MG.rnd((2, 3, 13, 17), 402)
    flow[0, :, 0, 0] = 0
    for fill in (1, 2):
        close(ops.flow_warp_forward(dev(img), dev(flow), fill).cpu().numpy(), gold[f"warp_gpu_fill{fill}"], 1e-6)
    di, df = ops.flow_warp_backward(dev(img), dev(flow), dev(wd))
    close(di.cpu().numpy(), gold["warp_gpu_di"], 1e-5)
    close(df.cpu().numpy(), gold["warp_gpu_df"], 1e-6)
    for i, ((hi, wi), (ho, wo)) in enumerate(MG.RESAMPLE):
        x = MG.rnd((2, 2, hi, wi), 500 + i)
        for t in (1, 2, 3):
            for aa in (0, 1):
                close(ops.resample_forward(dev(x), ho, wo, t, bool(aa)).cpu().numpy(), gold[f"resample{i}_t{t}_aa{aa}"], 2e-6)
    x = MG.rnd((2, 3, 9, 10), 600)
    close(ops.channel_norm_forward(dev(x)).cpu().numpy(), gold["cnorm_gpu"], 1e-6)
    for i, ((hi, wi), (ho, wo)) in enumerate(MG.DOWN):
        x = MG.rnd((1, 2, hi, wi), 700 + i)
        x[0, 0, :5, :7] = np.nan
        close(ops.downsample_forward(dev(x), ho, wo).cpu().numpy(), gold[f"down{i}"], 1e-6)
    for i, (shape, two, l2, pre, norm, eps, plateau, nans, lw) in enumerate(MG.L1):
        if f"l1_{i}_loss" not in gold:
            continue
        b0, b1 = MG.l1_inputs(i)
        p = ops.l1_params(l2_per_location=l2, l2_prescale_by_channels=pre, normalize_by_num_entries=norm, epsilon=eps, plateau=plateau)
        loss, ws = ops.l1loss_forward(p, dev(b0), dev(b1) if b1 is not None else None)
        ref_loss = float(gold[f"l1_{i}_loss"][0])
        assert abs(float(loss) - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss)), (i, float(loss), ref_loss)
        d0, d1 = ops.l1loss_backward(p, dev(b0), dev(b1) if b1 is not None else None, lw, ws)
        close(d0.cpu().numpy(), gold[f"l1_{i}_d0"], 2e-6)
        if d1 is not None:
            close(d1.cpu().numpy(), gold[f"l1_{i}_d1"], 2e-6)
    if "stock_stem" in gold:
        from flownet2_amd import functional as Fn
        x, w, b = MG.stock_inputs("stem")
        close(ops.conv_k7s2_relu_forward(dev(x), dev(w), dev(b), 0.1).cpu().numpy(), gold["stock_stem"], 3e-6)
        x, w, b = MG.stock_inputs("predict_flow")
        close(ops.predict_flow_conv_forward(dev(x), dev(w), dev(b)).cpu().numpy(), gold["stock_predict_flow"], 3e-6)
        x, w, b = MG.stock_inputs("upsample_flow")
        close(ops.upsample_flow_deconv_forward(dev(x), dev(w), dev(b)).cpu().numpy(), gold["stock_upsample_flow"], 2e-6)
        x, w, b = MG.stock_inputs("deconv")
        # 12 -> 8 / 12 -> 16 channels on 5x7 / 9x11 maps: outside every own kernel family (whole channel quads, 16-channel groups): the
        # layers' last resort, the library convolution + our bias / ReLU pass
        from flownet2_amd import nets
        with torch.no_grad():
            close(nets.deconv_forward(dev(x), dev(w), dev(b), True, Fn).cpu().numpy(), gold["stock_deconv_relu"], 1e-5)
            x, w, b = MG.stock_inputs("conv3x3")
            close(nets.conv_forward(dev(x), dev(w), dev(b), 2, 1, True, Fn).cpu().numpy(), gold["stock_conv3x3s2_relu"], 1e-5)
        close(ops.bias_leaky_relu_(dev(gold["stock_conv3x3s2_nobias"]), dev(b), 0.1).cpu().numpy(), gold["stock_conv3x3s2_relu"], 1e-6)
