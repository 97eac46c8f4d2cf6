"""CPU tests: the C oracle against an independent fp64 derivation (tests/ref_torch64.py), numeric
gradients, and domain invariants (SURVEY.md section 4 'what we adopt').  No GPU needed."""
import numpy as np
import pytest
import torch

import oracle
import ref_torch64 as R

CORR_CASES = [
    # (N, C, H, W, pad, K, md, s1, s2)
    (2, 5, 9, 11, 4, 1, 4, 1, 2),      # FlowNetC-like, small
    (1, 7, 8, 10, 3, 3, 2, 2, 1),      # kernel 3, stride1 2
    (2, 3, 7, 9, 3, 1, 3, 1, 1),
    (1, 4, 10, 9, 5, 3, 4, 1, 2),      # pad > md + kr
    (1, 33, 6, 7, 2, 1, 2, 1, 1),      # C > 32: exercises the 32-lane partial sums
    (1, 2, 6, 6, 6, 1, 4, 1, 2),       # pad > md -> top larger than bottom
]


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("case", CORR_CASES)
@pytest.mark.parametrize("ctype", [oracle.MULTIPLY, oracle.SUBTRACT])
def test_correlation_forward_vs_fp64(case, ctype):
    N, C, H, W, pad, K, md, s1, s2 = case
    b0, b1 = _rand((N, C, H, W), 1), _rand((N, C, H, W), 2)
    p = oracle.corr_params(pad, K, md, s1, s2, ctype)
    top = oracle.correlation_forward(p, b0, b1)
    ref = R.correlation(torch.from_numpy(b0).double(), torch.from_numpy(b1).double(), pad, K, md, s1, s2,
                        subtract=(ctype == oracle.SUBTRACT)).numpy()
    assert top.shape == ref.shape
    np.testing.assert_allclose(top, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("case", CORR_CASES)
def test_correlation_backward_is_gradient(case):
    N, C, H, W, pad, K, md, s1, s2 = case
    b0, b1 = _rand((N, C, H, W), 3), _rand((N, C, H, W), 4)
    p = oracle.corr_params(pad, K, md, s1, s2)
    tc, th, tw = oracle.correlation_out_shape(p, C, H, W)
    g = _rand((N, tc, th, tw), 5)
    d0, d1 = oracle.correlation_backward(p, b0, b1, g)
    t0 = torch.from_numpy(b0).double().requires_grad_()
    t1 = torch.from_numpy(b1).double().requires_grad_()
    R.correlation(t0, t1, pad, K, md, s1, s2).backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(d0, t0.grad.numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(d1, t1.grad.numpy(), rtol=0, atol=5e-6)


def test_correlation_subtract_backward_matches_reference_formula():
    """The reference's SUBTRACT backward is NOT the true gradient: both kernels take the sign from
    bottom0 and bottom1 at the SAME shifted position (correlation_layer.cu:337-341, :404-408)
    instead of sign(b0[m,l] - b1[m+s2p,l+s2o]).  The oracle restates that quirk; check it against an
    independent loop restatement of the formula in SURVEY.md Appendix A.2."""
    N, C, H, W, pad, K, md, s1, s2 = 1, 3, 7, 8, 3, 3, 2, 2, 1
    b0, b1 = _rand((N, C, H, W), 6), _rand((N, C, H, W), 7)
    p = oracle.corr_params(pad, K, md, s1, s2, oracle.SUBTRACT)
    tc, th, tw = oracle.correlation_out_shape(p, C, H, W)
    g = _rand((N, tc, th, tw), 8).astype(np.float64)
    d0, d1 = oracle.correlation_backward(p, b0, b1, g.astype(np.float32))
    kr, ngr = (K - 1) // 2, md // s2
    ngw = 2 * ngr + 1
    P0 = np.pad(b0, ((0, 0), (0, 0), (pad, pad), (pad, pad))).astype(np.float64)
    P1 = np.pad(b1, ((0, 0), (0, 0), (pad, pad), (pad, pad))).astype(np.float64)
    e0, e1 = np.zeros(b0.shape), np.zeros(b0.shape)
    import math
    for c in range(C):
        for y in range(H):
            for x in range(W):
                l, m = x + pad, y + pad
                for q in range(-ngr, ngr + 1):
                    for o in range(-ngr, ngr + 1):
                        ch = (q + ngr) * ngw + (o + ngr)
                        # bottom0: window of tops whose patch covers (m,l)
                        ys = [t for t in range(th) if t * s1 + md <= m <= t * s1 + md + 2 * kr]
                        xs = [t for t in range(tw) if t * s1 + md <= l <= t * s1 + md + 2 * kr]
                        sgn = 1.0 if P0[0, c, m + q * s2, l + o * s2] >= P1[0, c, m + q * s2, l + o * s2] else -1.0
                        e0[0, c, y, x] += sgn * sum(g[0, ch, yy, xx] for yy in ys for xx in xs)
                        ys = [t for t in range(th) if t * s1 + md <= m - q * s2 <= t * s1 + md + 2 * kr]
                        xs = [t for t in range(tw) if t * s1 + md <= l - o * s2 <= t * s1 + md + 2 * kr]
                        if ys and xs:
                            sgn = -1.0 if P0[0, c, m - q * s2, l - o * s2] >= P1[0, c, m - q * s2, l - o * s2] else 1.0
                            e1[0, c, y, x] += sgn * sum(g[0, ch, yy, xx] for yy in ys for xx in xs)
    np.testing.assert_allclose(d0, e0 / (K * K * C), rtol=0, atol=5e-6)
    np.testing.assert_allclose(d1, e1 / (K * K * C), rtol=0, atol=5e-6)


def test_correlation_self_peak_and_shape_errors():
    b = _rand((1, 8, 12, 12), 9)
    p = oracle.corr_params(4, 1, 4, 1, 2)
    top = oracle.correlation_forward(p, b, b)
    ngw = 5
    centre = (ngw // 2) * ngw + ngw // 2
    np.testing.assert_allclose(top[:, centre], (b * b).mean(1), atol=1e-6)   # zero displacement = mean square
    with pytest.raises(ValueError):
        oracle.correlation_out_shape(oracle.corr_params(4, 2, 4, 1, 1), 3, 8, 8)     # even kernel
    with pytest.raises(ValueError):
        oracle.correlation_out_shape(oracle.corr_params(1, 1, 4, 1, 1), 3, 8, 8)     # pad < md
    with pytest.raises(ValueError):
        oracle.correlation_out_shape(oracle.corr_params(0, 1, 0, 0, 1), 3, 8, 8)     # stride 0


CORR1D_CASES = [
    # (N, C, H, W, pad, K, md, s1, s2, single_direction)
    (2, 5, 7, 19, 4, 1, 4, 1, 2, 0),
    (1, 7, 9, 14, 3, 3, 2, 2, 1, 1),
    (1, 16, 5, 24, 10, 1, 10, 1, 1, -1),   # DispNetCorr1D-like: left only
    (1, 33, 4, 12, 6, 1, 6, 1, 2, -1),     # left, stride_2 2: overshoot of 2 columns lands in the zero padding
    (1, 4, 8, 15, 5, 3, 4, 1, 2, 0),
    (2, 3, 6, 9, 0, 1, 2, 1, 1, 0),        # no padding at all: top narrower than bottom
]


@pytest.mark.parametrize("case", CORR1D_CASES)
@pytest.mark.parametrize("ctype", [oracle.MULTIPLY, oracle.SUBTRACT])
def test_correlation1d_forward_vs_fp64(case, ctype):
    N, C, H, W, pad, K, md, s1, s2, sd = case
    b0, b1 = _rand((N, C, H, W), 21), _rand((N, C, H, W), 22)
    p = oracle.corr_params(pad, K, md, s1, s2, ctype, 0, sd)
    top = oracle.correlation1d_forward(p, b0, b1)
    ref = R.correlation1d(torch.from_numpy(b0).double(), torch.from_numpy(b1).double(), pad, K, md, s1, s2, sd,
                          subtract=(ctype == oracle.SUBTRACT)).numpy()
    assert top.shape == ref.shape == (N,) + oracle.correlation1d_out_shape(p, C, H, W)
    np.testing.assert_allclose(top, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("case", CORR1D_CASES)
def test_correlation1d_backward_is_gradient(case):
    N, C, H, W, pad, K, md, s1, s2, sd = case
    b0, b1 = _rand((N, C, H, W), 23), _rand((N, C, H, W), 24)
    p = oracle.corr_params(pad, K, md, s1, s2, oracle.MULTIPLY, 0, sd)
    g = _rand((N,) + oracle.correlation1d_out_shape(p, C, H, W), 25)
    d0, d1 = oracle.correlation1d_backward(p, b0, b1, g)
    t0 = torch.from_numpy(b0).double().requires_grad_()
    t1 = torch.from_numpy(b1).double().requires_grad_()
    R.correlation1d(t0, t1, pad, K, md, s1, s2, sd).backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(d0, t0.grad.numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(d1, t1.grad.numpy(), rtol=0, atol=5e-6)


def _flat_padded(b, pad):
    """The reference's scratch blob [N, H, W+2p, C] as one flat array of pixels (correlation_layer1d.cu:25-44)."""
    return np.pad(b, ((0, 0), (0, 0), (0, 0), (pad, pad))).transpose(0, 2, 3, 1).reshape(-1, b.shape[1]).astype(np.float64)


def test_correlation1d_left_mode_wraps_into_the_previous_row_like_the_flat_blob():
    """single_direction = -1 with pad < overshoot: the reference's flat index lands on DATA of the previous row
    (correlation_layer1d.cu:89 with x_shift = -grid_width, :467-468); positions in front of the blob read as 0."""
    N, C, H, W, pad, K, md, s1, s2 = 2, 3, 5, 11, 0, 1, 3, 1, 2
    b0, b1 = _rand((N, C, H, W), 26), _rand((N, C, H, W), 27)
    p = oracle.corr_params(pad, K, md, s1, s2, oracle.MULTIPLY, 0, -1)
    tc, th, tw = oracle.correlation1d_out_shape(p, C, H, W)
    assert (tc, th, tw) == (2, 5, 5)
    top = oracle.correlation1d_forward(p, b0, b1)
    pW = W + 2 * pad
    F0, F1 = _flat_padded(b0, pad), _flat_padded(b1, pad)
    exp = np.zeros((N, tc, th, tw))
    wrapped = 0
    for n in range(N):
        for c in range(tc):
            for y in range(th):
                for x in range(tw):
                    fa = (n * H + y) * pW + x + md
                    fb = fa + (c - tc) * s2
                    wrapped += (x + md + (c - tc) * s2) < 0
                    exp[n, c, y, x] = (F0[fa] * (F1[fb] if fb >= 0 else 0.0)).sum() / C
    assert wrapped > 0
    np.testing.assert_allclose(top, exp, rtol=0, atol=2e-6)
    assert np.abs(top[1, 0, 0, 0]) > 0          # first row of sample 1 reads the last row of sample 0
    # backward of bottom0 reads bottom1 through the same flat index (:156)
    g = _rand((N, tc, th, tw), 28)
    d0, d1 = oracle.correlation1d_backward(p, b0, b1, g)
    e0 = np.zeros((N, C, H, W))
    for n in range(N):
        for y in range(H):
            for x in range(W):
                l = x + pad
                xt = l - md                       # K = 1, s1 = 1: the one top column whose patch covers l
                if 0 <= xt < tw:
                    for c in range(tc):
                        f = (n * H + y) * pW + l + (c - tc) * s2
                        e0[n, :, y, x] += g[n, c, y, xt] * (F1[f] if f >= 0 else 0.0)
    np.testing.assert_allclose(d0, e0 / C, rtol=0, atol=5e-6)


def test_correlation1d_shape_errors():
    with pytest.raises(ValueError):
        oracle.correlation1d_out_shape(oracle.corr_params(4, 2, 4, 1, 1), 3, 8, 8)                 # even kernel
    with pytest.raises(ValueError):
        oracle.correlation1d_out_shape(oracle.corr_params(4, 1, 4, 1, 1, 0, 0, 2), 3, 8, 8)        # single_direction out of range
    with pytest.raises(ValueError):
        oracle.correlation1d_out_shape(oracle.corr_params(0, 1, 4, 1, 1), 3, 8, 8)                 # neighbourhood does not fit
    assert oracle.correlation1d_out_shape(oracle.corr_params(40, 1, 40, 1, 1, 0, 0, -1), 8, 6, 30) == (41, 6, 30)
    assert oracle.correlation1d_out_shape(oracle.corr_params(40, 1, 40, 1, 1), 8, 6, 30) == (81, 6, 30)


def test_flow_warp_forward_vs_fp64_and_identity():
    N, C, H, W = 2, 3, 13, 17
    img = _rand((N, C, H, W), 10)
    flow = (_rand((N, 2, H, W), 11) * 3).astype(np.float32)
    out = oracle.flow_warp_forward(img, flow)
    ref = R.flow_warp(torch.from_numpy(img).double(), torch.from_numpy(flow).double()).numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5)
    zero = np.zeros_like(flow)
    np.testing.assert_array_equal(oracle.flow_warp_forward(img, zero), img)
    nanout = oracle.flow_warp_forward(img, flow + 100, oracle.FILL_NAN)
    assert np.isnan(nanout).all()
    assert (oracle.flow_warp_forward(img, flow + 100) == 0).all()


def test_flow_warp_backward_is_gradient_inside():
    N, C, H, W = 1, 2, 9, 10
    rng = np.random.default_rng(12)
    img = _rand((N, C, H, W), 13)
    flow = rng.uniform(-2.5, 2.5, (N, 2, H, W)).astype(np.float32)
    g = _rand((N, C, H, W), 14)
    di, df = oracle.flow_warp_backward(img, flow, g)
    ti = torch.from_numpy(img).double().requires_grad_()
    tf = torch.from_numpy(flow).double().requires_grad_()
    R.flow_warp(ti, tf).backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(di, ti.grad.numpy(), rtol=0, atol=1e-5)
    # flow gradient: equal wherever the sample does not land in the clamped last row/column
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    x2, y2 = xs + flow[:, 0], ys + flow[:, 1]
    interior = (x2 >= 0) & (y2 >= 0) & (x2 < W - 1) & (y2 < H - 1)
    np.testing.assert_allclose(df[:, 0][interior], tf.grad.numpy()[:, 0][interior], rtol=0, atol=1e-5)
    np.testing.assert_allclose(df[:, 1][interior], tf.grad.numpy()[:, 1][interior], rtol=0, atol=1e-5)
    oob = ~((x2 >= 0) & (y2 >= 0) & (x2 < W) & (y2 < H))
    assert (df[:, 0][oob] == 0).all() and (df[:, 1][oob] == 0).all()
    di2, df2 = oracle.flow_warp_backward(img, flow, g, propagate_image=False)
    assert (di2 == 0).all() and np.array_equal(df2, df)


@pytest.mark.parametrize("kind,code", [("linear", oracle.LINEAR), ("cubic", oracle.CUBIC), ("nearest", oracle.NEAREST)])
@pytest.mark.parametrize("shape", [((6, 8), (24, 32)), ((16, 20), (8, 10)), ((9, 12), (9, 12)), ((12, 16), (7, 9))])
def test_resample_vs_fp64(kind, code, shape):
    (Hin, Win), (Hout, Wout) = shape
    x = _rand((2, 2, Hin, Win), 15)
    out = oracle.resample_forward(x, Hout, Wout, code, True)
    ref = R.resample(x.astype(np.float64), Hout, Wout, kind, True)
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
    if (Hin, Win) == (Hout, Wout):
        np.testing.assert_array_equal(out, x)


def test_resample_ramp_and_constant():
    ramp = np.tile(np.arange(8, dtype=np.float32), (1, 1, 4, 1))
    up = oracle.resample_forward(ramp, 16, 32, oracle.LINEAR)
    np.testing.assert_allclose(up[0, 0, 0, :8], [0, 0, .125, .375, .625, .875, 1.125, 1.375], atol=1e-6)
    const = np.full((1, 1, 5, 7), 3.25, np.float32)
    np.testing.assert_allclose(oracle.resample_forward(const, 20, 28, oracle.LINEAR), 3.25, atol=1e-6)
    with pytest.raises(ValueError):
        oracle.resample_forward(const, 3, 3, oracle.AREA)


L1_CASES = [
    dict(l2_per_location=True, normalize_by_num_entries=True),
    dict(l2_per_location=True, l2_prescale_by_channels=True, plateau=0.3),
    dict(l2_per_location=False),
    dict(l2_per_location=False, plateau=0.2, normalize_by_num_entries=True),
    dict(l2_per_location=True, epsilon=1e-3),
]


@pytest.mark.parametrize("kw", L1_CASES)
@pytest.mark.parametrize("two", [True, False])
def test_l1loss_vs_fp64_with_nans(kw, two):
    N, C, H, W = 3, 2, 7, 9
    b0, b1 = _rand((N, C, H, W), 16), _rand((N, C, H, W), 17)
    tgt = b1 if two else b0
    tgt[0, :, 2, 3] = np.nan
    tgt[1, 0, 4, 4] = np.nan
    p = oracle.l1_params(**kw)
    loss, norm = oracle.l1loss_forward(p, b0, b1 if two else None)
    t0 = torch.from_numpy(b0).double().requires_grad_()
    t1 = torch.from_numpy(b1).double().requires_grad_() if two else None
    rl, rn = R.l1loss(t0, t1, kw.get("l2_per_location", False), kw.get("l2_prescale_by_channels", False),
                      kw.get("normalize_by_num_entries", False), kw.get("epsilon", 1e-2), kw.get("plateau", 0.0))
    assert abs(norm - float(rn)) < 1e-6
    assert abs(loss - float(rl)) < 1e-5 * max(1, abs(float(rl)))
    d0, d1 = oracle.l1loss_backward(p, b0, b1 if two else None, 0.7, norm)
    (rl * 0.7).backward()
    g0 = np.nan_to_num(t0.grad.numpy())
    np.testing.assert_allclose(d0, g0, rtol=0, atol=2e-6)
    if two:
        np.testing.assert_allclose(d1, np.nan_to_num(t1.grad.numpy()), rtol=0, atol=2e-6)
    assert not np.isnan(d0).any()


def test_l1loss_masked_pixels_still_pay_sqrt_eps():
    # fully-NaN pixels add sqrt(eps) each while norm counts valid entries only.
    b0 = np.zeros((1, 2, 2, 2), np.float32)
    b1 = np.zeros((1, 2, 2, 2), np.float32)
    b1[0, :, 0, 0] = np.nan
    loss, norm = oracle.l1loss_forward(oracle.l1_params(l2_per_location=True, normalize_by_num_entries=True), b0, b1)
    assert norm == 3.0
    assert abs(loss - 4 * 0.1 / 3) < 1e-6


def test_channel_norm():
    x = _rand((2, 3, 5, 6), 18)
    top = oracle.channel_norm_forward(x)
    t = torch.from_numpy(x).double().requires_grad_()
    r = R.channel_norm(t)
    np.testing.assert_allclose(top, r.detach().numpy(), atol=1e-6)
    g = _rand((2, 1, 5, 6), 19)
    r.backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(oracle.channel_norm_backward(x, top, g), t.grad.numpy(), atol=1e-5)


@pytest.mark.parametrize("shape", [((16, 24), (4, 6)), ((17, 23), (5, 7)), ((8, 8), (8, 8))])
def test_downsample_vs_fp64_with_nan_voting(shape):
    (Hin, Win), (Hout, Wout) = shape
    x = _rand((1, 2, Hin, Win), 20)
    x[0, 0, :6, :9] = np.nan
    out = oracle.downsample_forward(x, Hout, Wout)
    ref = R.downsample(x, Hout, Wout)
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(out), np.nan_to_num(ref), atol=2e-6)


def test_downsample_multi_twin_is_the_layers_one_after_another():
    x = _rand((2, 2, 33, 47), 21)
    x[1, 1, 5:20, 7:30] = np.nan
    sizes = [(16, 23), (8, 11), (4, 5), (2, 2)]
    for (h, w), got in zip(sizes, oracle.downsample_forward_multi(x, sizes)):
        np.testing.assert_array_equal(got.view(np.uint32), oracle.downsample_forward(x, h, w).view(np.uint32))
    with pytest.raises(Exception):
        oracle.downsample_forward_multi(x, [(1, 4)])
    with pytest.raises(Exception):
        oracle.downsample_forward_multi(x, [(33, 47)])


def test_bias_leaky_relu_oracle_matches_torch():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 6, 5, 7)).astype(np.float32); b = rng.standard_normal(6).astype(np.float32)
    ref = torch.nn.functional.leaky_relu(torch.from_numpy(x) + torch.from_numpy(b).view(1, -1, 1, 1), 0.1).numpy()
    np.testing.assert_allclose(oracle.bias_leaky_relu_forward(x, b, 0.1), ref, rtol=0, atol=1e-7)
    np.testing.assert_allclose(oracle.bias_leaky_relu_forward(x, None, 0.1), torch.nn.functional.leaky_relu(torch.from_numpy(x), 0.1).numpy(), rtol=0, atol=1e-7)


def test_stem_conv_oracle_matches_torch():
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 3, 13, 18)).astype(np.float32); w = (0.1 * rng.standard_normal((4, 3, 7, 7))).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                                                                    torch.from_numpy(b).double(), stride=2, padding=3), 0.1).numpy()
    got = oracle.conv_k7s2_relu_forward(x, w, b, 0.1)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_im2col_col2im_oracle():
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 3, 6, 7)).astype(np.float32)
    for k, p, s in ((3, 1, 1), (3, 1, 2), (4, 1, 2), (5, 2, 1)):
        np.testing.assert_array_equal(oracle.im2col_forward(x, k, p, s), torch.nn.functional.unfold(torch.from_numpy(x), k, padding=p, stride=s).numpy())
    # deconvolution 4/2/1 through GEMM + col2im against conv_transpose2d
    w = (0.1 * rng.standard_normal((3, 5, 4, 4))).astype(np.float32); b = rng.standard_normal(5).astype(np.float32)
    col = np.matmul(w.reshape(3, 80).T.astype(np.float64), x.reshape(2, 3, 42).astype(np.float64)).astype(np.float32)
    got = oracle.col2im_bias_relu_forward(col, b, 2, 5, 12, 14, 4, 1, 2, relu=False)
    ref = torch.nn.functional.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=1)
    np.testing.assert_allclose(got, ref.numpy(), rtol=0, atol=3e-6)
    # col2im is the adjoint-shaped fold: against torch.nn.functional.fold
    colr = rng.standard_normal((2, 5 * 16, 42)).astype(np.float32)
    fold = torch.nn.functional.fold(torch.from_numpy(colr).double(), (12, 14), 4, padding=1, stride=2).numpy()
    np.testing.assert_allclose(oracle.col2im_bias_relu_forward(colr, None, 2, 5, 12, 14, 4, 1, 2, relu=False), fold, rtol=0, atol=2e-6)


def test_l1loss_multi_is_the_layers_one_by_one_plus_the_net_sum():
    """oracle.l1loss_forward_multi / _backward_multi (the twins of fn2_l1loss_*_multi): per layer the single-layer oracle, total = the
    float sum of loss_weight * loss in layer order (net.cpp:565-579)."""
    rng = np.random.default_rng(7)
    shapes = [(2, 2, 20, 28), (2, 2, 10, 14), (2, 2, 5, 7)]
    preds = [rng.standard_normal(sh).astype(np.float32) for sh in shapes]
    gts = [rng.standard_normal(sh).astype(np.float32) for sh in shapes]
    gts[1][0, :, 2, 2] = np.nan
    w = [0.005, 0.01, 0.32]
    p = oracle.l1_params(l2_per_location=True, normalize_by_num_entries=True)
    total, losses, norms = oracle.l1loss_forward_multi(p, preds, gts, w)
    want = np.float32(0)
    for k in range(3):
        l, nrm = oracle.l1loss_forward(p, preds[k], gts[k])
        assert losses[k] == np.float32(l) and norms[k] == np.float32(nrm)
        want = np.float32(want + np.float32(np.float32(w[k]) * np.float32(l)))
    assert np.float32(total) == want
    d0s, d1s = oracle.l1loss_backward_multi(p, preds, gts, w, 0.5, norms)
    for k in range(3):
        r0, r1 = oracle.l1loss_backward(p, preds[k], gts[k], float(np.float32(np.float32(w[k]) * np.float32(0.5))), float(norms[k]))
        assert np.array_equal(d0s[k], r0) and np.array_equal(d1s[k], r1)


VIEWS = [  # blob shape [A][B][k][k], operand Cout, Cin, src_cout, src_cin, stride_cout, stride_cin, flip, the torch expression it equals
    ((64, 24, 3, 3), 64, 24, 64, 24, 24 * 9, 9, False, lambda w: w),
    ((40, 64, 5, 5), 64, 40, 64, 40, 25, 64 * 25, False, lambda w: w.transpose(0, 1)),
    ((70, 12, 4, 4), 128, 12, 70, 12, 12 * 16, 16, False, lambda w: torch.cat([w, w.new_zeros((58, 12, 4, 4))], 0)),
    ((32, 20, 1, 1), 32, 32, 20, 32, 1, 20, False, lambda w: torch.cat([w.transpose(0, 1), w.new_zeros((12, 32, 1, 1))], 0)),
    ((16, 50, 3, 3), 64, 16, 50, 16, 9, 50 * 9, True, lambda w: torch.cat([w.flip(2, 3).transpose(0, 1), w.new_zeros((14, 16, 3, 3))], 0)),
    ((10, 8, 4, 4), 128, 10, 128, 10, 1, 128, False, lambda w: w.reshape(10, 128).t().reshape(128, 10, 1, 1))]


@pytest.mark.parametrize("view", VIEWS)
def test_packed_operand_of_a_strided_view_equals_packing_the_materialised_tensor(view):
    """oracle.conv_mfma_pack_weights_view (twin of fn2_conv_mfma_pack_weights_view): transposed / rotated / zero-padded operands and the GEMM
    operand of a Deconvolution straight from the blob == conv_mfma_pack_weights of the tensor torch would have materialised."""
    shape, Cout, Cin, sco, sci, st_co, st_ci, flip, expr = view
    w = np.random.default_rng(5).standard_normal(shape).astype(np.float32)
    k = 1 if (shape[2] == 4 and Cout == 128 and Cin == 10) else shape[2]
    got = oracle.conv_mfma_pack_weights_view(w, Cout, Cin, k, sco, sci, st_co, st_ci, flip)
    want = oracle.conv_mfma_pack_weights(expr(torch.from_numpy(w)).contiguous().numpy())
    assert np.array_equal(got, want)
