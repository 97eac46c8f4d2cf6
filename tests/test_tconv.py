"""Transposed convolution, stride 2 (csrc/tconv_mfma.hip): the Deconvolution forward and the data gradient of stride-2 convolutions.
CPU: the oracle twin against torch's fp64 conv_transpose2d.  GPU: every tile variant against the twin BIT FOR BIT (ragged sizes,
channel slices, bias + ReLU), fp64 at the real FlowNetC shapes, and the data-gradient use against autograd."""
import numpy as np
import pytest
import torch

import oracle
from flownet2_amd import ops


def rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def torch_tconv64(x, w, b, k, p, out_hw):
    y = torch.nn.functional.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None if b is None else torch.from_numpy(b).double(),
                                             stride=2, padding=p, output_padding=(out_hw[0] - (2 * (x.shape[2] - 1) + k - 2 * p), out_hw[1] - (2 * (x.shape[3] - 1) + k - 2 * p)))
    return y.numpy()


# (N, Cin, Hin, Win, Cout, kernel, pad, extra output rows / columns)
CASES = [(2, 13, 5, 8, 64, 4, 1, 0), (1, 20, 7, 12, 128, 5, 2, 1), (2, 9, 6, 16, 64, 3, 1, 1), (1, 64, 3, 4, 64, 4, 1, 0),
         (1, 33, 9, 20, 64, 5, 2, 0), (3, 8, 4, 8, 192, 3, 1, 0)]


def out_size(case):
    N, Cin, H, W, Cout, k, p, e = case
    return 2 * (H - 1) + k - 2 * p + e, 2 * (W - 1) + k - 2 * p + e


@pytest.mark.parametrize("case", CASES)
def test_oracle_twin_matches_fp64_conv_transpose(case):
    N, Cin, H, W, Cout, k, p, e = case
    x, w, b = rand((N, Cin, H, W), 1), rand((Cin, Cout, k, k), 2, 0.2), rand((Cout,), 3)
    hw = out_size(case)
    got = oracle.tconv_forward(x, w, b, k, p, out_hw=hw)
    ref = torch_tconv64(x, w, b, k, p, hw)
    assert np.abs(got - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max())
    got = oracle.tconv_forward(x, w, b, k, p, out_hw=hw, relu=True, negative_slope=0.1)
    np.testing.assert_allclose(got, np.where(ref > 0, ref, 0.1 * ref), atol=3e-6 * max(1.0, np.abs(ref).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_every_variant_is_bit_identical_to_the_twin(case):
    N, Cin, H, W, Cout, k, p, e = case
    x, w, b = rand((N, Cin, H, W), 11), rand((Cin, Cout, k, k), 12, 0.2), rand((Cout,), 13)
    hw = out_size(case)
    assert ops.tconv_supported(Cin, H, W, Cout, hw[0], hw[1], k, p)
    ref = oracle.tconv_forward(x, w, b, k, p, out_hw=hw, relu=True, negative_slope=0.1)
    xd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(b).cuda()
    pw = ops.tconv_pack_weights(torch.from_numpy(w).cuda())
    got = ops.tconv_forward(xd, pw, bd, Cout, k, p, out_hw=hw, relu=True, negative_slope=0.1).cpu().numpy()        # first-use selection
    np.testing.assert_array_equal(got, ref)
    ran = 0
    try:
        for v in range(ops.tconv_num_variants()):
            ops.set_tconv_variant(v)
            try:
                got = ops.tconv_forward(xd, pw, bd, Cout, k, p, out_hw=hw, relu=True, negative_slope=0.1).cpu().numpy()
            except Exception as ex:      # variants of other tap classes / channel blockings are refused
                assert "does not apply" in str(ex)
                continue
            np.testing.assert_array_equal(got, ref, err_msg=f"variant {v}")
            ran += 1
    finally:
        ops.set_tconv_variant(-1)
    assert ran >= 6


@pytest.mark.gpu
def test_channel_slices_and_no_bias():
    N, Cin, H, W, Cout, k, p = 2, 24, 6, 12, 64, 4, 1
    blob = rand((N, Cin + 9, H, W), 21)
    w = rand((Cin, Cout, k, k), 22, 0.2)
    top = np.full((N, Cout + 5, 2 * H, 2 * W), 7.0, np.float32)
    ref = oracle.tconv_forward(blob, w, None, k, p, out=top.copy(), out_c0=3, in_c0=4, Cin=Cin)
    out = torch.from_numpy(top.copy()).cuda()
    ops.tconv_forward(torch.from_numpy(blob).cuda(), ops.tconv_pack_weights(torch.from_numpy(w).cuda()), None, Cout, k, p, out=out, out_c0=3, in_c0=4, Cin=Cin)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    assert (ref[:, :3] == 7.0).all() and (ref[:, 3 + Cout:] == 7.0).all()


# FlowNetC shapes (batch 8 @448x320): (x shape, Cout, kernel, pad, output size)
LAYERS = {"deconv2 fwd": ((8, 386, 40, 56), 64, 4, 1, (80, 112)), "deconv3 fwd": ((8, 770, 20, 28), 128, 4, 1, (40, 56)),
          "conv2 dgrad": ((16, 128, 80, 112), 64, 5, 2, (160, 224)), "conv3 dgrad": ((16, 256, 40, 56), 128, 5, 2, (80, 112)),
          "conv4 dgrad": ((8, 512, 20, 28), 256, 3, 1, (40, 56))}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LAYERS))
def test_flownetc_shapes_vs_fp64(name):
    sx, Cout, k, p, hw = LAYERS[name]
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(sx, device="cuda", generator=g)
    w = torch.randn((sx[1], Cout, k, k), device="cuda", generator=g) * 0.05
    got = ops.tconv_forward(x, ops.tconv_pack_weights(w), None, Cout, k, p, out_hw=hw)
    opad = (hw[0] - (2 * (sx[2] - 1) + k - 2 * p), hw[1] - (2 * (sx[3] - 1) + k - 2 * p))
    ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=p, output_padding=opad)
    err, scale = float((got.double() - ref).abs().max()), float(ref.abs().max())
    assert err <= 1e-5 * scale, (name, err, scale)


@pytest.mark.gpu
def test_data_gradient_of_a_stride2_convolution_matches_autograd():
    for (Cin, Cout, k, p, H, W) in [(64, 128, 5, 2, 24, 40), (128, 64, 3, 1, 17, 28)]:
        g = torch.Generator(device="cuda").manual_seed(3)
        x = torch.randn((2, Cin, H, W), device="cuda", generator=g, dtype=torch.float64).requires_grad_(True)
        wgt = torch.randn((Cout, Cin, k, k), device="cuda", generator=g, dtype=torch.float64) * 0.1
        y = torch.nn.functional.conv2d(x, wgt, stride=2, padding=p)
        d = torch.randn(y.shape, device="cuda", generator=g, dtype=torch.float64)
        (y * d).sum().backward()
        if y.shape[3] % 4:
            continue
        # bottom_diff = tconv(top_diff, W read as [in = Cout_conv][out = Cin_conv][k][k]) at the bottom's size
        got = ops.tconv_forward(d.float(), ops.tconv_pack_weights(wgt.float()), None, Cin, k, p, out_hw=(H, W))
        assert float((got.double() - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max())


@pytest.mark.gpu
def test_training_step_data_gradients_run_on_the_own_kernels():
    """functional._own_bwd_data: stride-2 convolutions through the transposed-convolution kernel, Deconvolution{4,2,1} through the
    4x4 / 2 direct convolution, both against torch's fp64 autograd gradient (conv_layer.cu:53-57, deconv_layer.cu:52-56)."""
    from flownet2_amd import functional as Fn
    g = torch.Generator(device="cuda").manual_seed(9)
    for (Cin, Cout, k, p, H, W) in [(64, 128, 5, 2, 32, 48), (128, 64, 3, 1, 24, 40)]:
        x = torch.randn((2, Cin, H, W), device="cuda", generator=g)
        w = torch.randn((Cout, Cin, k, k), device="cuda", generator=g) * 0.1
        xd = x.double().requires_grad_(True)
        y = torch.nn.functional.conv2d(xd, w.double(), stride=2, padding=p)
        d = torch.randn(y.shape, device="cuda", generator=g)
        (y * d.double()).sum().backward()
        got = Fn._own_bwd_data(d, w, 2, p, False, x.shape)
        assert got is not None and float((got.double() - xd.grad).abs().max()) <= 1e-5 * float(xd.grad.abs().max())
    for (Cin, Cout, H, W) in [(70, 64, 12, 16), (128, 128, 10, 14)]:
        x = torch.randn((2, Cin, H, W), device="cuda", generator=g)
        w = torch.randn((Cin, Cout, 4, 4), device="cuda", generator=g) * 0.1
        xd = x.double().requires_grad_(True)
        y = torch.nn.functional.conv_transpose2d(xd, w.double(), stride=2, padding=1)
        d = torch.randn(y.shape, device="cuda", generator=g)
        (y * d.double()).sum().backward()
        got = Fn._own_bwd_data(d, w, 2, 1, True, x.shape)
        assert got is not None and tuple(got.shape) == tuple(x.shape)
        assert float((got.double() - xd.grad).abs().max()) <= 1e-5 * float(xd.grad.abs().max())
