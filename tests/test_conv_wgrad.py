"""Weight gradient of Convolution / Deconvolution layers (csrc/conv_wgrad.hip): the oracle twin against torch's fp64 gradient on the
CPU; the HIP kernel against the twin BIT FOR BIT (every tap class / chunk width, ragged channel counts, channel slices, widths that
are not multiples of 4, accumulation) and against fp64 at the real FlowNetC training shapes on the GPU."""
import numpy as np
import pytest
import torch

import oracle
from flownet2_amd import ops


def rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def torch_wgrad64(a, b, k, s, p):
    """d/dW of sum(conv2d(b, W, stride s, pad p) * a) in fp64 = the sum the kernel computes."""
    bt = torch.from_numpy(b).double()
    W = torch.zeros((a.shape[1], b.shape[1], k, k), dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(bt, W, stride=s, padding=p)
    at = torch.from_numpy(a).double()
    hy, wy = min(y.shape[2], at.shape[2]), min(y.shape[3], at.shape[3])
    assert y.shape[2] >= at.shape[2] and y.shape[3] >= at.shape[3]
    (y[:, :, :hy, :wy] * at).sum().backward()
    return W.grad.numpy()


# (N, Ca, Ha, Wa, Cb, kernel, stride, pad): b is the map a convolution with these parameters would have produced `a` from
CASES = [(2, 20, 6, 8, 9, 3, 1, 1), (1, 70, 5, 7, 33, 3, 2, 1), (2, 64, 4, 12, 32, 5, 2, 2), (1, 16, 7, 28, 40, 4, 2, 1),
         (3, 32, 3, 5, 100, 1, 1, 0), (1, 8, 9, 30, 5, 3, 1, 1), (2, 65, 6, 16, 17, 3, 2, 1), (1, 40, 3, 60, 24, 3, 1, 1),
         (1, 24, 2, 64, 16, 5, 2, 2), (2, 130, 5, 7, 70, 3, 1, 1), (1, 33, 10, 14, 66, 4, 2, 1)]


def shapes_of(case):
    N, Ca, Ha, Wa, Cb, k, s, p = case
    Hb, Wb = s * (Ha - 1) + k - 2 * p, s * (Wa - 1) + k - 2 * p
    return (N, Ca, Ha, Wa), (N, Cb, Hb, Wb)


@pytest.mark.parametrize("case", CASES[:6])
def test_oracle_twin_matches_fp64_autograd(case):
    N, Ca, Ha, Wa, Cb, k, s, p = case
    sa, sb = shapes_of(case)
    a, b = rand(sa, 1), rand(sb, 2)
    ref = torch_wgrad64(a, b, k, s, p)
    for ksplit in (1, 3):
        if ksplit > N * Ha:
            continue
        got = oracle.conv_wgrad(a, b, k, s, p, ksplit)
        assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) * np.sqrt(N * Ha * Wa)
    acc = rand(ref.shape, 3)
    got = oracle.conv_wgrad(a, b, k, s, p, 1, out=acc.copy(), accumulate=True)
    np.testing.assert_array_equal(got, acc + oracle.conv_wgrad(a, b, k, s, p, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_kernel_is_bit_identical_to_the_twin(case):
    N, Ca, Ha, Wa, Cb, k, s, p = case
    sa, sb = shapes_of(case)
    a, b = rand(sa, 11), rand(sb, 12)
    assert ops.conv_wgrad_supported(N, Ca, Ha, Wa, Cb, sb[2], sb[3], k, s, p)
    ks = ops.conv_wgrad_ksplit(N, Ca, Ha, Wa, Cb, sb[2], sb[3], k, s, p)
    got = ops.conv_wgrad(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), k, s, p).cpu().numpy()
    ref = oracle.conv_wgrad(a, b, k, s, p, ks)
    np.testing.assert_array_equal(got, ref)
    again = ops.conv_wgrad(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), k, s, p).cpu().numpy()
    np.testing.assert_array_equal(got, again)


@pytest.mark.gpu
def test_kernel_channel_slices_accumulate_and_short_b():
    # a and b as channel slices of wider blobs, dw accumulated into; b one row / column short of what the taps reach (zeros there)
    N, Ca, Ha, Wa, Cb, k, s, p = 2, 48, 6, 12, 40, 3, 1, 1
    a_blob, b_blob = rand((N, Ca + 7, Ha, Wa), 21), rand((N, Cb + 5, Ha, Wa), 22)
    acc = rand((Ca, Cb, k, k), 23)
    ks = ops.conv_wgrad_ksplit(N, Ca, Ha, Wa, Cb, Ha, Wa, k, s, p)
    out = torch.from_numpy(acc.copy()).cuda()
    ops.conv_wgrad(torch.from_numpy(a_blob).cuda(), torch.from_numpy(b_blob).cuda(), k, s, p, out=out, accumulate=True, a_c0=3, Ca=Ca, b_c0=2, Cb=Cb)
    ref = oracle.conv_wgrad(a_blob, b_blob, k, s, p, ks, out=acc.copy(), accumulate=True, a_c0=3, Ca=Ca, b_c0=2, Cb=Cb)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    a, b = rand((1, 16, 5, 8), 24), rand((1, 16, 9, 15), 25)           # stride 2, kernel 3, pad 1: the taps reach row 9 / column 16
    ks = ops.conv_wgrad_ksplit(1, 16, 5, 8, 16, 9, 15, 3, 2, 1)
    got = ops.conv_wgrad(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 3, 2, 1).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.conv_wgrad(a, b, 3, 2, 1, ks))


# FlowNetC training shapes (batch 8 @448x320; the towers carry 16 samples): (a shape, b shape, k, s, p)
LAYERS = {"conv2": ((16, 128, 80, 112), (16, 64, 160, 224), 5, 2, 2), "conv3": ((16, 256, 40, 56), (16, 128, 80, 112), 5, 2, 2),
          "conv_redir": ((8, 32, 40, 56), (8, 256, 40, 56), 1, 1, 0), "conv3_1": ((8, 256, 40, 56), (8, 473, 40, 56), 3, 1, 1),
          "conv4": ((8, 512, 20, 28), (8, 256, 40, 56), 3, 2, 1), "conv4_1": ((8, 512, 20, 28), (8, 512, 20, 28), 3, 1, 1),
          "conv5": ((8, 512, 10, 14), (8, 512, 20, 28), 3, 2, 1), "conv6_1": ((8, 1024, 5, 7), (8, 1024, 5, 7), 3, 1, 1),
          "deconv5": ((8, 1024, 5, 7), (8, 512, 10, 14), 4, 2, 1), "deconv2": ((8, 386, 40, 56), (8, 64, 80, 112), 4, 2, 1)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LAYERS))
def test_kernel_at_flownetc_training_shapes_vs_fp64(name):
    sa, sb, k, s, p = LAYERS[name]
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(sa, device="cuda", generator=g)
    b = torch.randn(sb, device="cuda", generator=g)
    got = ops.conv_wgrad(a, b, k, s, p)
    if k == 4:      # Deconvolution: a = bottom, b = top_diff; weight [Cin, Cout, 4, 4]
        ref = torch.ops.aten.convolution_backward(b.double(), a.double(), torch.zeros((sa[1], sb[1], k, k), dtype=torch.float64, device="cuda"), None,
                                                  [s, s], [p, p], [1, 1], True, [0, 0], 1, [False, True, False])[1]
    else:
        ref = torch.ops.aten.convolution_backward(a.double(), b.double(), torch.zeros((sa[1], sb[1], k, k), dtype=torch.float64, device="cuda"), None,
                                                  [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    err = float((got.double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert err <= 1e-5 * scale, (name, err, scale)
    assert torch.equal(got, ops.conv_wgrad(a, b, k, s, p)), "weight gradient must be bit-reproducible"
