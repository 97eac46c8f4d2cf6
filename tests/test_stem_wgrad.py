"""Weight gradient of the 7x7 / 2 / 3 stem convolution (csrc/conv_stem_wgrad.hip, fn2_conv_k7s2_wgrad <- ConvolutionLayer::Backward_gpu ->
weight_gpu_gemm, conv_layer.cu:40-52, base_conv_layer.cpp:368-384): the oracle twin against torch's fp64 gradient on the CPU; the HIP
kernel against the twin BIT FOR BIT (3 / 6 / 12 bottom channels, ragged heights and last x segments, several part counts) and against
fp64 at the FlowNetC training shape (batch 8 = 16 tower samples @448x320) on the GPU; through autograd the stem's weight gradient no
longer goes to the library."""
import numpy as np
import pytest
import torch

import oracle
from flownet2_amd import ops


def rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def wgrad64(d, x):
    xt = torch.from_numpy(x).double()
    W = torch.zeros((d.shape[1], x.shape[1], 7, 7), dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(xt, W, stride=2, padding=3) * torch.from_numpy(d).double()).sum().backward()
    return W.grad.numpy()


CASES = [(2, 3, 16, 24), (1, 6, 10, 40), (3, 3, 9, 72), (1, 12, 8, 16), (2, 3, 30, 136), (1, 3, 5, 8)]      # N, Cin, H, W (W % 8 == 0; odd H; W / 2 % 32 != 0)


@pytest.mark.parametrize("case", CASES[:4])
def test_oracle_twin_matches_fp64_autograd(case):
    N, Cin, H, W = case
    x, d = rand((N, Cin, H, W), 1), rand((N, 64, (H - 1) // 2 + 1, W // 2), 2)
    ref = wgrad64(d, x)
    units = N * (((H - 1) // 2 + 1 + 1) // 2) * ((W // 2 + 31) // 32)
    for parts in (1, 3):
        if parts > units:
            continue
        got = oracle.conv_k7s2_wgrad(d, x, parts)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) * np.sqrt(d[0, 0].size * N)


def test_library_reports_support_and_parts():
    assert ops.conv_k7s2_wgrad_supported(16, 3, 320, 448, 64) and ops.conv_k7s2_wgrad_supported(4, 12, 384, 768, 64)
    assert not ops.conv_k7s2_wgrad_supported(16, 4, 320, 448, 64) and not ops.conv_k7s2_wgrad_supported(16, 3, 320, 452, 64)
    assert not ops.conv_k7s2_wgrad_supported(16, 3, 320, 448, 128)
    assert ops.conv_k7s2_wgrad_ksplit(16, 3, 320, 448, 64) == 768 and ops.conv_k7s2_wgrad_ksplit(1, 3, 16, 24, 64) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_kernel_is_bit_identical_to_the_twin(case):
    N, Cin, H, W = case
    x, d = rand((N, Cin, H, W), 11), rand((N, 64, (H - 1) // 2 + 1, W // 2), 12)
    parts = ops.conv_k7s2_wgrad_ksplit(N, Cin, H, W, 64)
    want = oracle.conv_k7s2_wgrad(d, x, parts)
    got = ops.conv_k7s2_wgrad(torch.from_numpy(d).cuda(), torch.from_numpy(x).cuda())
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), float(np.abs(got.cpu().numpy() - want).max())
    assert np.abs(want - wgrad64(d, x)).max() <= 2e-6 * max(1.0, np.abs(want).max()) * np.sqrt(d[0, 0].size * N)


@pytest.mark.gpu
def test_kernel_at_the_flownetc_training_shape_vs_fp64_and_through_autograd():
    from flownet2_amd import functional as Fn
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((16, 3, 320, 448), device="cuda", generator=g)
    d = torch.randn((16, 64, 160, 224), device="cuda", generator=g)
    got = ops.conv_k7s2_wgrad(d, x)
    ref = torch.ops.aten.convolution_backward(d.double(), x.double(), torch.zeros((64, 3, 7, 7), dtype=torch.float64, device="cuda"), None,
                                              [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    err, scale = float((got.double() - ref).abs().max()), float(ref.abs().max())
    assert err <= 1e-5 * scale, (err, scale)
    assert torch.equal(got, ops.conv_k7s2_wgrad(d, x)), "weight gradient must be bit-reproducible"
    w = torch.randn((64, 3, 7, 7), device="cuda", generator=g) * 0.05
    assert torch.equal(Fn._own_bwd_weight(d, x, w, 2, 3, False), got)
