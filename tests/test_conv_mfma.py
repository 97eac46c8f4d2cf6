"""Direct MFMA convolution (csrc/conv_mfma.hip, fn2_conv_mfma_*): the oracle twin against torch's fp64 convolution (CPU), the HIP
kernels against the oracle BIT FOR BIT in every tile variant, and against the reference's own Convolution + ReLU layers
(oracle/_ref: conv_layer.cu:8-23, base_conv_layer.cpp:326-348, relu_layer.cu:8-27) -- small shapes here, the BASELINE layer shapes
in test_conv_mfma_at_flownet_shapes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import flownet2_amd

import oracle

CASES = [  # N, Cin, H, W, Cout, k, s, p
    (2, 8, 12, 16, 64, 3, 1, 1), (1, 13, 9, 20, 64, 3, 1, 1), (2, 5, 11, 12, 128, 3, 2, 1), (1, 16, 16, 24, 64, 5, 2, 2),
    (1, 7, 13, 28, 128, 5, 2, 2), (1, 4, 8, 8, 64, 3, 1, 0), (1, 9, 10, 12, 64, 3, 2, 0), (1, 12, 20, 32, 64, 7, 2, 3), (2, 5, 18, 24, 64, 7, 2, 3),
    (2, 10, 12, 16, 64, 4, 2, 1), (1, 64, 10, 28, 128, 4, 2, 1),      # 4x4 / 2 / 1: the Deconvolution layers' data gradient
    (2, 70, 9, 12, 32, 1, 1, 0), (1, 37, 6, 20, 192, 1, 1, 0)]         # 1x1: conv_redir (32 channels) and the Deconvolution GEMM


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def torch64(x, w, b, s, p, relu):
    y = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=p)
    return (F.leaky_relu(y, 0.1) if relu else y).numpy()


@pytest.mark.parametrize("case", CASES)
def test_oracle_conv_matches_fp64_convolution(case):
    N, Cin, H, W, Cout, k, s, p = case
    x, w, b = rnd((N, Cin, H, W), 1), rnd((Cout, Cin, k, k), 2, 0.2), rnd((Cout,), 3)
    pw = oracle.conv_mfma_pack_weights(w)
    for relu in (True, False):
        got = oracle.conv_mfma_forward(x, pw, b, Cout, k, s, p, relu, 0.1)
        want = torch64(x, w, b, s, p, relu)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def test_oracle_conv_channel_slices():
    """bottom / top as channel slices of wider blobs: what a Concat consumer sees (concat_layer.cu) when the producer writes in place."""
    x, w, b = rnd((2, 10, 8, 12), 4), rnd((64, 6, 3, 3), 5, 0.2), rnd((64,), 6)
    pw = oracle.conv_mfma_pack_weights(w)
    out = np.full((2, 70, 8, 12), 7.0, np.float32)
    oracle.conv_mfma_forward(x, pw, b, 64, 3, 1, 1, True, 0.1, out=out, out_c0=3, in_c0=2, Cin=6)
    want = oracle.conv_mfma_forward(np.ascontiguousarray(x[:, 2:8]), pw, b, 64, 3, 1, 1, True, 0.1)
    assert np.array_equal(out[:, 3:67], want) and (out[:, :3] == 7).all() and (out[:, 67:] == 7).all()


def test_packed_layout():
    w = rnd((64, 5, 3, 3), 7)
    pw = oracle.conv_mfma_pack_weights(w).reshape(1, -1, 64, 4)
    assert pw.shape[1] == 2 * 9 + 8                      # 5 channels -> 2 quads, padded to a whole chunk of 2 quads; 8 spare k-steps
    # lane = 16 * kq + co, element j <-> W[16 j + co][4 cq + kq][ky][kx]
    assert pw[0, 4, 16 * 2 + 3, 1] == w[16 + 3, 2, 1, 1]  # k-step 4 = (cq 0, ky 1, kx 1)
    assert pw[0, 9 + 2, 5, 3] == w[48 + 5, 4, 0, 2]       # k-step 11 = (cq 1, ky 0, kx 2), kq 0 -> channel 4
    assert (pw[0, 9:, 16:, :] == 0).all()                 # channels 5..7 do not exist
    assert (pw[0, 18:] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_conv_equals_oracle_bitwise_in_every_variant(case):
    from flownet2_amd import ops
    N, Cin, H, W, Cout, k, s, p = case
    x, w, b = rnd((N, Cin, H, W), 11), rnd((Cout, Cin, k, k), 12, 0.2), rnd((Cout,), 13)
    dv = lambda a: torch.from_numpy(a).cuda()
    pw = ops.conv_mfma_pack_weights(dv(w))
    assert np.array_equal(pw.cpu().numpy(), oracle.conv_mfma_pack_weights(w))
    want = oracle.conv_mfma_forward(x, pw.cpu().numpy(), b, Cout, k, s, p, True, 0.1)
    ran = 0
    try:
        nv = ops.conv_num_variants()
        for v in list(range(nv)) + [1000 + i for i in range(nv)]:          # plain launches, then the split-tail launches
            ops.set_conv_variant(v)
            try:
                got = ops.conv_mfma_forward(dv(x), pw, dv(b), Cout, k, s, p, True, 0.1)
            except flownet2_amd.Fn2Error:
                continue                                  # variant for another kernel size / stride / channel multiple
            ran += 1
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"variant {v}"
    finally:
        ops.set_conv_variant(-1)
    assert ran >= (3 if Cout % 64 else 5)          # 32-channel layers: only the one-channel-block variants apply
    got = ops.conv_mfma_forward(dv(x), pw, None, Cout, k, s, p, False, 0.1)
    assert np.array_equal(got.cpu().numpy(), oracle.conv_mfma_forward(x, pw.cpu().numpy(), None, Cout, k, s, p, False, 0.1))


@pytest.mark.gpu
def test_hip_conv_channel_slices_and_reference_layer():
    from flownet2_amd import ops
    from oracle import ref
    dv = lambda a: torch.from_numpy(a).cuda()
    x, w, b = rnd((2, 10, 8, 12), 4), rnd((64, 6, 3, 3), 5, 0.2), rnd((64,), 6)
    pw = ops.conv_mfma_pack_weights(dv(w))
    out = torch.full((2, 70, 8, 12), 7.0, device="cuda")
    ops.conv_mfma_forward(dv(x), pw, dv(b), 64, 3, 1, 1, True, 0.1, out=out, out_c0=3, in_c0=2, Cin=6)
    want = oracle.conv_mfma_forward(np.ascontiguousarray(x[:, 2:8]), pw.cpu().numpy(), b, 64, 3, 1, 1, True, 0.1)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, 3:67], want) and (o[:, :3] == 7).all() and (o[:, 67:] == 7).all()
    if ref.available():
        for (k, s, p) in [(3, 1, 1), (3, 2, 1), (5, 2, 2)]:
            x, w, b = rnd((2, 24, 16, 24), 20 + k), rnd((128, 24, k, k), 21 + s, 0.1), rnd((128,), 22)
            r = ref.convolution(x, w, b, kernel=k, stride=s, pad=p, relu=True)
            got = ops.conv_mfma_forward(dv(x), ops.conv_mfma_pack_weights(dv(w)), dv(b), 128, k, s, p, True, 0.1).cpu().numpy()
            assert np.abs(got - r).max() <= 1e-5 * max(1.0, np.abs(r).max())


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [("conv2", 16, 64, 160, 224, 128, 5, 2, 2), ("conv3", 16, 128, 80, 112, 256, 5, 2, 2),
                                   ("conv3_1", 8, 473, 40, 56, 256, 3, 1, 1), ("conv4", 8, 256, 40, 56, 512, 3, 2, 1),
                                   ("conv4_1", 8, 512, 20, 28, 512, 3, 1, 1), ("conv3_1@768", 4, 256, 48, 96, 256, 3, 1, 1),
                                   ("net2_conv1@768", 4, 12, 384, 768, 64, 7, 2, 3)])
def test_conv_mfma_at_flownet_shapes(layer):
    """The layers of BASELINE.json's configs at full size: against torch's fp64-accumulated result on a sample of outputs and
    against MIOpen's fp32 result everywhere, at 1e-5 x scale."""
    from flownet2_amd import ops
    name, N, Cin, H, W, Cout, k, s, p = layer
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    got = ops.conv_mfma_forward(x, ops.conv_mfma_pack_weights(w), b, Cout, k, s, p, True, 0.1)
    lib = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=p), 0.1)
    scale = max(1.0, float(lib.abs().max()))
    assert float((got - lib).abs().max()) <= 1e-5 * scale
    want64 = F.leaky_relu(F.conv2d(x[:1].double(), w.double(), b.double(), stride=s, padding=p), 0.1)
    assert float((got[:1].double() - want64).abs().max()) <= 4e-6 * scale
