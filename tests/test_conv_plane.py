"""Small-map 3x3 convolution (csrc/conv_plane.hip, fn2_conv_plane_*): the oracle twin against torch's fp64 convolution (CPU), the HIP
kernels against the oracle BIT FOR BIT in every tile variant and for several K splits, against the reference's own Convolution + ReLU
layers (oracle/_ref: conv_layer.cu:8-23, base_conv_layer.cpp:326-348, relu_layer.cu:8-27), and the encoder layers of BASELINE.json's
configs below 1/16 resolution at full size."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import flownet2_amd

import oracle

CASES = [  # N, Cin, H, W, Cout, s, p    (odd widths -> dword DMA, widths % 4 == 0 -> 16-byte DMA; ragged batches; one sample larger than a tile block)
    (8, 16, 5, 7, 64, 1, 1), (3, 8, 10, 14, 128, 1, 1), (5, 16, 10, 14, 64, 2, 1), (2, 8, 20, 28, 64, 2, 1), (2, 16, 12, 24, 64, 1, 1),
    (1, 8, 40, 56, 64, 2, 1), (3, 24, 6, 12, 64, 1, 0), (2, 8, 14, 32, 128, 1, 1), (1, 32, 24, 48, 64, 2, 1), (7, 8, 7, 16, 64, 2, 1),
    (2, 16, 4, 6, 64, 1, 1), (2, 16, 4, 6, 128, 2, 1), (2, 24, 2, 3, 64, 1, 1), (1, 8, 3, 4, 64, 1, 1), (1, 8, 1, 1, 64, 1, 1)]      # planes smaller than one 64-slot DMA run (a 192x128 image at 1/32 .. 1/64)


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def torch64(x, w, b, s, p, relu):
    y = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double() if b is not None else None, stride=s, padding=p)
    return (F.leaky_relu(y, 0.1) if relu else y).numpy()


@pytest.mark.parametrize("case", CASES[:6])
def test_oracle_plane_conv_matches_fp64_convolution(case):
    N, Cin, H, W, Cout, s, p = case
    x, w, b = rnd((N, Cin, H, W), 1), rnd((Cout, Cin, 3, 3), 2, 0.2), rnd((Cout,), 3)
    pw = oracle.conv_mfma_pack_weights(w)
    for ksplit in (1, 2) if Cin >= 16 else (1,):
        for relu in (True, False):
            got = oracle.conv_plane_forward(x, pw, b, Cout, s, p, ksplit, relu, 0.1)
            want = torch64(x, w, b, s, p, relu)
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    # one part = the direct kernel's chain, bit for bit
    assert np.array_equal(oracle.conv_plane_forward(x, pw, b, Cout, s, p, 1), oracle.conv_mfma_forward(x, pw, b, Cout, 3, s, p))


CASES_K4 = [  # N, Cin, H, W, Cout: Convolution{4, 2, 1} = the data gradient of a Deconvolution{4, 2, 1} (deconv5 of FlowNetC: 10x14 -> 5x7)
    (3, 16, 10, 14, 64), (8, 8, 10, 14, 128), (2, 16, 12, 24, 64), (1, 8, 20, 28, 64), (5, 24, 6, 10, 64), (2, 8, 7, 9, 64)]


@pytest.mark.parametrize("case", CASES_K4[:4])
def test_oracle_plane_conv_k4s2_matches_fp64_convolution(case):
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 21), rnd((Cout, Cin, 4, 4), 22, 0.2), rnd((Cout,), 23)
    pw = oracle.conv_mfma_pack_weights(w)
    for ksplit in (1, 2) if Cin >= 16 else (1,):
        got = oracle.conv_plane_forward(x, pw, b, Cout, 2, 1, ksplit, False, 0.1, kernel=4)
        want = torch64(x, w, b, 2, 1, False)
        assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    assert np.array_equal(oracle.conv_plane_forward(x, pw, b, Cout, 2, 1, 1, kernel=4), oracle.conv_mfma_forward(x, pw, b, Cout, 4, 2, 1))


CASES_K5 = [  # N, Cin, H, W, Cout: Convolution{5, 2, 2} (conv2 / conv3 of the encoders) on one or two samples
    (1, 16, 28, 64, 64), (1, 8, 15, 29, 64), (2, 24, 16, 24, 128), (1, 16, 56, 128, 64), (3, 8, 9, 13, 64), (1, 32, 30, 256, 64),
    (2, 16, 28, 64, 64), (4, 16, 64, 96, 128), (4, 32, 32, 48, 256), (3, 8, 20, 36, 64)]      # row bands of several samples


@pytest.mark.parametrize("case", CASES_K5[:3])
def test_oracle_plane_conv_k5s2_matches_fp64_convolution(case):
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 41), rnd((Cout, Cin, 5, 5), 42, 0.2), rnd((Cout,), 43)
    pw = oracle.conv_mfma_pack_weights(w)
    for ksplit in (1, 2) if Cin >= 16 else (1,):
        got = oracle.conv_plane_forward(x, pw, b, Cout, 2, 2, ksplit, False, 0.1, kernel=5)
        want = torch64(x, w, b, 2, 2, False)
        assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    assert np.array_equal(oracle.conv_plane_forward(x, pw, b, Cout, 2, 2, 1, kernel=5), oracle.conv_mfma_forward(x, pw, b, Cout, 5, 2, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES_K5)
def test_hip_plane_conv_k5s2_equals_oracle_bitwise_in_every_variant(case):
    from flownet2_amd import ops
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 51), rnd((Cout, Cin, 5, 5), 52, 0.2), rnd((Cout,), 53)
    dv = lambda a: torch.from_numpy(a).cuda()
    pw = ops.conv_mfma_pack_weights(dv(w))
    pwh = pw.cpu().numpy()
    assert ops.conv_plane_k_supported(N, Cin, H, W, Cout, 5, 2, 2) and not ops.conv_plane_k_supported(N, Cin, H, W, Cout, 5, 1, 2)
    assert not ops.conv_plane_k_supported(N, Cin, H, W, Cout, 5, 2, 1)
    ran = 0
    try:
        for ksplit in (0, 1, 2):
            ops.set_plane_ksplit(ksplit)
            ks = ops.conv_plane_k_ksplit(N, Cin, H, W, Cout, 5, 2, 2)
            want = oracle.conv_plane_forward(x, pwh, b, Cout, 2, 2, ks, True, 0.1, kernel=5)
            for v in range(ops.plane_num_variants()):
                ops.set_plane_variant(v)
                try:
                    got = ops.conv_plane_forward(dv(x), pw, dv(b), Cout, 2, 2, True, 0.1, kernel=5)
                except flownet2_amd.Fn2Error:
                    continue                              # another tap class, the other DMA width, or a window beyond the LDS
                ran += 1
                assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"variant {v}, ksplit {ks}"
            ops.set_plane_variant(-1)
            got = ops.conv_plane_forward(dv(x), pw, None, Cout, 2, 2, False, 0.1, kernel=5)
            assert np.array_equal(got.cpu().numpy(), oracle.conv_plane_forward(x, pwh, None, Cout, 2, 2, ks, False, 0.1, kernel=5))
    finally:
        ops.set_plane_variant(-1)
        ops.set_plane_ksplit(0)
    assert ran >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES_K4)
def test_hip_plane_conv_k4s2_equals_oracle_bitwise_in_every_variant(case):
    from flownet2_amd import ops
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 31), rnd((Cout, Cin, 4, 4), 32, 0.2), rnd((Cout,), 33)
    dv = lambda a: torch.from_numpy(a).cuda()
    pw = ops.conv_mfma_pack_weights(dv(w))
    pwh = pw.cpu().numpy()
    assert ops.conv_plane_k_supported(N, Cin, H, W, Cout, 4, 2, 1) and not ops.conv_plane_k_supported(N, Cin, H, W, Cout, 4, 1, 1)
    ran = 0
    try:
        for ksplit in (0, 1, 2):
            ops.set_plane_ksplit(ksplit)
            ks = ops.conv_plane_k_ksplit(N, Cin, H, W, Cout, 4, 2, 1)
            want = oracle.conv_plane_forward(x, pwh, b, Cout, 2, 1, ks, True, 0.1, kernel=4)
            for v in range(ops.plane_num_variants()):
                ops.set_plane_variant(v)
                try:
                    got = ops.conv_plane_forward(dv(x), pw, dv(b), Cout, 2, 1, True, 0.1, kernel=4)
                except flownet2_amd.Fn2Error:
                    continue                              # a 3x3 / deconvolution variant, or the other DMA width
                ran += 1
                assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"variant {v}, ksplit {ks}"
            ops.set_plane_variant(-1)
            got = ops.conv_plane_forward(dv(x), pw, None, Cout, 2, 1, False, 0.1, kernel=4)
            assert np.array_equal(got.cpu().numpy(), oracle.conv_plane_forward(x, pwh, None, Cout, 2, 1, ks, False, 0.1, kernel=4))
    finally:
        ops.set_plane_variant(-1)
        ops.set_plane_ksplit(0)
    assert ran >= 2


def test_oracle_plane_conv_channel_slices():
    x, w, b = rnd((2, 20, 6, 9), 4), rnd((64, 8, 3, 3), 5, 0.2), rnd((64,), 6)
    pw = oracle.conv_mfma_pack_weights(w)
    out = np.full((2, 70, 6, 9), 7.0, np.float32)
    oracle.conv_plane_forward(x, pw, b, 64, 1, 1, 1, True, 0.1, out=out, out_c0=3, in_c0=4, Cin=8)
    want = oracle.conv_plane_forward(np.ascontiguousarray(x[:, 4:12]), pw, b, 64, 1, 1, 1, True, 0.1)
    assert np.array_equal(out[:, 3:67], want) and (out[:, :3] == 7).all() and (out[:, 67:] == 7).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_plane_conv_equals_oracle_bitwise_in_every_variant(case):
    from flownet2_amd import ops
    N, Cin, H, W, Cout, s, p = case
    x, w, b = rnd((N, Cin, H, W), 11), rnd((Cout, Cin, 3, 3), 12, 0.2), rnd((Cout,), 13)
    dv = lambda a: torch.from_numpy(a).cuda()
    pw = ops.conv_mfma_pack_weights(dv(w))
    pwh = pw.cpu().numpy()
    assert ops.conv_plane_supported(N, Cin, H, W, Cout, s, p)
    ran = 0
    try:
        for ksplit in (0, 1, 2, 3, 4):                   # 0: the geometry's own; forced values are clamped to the number of 2-quad units
            ops.set_plane_ksplit(ksplit)
            ks = ops.conv_plane_ksplit(N, Cin, H, W, Cout, s, p)
            assert ksplit == 0 or ks == ksplit or ks <= (Cin // 4 + 1) // 2
            want = oracle.conv_plane_forward(x, pwh, b, Cout, s, p, ks, True, 0.1)
            for v in range(ops.plane_num_variants()):
                ops.set_plane_variant(v)
                try:
                    got = ops.conv_plane_forward(dv(x), pw, dv(b), Cout, s, p, True, 0.1)
                except flownet2_amd.Fn2Error:
                    continue                              # variant of the other stride / DMA width, or a plane too small for 16-byte runs
                ran += 1
                assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"variant {v}, ksplit {ks}"
            ops.set_plane_variant(-1)
            got = ops.conv_plane_forward(dv(x), pw, None, Cout, s, p, False, 0.1)              # the autotuned pick, no bias, no ReLU
            assert np.array_equal(got.cpu().numpy(), oracle.conv_plane_forward(x, pwh, None, Cout, s, p, ks, False, 0.1))
    finally:
        ops.set_plane_variant(-1)
        ops.set_plane_ksplit(0)
    assert ran >= 2


@pytest.mark.gpu
def test_hip_plane_conv_channel_slices_and_reference_layer():
    from flownet2_amd import ops
    from oracle import ref
    dv = lambda a: torch.from_numpy(a).cuda()
    x, w, b = rnd((3, 20, 6, 9), 4), rnd((64, 8, 3, 3), 5, 0.2), rnd((64,), 6)
    pw = ops.conv_mfma_pack_weights(dv(w))
    out = torch.full((3, 70, 6, 9), 7.0, device="cuda")
    ops.conv_plane_forward(dv(x), pw, dv(b), 64, 1, 1, True, 0.1, out=out, out_c0=3, in_c0=4, Cin=8)
    ks = ops.conv_plane_ksplit(3, 8, 6, 9, 64, 1, 1)
    want = oracle.conv_plane_forward(np.ascontiguousarray(x[:, 4:12]), pw.cpu().numpy(), b, 64, 1, 1, ks, True, 0.1)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, 3:67], want) and (o[:, :3] == 7).all() and (o[:, 67:] == 7).all()
    if ref.available():
        for (s, H, W) in [(1, 10, 14), (2, 20, 28), (1, 5, 7)]:
            x, w, b = rnd((4, 64, H, W), 20 + s), rnd((128, 64, 3, 3), 21 + s, 0.1), rnd((128,), 22)
            r = ref.convolution(x, w, b, kernel=3, stride=s, pad=1, relu=True)
            got = ops.conv_plane_forward(dv(x), ops.conv_mfma_pack_weights(dv(w)), dv(b), 128, s, 1, True, 0.1).cpu().numpy()
            assert np.abs(got - r).max() <= 1e-5 * max(1.0, np.abs(r).max())


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [("conv4", 8, 256, 40, 56, 512, 2), ("conv5", 8, 512, 20, 28, 512, 2), ("conv5_1", 8, 512, 10, 14, 512, 1),
                                   ("conv6", 8, 512, 10, 14, 1024, 2), ("conv6_1", 8, 1024, 5, 7, 1024, 1),
                                   ("conv4@768", 4, 256, 48, 96, 512, 2), ("conv4@1024", 1, 256, 56, 128, 512, 2), ("conv5@768", 4, 512, 24, 48, 512, 2), ("conv5_1@768", 4, 512, 12, 24, 512, 1), ("conv6_1@768", 4, 1024, 6, 12, 1024, 1),
                                   ("conv5_1@1024", 1, 512, 14, 32, 512, 1), ("conv6_1@1024", 1, 1024, 7, 16, 1024, 1)])
def test_conv_plane_at_flownet_shapes(layer):
    """The small-map encoder layers of BASELINE.json's configs at full size: against the library's fp32 result everywhere and torch's
    fp64-accumulated result on the first sample, at 1e-5 x scale."""
    from flownet2_amd import ops
    name, N, Cin, H, W, Cout, s = layer
    assert ops.conv_plane_supported(N, Cin, H, W, Cout, s, 1), name
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    got = ops.conv_plane_forward(x, ops.conv_mfma_pack_weights(w), b, Cout, s, 1, True, 0.1)
    lib = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=1), 0.1)
    scale = max(1.0, float(lib.abs().max()))
    assert float((got - lib).abs().max()) <= 1e-5 * scale
    want64 = F.leaky_relu(F.conv2d(x[:1].double(), w.double(), b.double(), stride=s, padding=1), 0.1)
    assert float((got[:1].double() - want64).abs().max()) <= 4e-6 * scale


# ------------------------------------------------------------------------------------------------ deconvolution 4x4 / stride 2 / pad 1
DECONV_CASES = [  # N, Cin, H, W, Cout     (Cin not a multiple of 4 / 8: the refinement stages concatenate 2 flow channels)
    (8, 16, 5, 7, 64), (3, 10, 10, 14, 128), (2, 26, 20, 28, 64), (1, 6, 40, 56, 64), (2, 16, 6, 12, 64), (5, 32, 7, 16, 64), (1, 9, 12, 24, 128),
    (2, 16, 2, 3, 64), (1, 10, 1, 1, 64), (3, 8, 4, 6, 128)]      # planes smaller than one DMA run


def torch64_deconv(x, w, b, relu):
    y = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double() if b is not None else None,
                           stride=2, padding=1)
    return (F.leaky_relu(y, 0.1) if relu else y).numpy()


@pytest.mark.parametrize("case", DECONV_CASES[:5])
def test_oracle_plane_deconv_matches_fp64_deconvolution(case):
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 1), rnd((Cin, Cout, 4, 4), 2, 0.2), rnd((Cout,), 3)
    pw = oracle.deconv_plane_pack_weights(w)
    for ksplit in (1, 2) if Cin > 8 else (1,):
        for relu in (True, False):
            got = oracle.deconv_plane_forward(x, pw, b, Cout, ksplit, relu, 0.1)
            want = torch64_deconv(x, w, b, relu)
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def test_oracle_plane_deconv_matches_the_oracle_deconvolution_layer_and_slices():
    """Against the oracle's plain Deconvolution restatement (deconv_layer.cpp:8-26 through col2im), and into a channel slice of a wider blob."""
    x, w, b = rnd((2, 10, 6, 9), 4), rnd((10, 64, 4, 4), 5, 0.2), rnd((64,), 6)
    pw = oracle.deconv_plane_pack_weights(w)
    want = torch64_deconv(x, w, b, True)
    out = np.full((2, 70, 12, 18), 7.0, np.float32)
    big = np.concatenate([rnd((2, 3, 6, 9), 7), x, rnd((2, 2, 6, 9), 8)], 1)
    oracle.deconv_plane_forward(big, pw, b, 64, 1, True, 0.1, out=out, out_c0=4, in_c0=3, Cin=10)
    assert np.abs(out[:, 4:68] - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    assert (out[:, :4] == 7).all() and (out[:, 68:] == 7).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", DECONV_CASES)
def test_hip_plane_deconv_equals_oracle_bitwise_in_every_variant(case):
    from flownet2_amd import ops
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 11), rnd((Cin, Cout, 4, 4), 12, 0.2), rnd((Cout,), 13)
    dv = lambda a: torch.from_numpy(a).cuda()
    pw = ops.deconv_plane_pack_weights(dv(w))
    pwh = pw.cpu().numpy()
    assert np.array_equal(pwh, oracle.deconv_plane_pack_weights(w))
    assert ops.deconv_plane_supported(N, Cin, H, W, Cout)
    ran = 0
    try:
        for ksplit in (0, 1, 2, 3):
            ops.set_plane_ksplit(ksplit)
            ks = ops.deconv_plane_ksplit(N, Cin, H, W, Cout)
            want = oracle.deconv_plane_forward(x, pwh, b, Cout, ks, True, 0.1)
            for v in range(ops.plane_num_variants()):
                ops.set_plane_variant(v)
                try:
                    got = ops.deconv_plane_forward(dv(x), pw, dv(b), Cout, True, 0.1)
                except flownet2_amd.Fn2Error:
                    continue
                ran += 1
                assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"variant {v}, ksplit {ks}"
            ops.set_plane_variant(-1)
            got = ops.deconv_plane_forward(dv(x), pw, None, Cout, False, 0.1)
            assert np.array_equal(got.cpu().numpy(), oracle.deconv_plane_forward(x, pwh, None, Cout, ks, False, 0.1))
    finally:
        ops.set_plane_variant(-1)
        ops.set_plane_ksplit(0)
    assert ran >= 2


@pytest.mark.gpu
def test_hip_plane_deconv_channel_slices_and_reference_layer():
    from flownet2_amd import ops
    from oracle import ref
    dv = lambda a: torch.from_numpy(a).cuda()
    x, w, b = rnd((3, 10, 6, 9), 4), rnd((10, 64, 4, 4), 5, 0.2), rnd((64,), 6)
    big = np.concatenate([rnd((3, 3, 6, 9), 7), x, rnd((3, 2, 6, 9), 8)], 1)
    pw = ops.deconv_plane_pack_weights(dv(w))
    out = torch.full((3, 70, 12, 18), 7.0, device="cuda")
    ops.deconv_plane_forward(dv(big), pw, dv(b), 64, True, 0.1, out=out, out_c0=4, in_c0=3, Cin=10)
    ks = ops.deconv_plane_ksplit(3, 10, 6, 9, 64)
    want = oracle.deconv_plane_forward(x, pw.cpu().numpy(), b, 64, ks, True, 0.1)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, 4:68], want) and (o[:, :4] == 7).all() and (o[:, 68:] == 7).all()
    if ref.available():
        x, w, b = rnd((2, 34, 10, 14), 20), rnd((34, 128, 4, 4), 21, 0.1), rnd((128,), 22)
        r = ref.convolution(x, w, b, kernel=4, stride=2, pad=1, deconv=True, relu=True)
        got = ops.deconv_plane_forward(dv(x), ops.deconv_plane_pack_weights(dv(w)), dv(b), 128, True, 0.1).cpu().numpy()
        assert np.abs(got - r).max() <= 1e-5 * max(1.0, np.abs(r).max())


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [("deconv5", 8, 1024, 5, 7, 512), ("deconv4", 8, 1026, 10, 14, 256), ("deconv3", 8, 770, 20, 28, 128),
                                   ("deconv2", 8, 386, 40, 56, 64), ("deconv4@768", 4, 1026, 12, 24, 256), ("deconv3@768", 4, 770, 24, 48, 128),
                                   ("deconv5@1024", 1, 1024, 7, 16, 512), ("deconv3@1024", 1, 770, 28, 64, 128)])
def test_deconv_plane_at_flownet_shapes(layer):
    """The refinement deconvolutions of BASELINE.json's configs at full size: against the library's fp32 result everywhere and torch's
    fp64-accumulated result on the first sample, at 1e-5 x scale."""
    from flownet2_amd import ops
    name, N, Cin, H, W, Cout = layer
    assert ops.deconv_plane_supported(N, Cin, H, W, Cout), name
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cin, Cout, 4, 4, device="cuda", generator=g) * (2.0 / (Cin * 4)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    got = ops.deconv_plane_forward(x, ops.deconv_plane_pack_weights(w), b, Cout, True, 0.1)
    lib = F.leaky_relu(F.conv_transpose2d(x, w, b, stride=2, padding=1), 0.1)
    scale = max(1.0, float(lib.abs().max()))
    assert float((got - lib).abs().max()) <= 1e-5 * scale
    want64 = F.leaky_relu(F.conv_transpose2d(x[:1].double(), w.double(), b.double(), stride=2, padding=1), 0.1)
    assert float((got[:1].double() - want64).abs().max()) <= 4e-6 * scale


# ------------------------------------------------------------------------------------------------ seeded random geometries
def _random_geometries(seed, n, deconv):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        N = int(rng.integers(1, 10))
        H, W = int(rng.integers(1, 34)), int(rng.integers(1, 50))
        Cout = 64 * int(rng.integers(1, 3))
        if deconv:
            Cin = int(rng.integers(1, 41))
            out.append((N, Cin, H, W, Cout))
        else:
            Cin = 8 * int(rng.integers(1, 6))
            s, p = int(rng.integers(1, 3)), int(rng.integers(0, 2))
            if H + 2 * p < 3 or W + 2 * p < 3:
                continue
            out.append((N, Cin, H, W, Cout, s, p))
    return out


@pytest.mark.gpu
def test_hip_plane_conv_random_geometries_equal_oracle_bitwise():
    """40 seeded random layer geometries (sample groups, row bands, odd widths, ragged batches, both strides and paddings): wherever the
    kernel family reports support, the autotuned launch and two forced tile variants equal the oracle twin bit for bit."""
    from flownet2_amd import ops
    dv = lambda a: torch.from_numpy(a).cuda()
    ran = 0
    for i, (N, Cin, H, W, Cout, s, p) in enumerate(_random_geometries(2024, 40, False)):
        if not ops.conv_plane_supported(N, Cin, H, W, Cout, s, p):
            continue
        x, w, b = rnd((N, Cin, H, W), 100 + i), rnd((Cout, Cin, 3, 3), 200 + i, 0.2), rnd((Cout,), 300 + i)
        pw = ops.conv_mfma_pack_weights(dv(w))
        ks = ops.conv_plane_ksplit(N, Cin, H, W, Cout, s, p)
        want = oracle.conv_plane_forward(x, pw.cpu().numpy(), b, Cout, s, p, ks, True, 0.1)
        got = ops.conv_plane_forward(dv(x), pw, dv(b), Cout, s, p, True, 0.1).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, Cin, H, W, Cout, s, p, ks)
        try:
            for v in range(ops.plane_num_variants()):
                if v % 5 != i % 5:
                    continue
                ops.set_plane_variant(v)
                try:
                    got = ops.conv_plane_forward(dv(x), pw, dv(b), Cout, s, p, True, 0.1).cpu().numpy()
                except flownet2_amd.Fn2Error:
                    continue
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, Cin, H, W, Cout, s, p, ks, v)
        finally:
            ops.set_plane_variant(-1)
        ran += 1
    assert ran >= 15


@pytest.mark.gpu
def test_hip_plane_deconv_random_geometries_equal_oracle_bitwise():
    from flownet2_amd import ops
    dv = lambda a: torch.from_numpy(a).cuda()
    ran = 0
    for i, (N, Cin, H, W, Cout) in enumerate(_random_geometries(4048, 30, True)):
        if not ops.deconv_plane_supported(N, Cin, H, W, Cout):
            continue
        x, w, b = rnd((N, Cin, H, W), 400 + i), rnd((Cin, Cout, 4, 4), 500 + i, 0.2), rnd((Cout,), 600 + i)
        pw = ops.deconv_plane_pack_weights(dv(w))
        ks = ops.deconv_plane_ksplit(N, Cin, H, W, Cout)
        want = oracle.deconv_plane_forward(x, pw.cpu().numpy(), b, Cout, ks, True, 0.1)
        got = ops.deconv_plane_forward(dv(x), pw, dv(b), Cout, True, 0.1).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, Cin, H, W, Cout, ks)
        try:
            for v in range(ops.plane_num_variants()):
                if v % 4 != i % 4:
                    continue
                ops.set_plane_variant(v)
                try:
                    got = ops.deconv_plane_forward(dv(x), pw, dv(b), Cout, True, 0.1).cpu().numpy()
                except flownet2_amd.Fn2Error:
                    continue
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, Cin, H, W, Cout, ks, v)
        finally:
            ops.set_plane_variant(-1)
        ran += 1
    assert ran >= 12


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 64, 4), (10, 128, 4), (26, 64, 3), (512, 256, 3), (9, 64, 3)])
def test_hip_deconv_weight_packing_equals_the_oracle_bitwise(shape):
    """fn2_deconv_plane_pack_weights_k (LDS-tiled): the 4x4 blob, and a 3x3 blob read as the 4x4 one with a zero fourth tap row / column
    (== packing torch's zero-padded copy)."""
    from flownet2_amd import ops
    Cin, Cout, k = shape
    w = rnd((Cin, Cout, k, k), 61)
    got = ops.deconv_plane_pack_weights(torch.from_numpy(w).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.deconv_plane_pack_weights(w))
    if k == 3:
        padded = F.pad(torch.from_numpy(w), (0, 1, 0, 1)).contiguous()
        assert np.array_equal(got, ops.deconv_plane_pack_weights(padded.cuda()).cpu().numpy())
        assert np.array_equal(got, oracle.deconv_plane_pack_weights(padded.numpy()))
