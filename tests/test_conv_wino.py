"""Winograd F(2x2, 3x3) convolution (csrc/conv_wino.hip, fn2_conv_wino_*): the oracle twin against torch's fp64 convolution and
the direct-sum oracle (CPU); the HIP kernels against the oracle twin BIT FOR BIT in every tile variant (plain and split-tail
launches), against the reference's own Convolution + ReLU layers (oracle/_ref) and against fp64 at the BASELINE layer shapes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import flownet2_amd
import oracle

CASES = [  # N, Cin, H, W, Cout
    (2, 8, 12, 16, 16), (1, 13, 9, 20, 32), (2, 5, 11, 12, 48), (1, 16, 16, 24, 16), (1, 7, 5, 28, 32), (1, 4, 8, 8, 16), (3, 9, 20, 36, 16)]


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def torch64(x, w, b, relu):
    y = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=1, padding=1)
    return (F.leaky_relu(y, 0.1) if relu else y).numpy()


@pytest.mark.parametrize("case", CASES)
def test_oracle_winograd_matches_fp64_and_the_direct_sum(case):
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 1), rnd((Cout, Cin, 3, 3), 2, 0.2), rnd((Cout,), 3)
    pw = oracle.conv_wino_pack_weights(w)
    got = oracle.conv_wino_forward(x, pw, b, Cout, 1, True, 0.1)
    want = torch64(x, w, b, True)
    scale = max(1.0, np.abs(want).max())
    assert got.shape == want.shape and np.abs(got - want).max() <= 4e-6 * scale
    if Cout % 64 == 0 or True:
        # the two fp32 formulations (direct k-ordered sum vs Winograd) agree at rounding level
        w64 = np.concatenate([w] * (64 // Cout + 1))[:64] if Cout < 64 else w
        direct = oracle.conv_mfma_forward(x, oracle.conv_mfma_pack_weights(w64), None, 64, 3, 1, 1, False, 0.1)[:, :min(Cout, 64)]
        wino = oracle.conv_wino_forward(x, pw, None, Cout, 1, False, 0.1)[:, :min(Cout, 64)]
        assert np.abs(direct - wino).max() <= 4e-6 * scale


def test_packed_u_layout_and_transform():
    w = rnd((16, 5, 3, 3), 7)
    pw = oracle.conv_wino_pack_weights(w).reshape(1, 2, 4, 64, 4)         # 5 channels -> 2 quads (one chunk)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    for (co, ci) in [(3, 0), (11, 4), (0, 2)]:
        U = G @ w[co, ci].astype(np.float64) @ G.T
        got = pw[0, ci // 4, :, 16 * (ci % 4) + co, :]                        # [position quad = xi][nu]
        assert np.abs(got - U).max() <= 1e-6
    assert (pw[0, 1, :, 16:, :] == 0).all()                                   # channels 5..7 do not exist


def test_oracle_winograd_channel_slices():
    x, w, b = rnd((2, 10, 8, 12), 4), rnd((16, 6, 3, 3), 5, 0.2), rnd((16,), 6)
    pw = oracle.conv_wino_pack_weights(w)
    out = np.full((2, 21, 8, 12), 7.0, np.float32)
    oracle.conv_wino_forward(x, pw, b, 16, 1, True, 0.1, out=out, out_c0=3, in_c0=2, Cin=6)
    want = oracle.conv_wino_forward(np.ascontiguousarray(x[:, 2:8]), pw, b, 16, 1, True, 0.1)
    assert np.array_equal(out[:, 3:19], want) and (out[:, :3] == 7).all() and (out[:, 19:] == 7).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_winograd_equals_oracle_bitwise_in_every_variant(case):
    from flownet2_amd import ops
    N, Cin, H, W, Cout = case
    x, w, b = rnd((N, Cin, H, W), 11), rnd((Cout, Cin, 3, 3), 12, 0.2), rnd((Cout,), 13)
    dv = lambda a: torch.from_numpy(a).cuda()
    pw = ops.conv_wino_pack_weights(dv(w))
    assert np.array_equal(pw.cpu().numpy().view(np.uint32), oracle.conv_wino_pack_weights(w).view(np.uint32))
    want = oracle.conv_wino_forward(x, pw.cpu().numpy(), b, Cout, 1, True, 0.1)
    ran = 0
    try:
        nv = ops.wino_num_variants()
        for v in list(range(nv)) + [1000 + i for i in range(nv)]:
            ops.set_wino_variant(v)
            try:
                got = ops.conv_wino_forward(dv(x), pw, dv(b), Cout, 1, True, 0.1)
            except flownet2_amd.Fn2Error:
                continue                                  # a variant without a split-tail form
            ran += 1
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"variant {v}"
    finally:
        ops.set_wino_variant(-1)
    assert ran >= 12
    got = ops.conv_wino_forward(dv(x), pw, None, Cout, 1, False, 0.1)
    assert np.array_equal(got.cpu().numpy(), oracle.conv_wino_forward(x, pw.cpu().numpy(), None, Cout, 1, False, 0.1))


@pytest.mark.gpu
def test_hip_winograd_channel_slices_and_reference_layer():
    from flownet2_amd import ops
    from oracle import ref
    dv = lambda a: torch.from_numpy(a).cuda()
    x, w, b = rnd((2, 10, 8, 12), 4), rnd((16, 6, 3, 3), 5, 0.2), rnd((16,), 6)
    pw = ops.conv_wino_pack_weights(dv(w))
    out = torch.full((2, 21, 8, 12), 7.0, device="cuda")
    ops.conv_wino_forward(dv(x), pw, dv(b), 16, 1, True, 0.1, out=out, out_c0=3, in_c0=2, Cin=6)
    want = oracle.conv_wino_forward(np.ascontiguousarray(x[:, 2:8]), pw.cpu().numpy(), b, 16, 1, True, 0.1)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, 3:19], want) and (o[:, :3] == 7).all() and (o[:, 19:] == 7).all()
    if ref.available():
        x, w, b = rnd((2, 24, 16, 24), 21), rnd((128, 24, 3, 3), 22, 0.1), rnd((128,), 23)
        r = ref.convolution(x, w, b, kernel=3, stride=1, pad=1, relu=True)
        got = ops.conv_wino_forward(dv(x), ops.conv_wino_pack_weights(dv(w)), dv(b), 128, 1, True, 0.1).cpu().numpy()
        assert np.abs(got - r).max() <= 1e-5 * max(1.0, np.abs(r).max())


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [("conv3_1", 8, 473, 40, 56, 256), ("conv4_1", 8, 512, 20, 28, 512), ("conv5_1", 8, 512, 10, 14 + 2, 512),
                                   ("conv3_1@768", 4, 256, 48, 96, 256), ("netsd_conv0", 2, 6, 384, 768, 64)])
def test_winograd_at_flownet_shapes(layer):
    """BASELINE layer shapes: against fp64 on one sample and against the library's fp32 result everywhere, at 1e-5 x scale."""
    from flownet2_amd import ops
    name, N, Cin, H, W, Cout = layer
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    got = ops.conv_wino_forward(x, ops.conv_wino_pack_weights(w), b, Cout, 1, True, 0.1)
    lib = F.leaky_relu(F.conv2d(x, w, b, stride=1, padding=1), 0.1)
    scale = max(1.0, float(lib.abs().max()))
    assert float((got - lib).abs().max()) <= 1e-5 * scale
    want64 = F.leaky_relu(F.conv2d(x[:1].double(), w.double(), b.double(), stride=1, padding=1), 0.1)
    assert float((got[:1].double() - want64).abs().max()) <= 6e-6 * scale
