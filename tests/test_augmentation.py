"""FlowAugmentation: the coefficient-array -> matrix helpers and the flow warp.

CPU part: the C oracle and the host function of libflownet2_hip.so against an independent fp64 composition of the affine maps.
GPU part (-m gpu): HIP kernel vs oracle, vs the reference's own layer (oracle/_ref, live and through the golden file) and through
the Layer mirror."""
import os
import sys

import numpy as np
import pytest
import torch

import flownet2_amd
import oracle
from flownet2_amd import ops
from oracle import ref

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

GOLD = os.path.join(HERE, "golden", "ref_golden.npz")


def coeff_array(mirror=0.0, dx=0.0, dy=0.0, angle=0.0, zoom_x=1.0, zoom_y=1.0):
    """One AugmentationCoeff as coeff_to_array writes it (augmentation_layer_base.cpp:352-365): declaration order, log() for the
    fields whose default is 1.  Everything but the six spatial fields stays at its default (0, or log 1 = 0)."""
    a = np.zeros(42, np.float32)
    a[:6] = [mirror, dx, dy, angle, np.log(zoom_x), np.log(zoom_y)]
    return a


def matrix64(mirror, dx, dy, angle, zx, zy, cw, ch, W, H):
    """fp64 composition of tTransMat::fromCoeff (augmentation_layer_base.cpp:38-49) as 3x3 homogeneous matrices, applied left to right."""
    def T(tx, ty): return np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1.0]])
    M = np.eye(3)
    M = (np.array([[-1, 0, .5 * cw], [0, 1, -.5 * ch], [0, 0, 1.0]]) if mirror else T(-.5 * cw, -.5 * ch)) @ M
    c, s = np.cos(angle), np.sin(angle)
    M = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]) @ M
    M = T(dx * cw, dy * ch) @ M
    M = np.diag([1 / zx, 1 / zy, 1.0]) @ M
    return T(.5 * W, .5 * H) @ M


def as6(M):
    return np.array([M[0, 0], M[1, 0], M[0, 1], M[1, 1], M[0, 2], M[1, 2]])     # t0 t1 t2 t3 t4 t5


CASES = [  # mirror, dx, dy, angle, zoom_x, zoom_y, crop_w, crop_h, W, H
    (0, 0, 0, 0, 1, 1, 64, 48, 64, 48), (0, 0.05, -0.1, 0.2, 1.2, 0.9, 56, 40, 64, 48), (1, -0.08, 0.03, -0.35, 0.8, 1.1, 448, 320, 512, 384),
    (1, 0, 0, 0, 1, 1, 30, 20, 31, 21), (0, 0.2, 0.2, 3.0, 2.0, 2.0, 16, 16, 40, 24),
]


@pytest.mark.parametrize("case", CASES)
def test_augmentation_matrix_matches_fp64_composition(case):
    mirror, dx, dy, angle, zx, zy, cw, ch, W, H = case
    arr = coeff_array(mirror, dx, dy, angle, zx, zy)
    want = matrix64(mirror, dx, dy, angle, zx, zy, cw, ch, W, H)
    for impl in (oracle.augmentation_matrix, ops.augmentation_matrix):
        m = impl(arr, cw, ch, W, H)
        np.testing.assert_allclose(m, as6(want), rtol=2e-6, atol=2e-4)
        inv = impl(arr, cw, ch, W, H, invert=True)
        np.testing.assert_allclose(inv, as6(np.linalg.inv(want)), rtol=2e-5, atol=5e-4)
    # the product's host function and the oracle agree to the bit (same operation order, same number types)
    assert np.array_equal(oracle.augmentation_matrix(arr, cw, ch, W, H).view(np.uint32), ops.augmentation_matrix(arr, cw, ch, W, H).view(np.uint32))
    assert np.array_equal(oracle.augmentation_matrix(arr, cw, ch, W, H, True).view(np.uint32), ops.augmentation_matrix(arr, cw, ch, W, H, True).view(np.uint32))


def smooth_flow(N, H, W, seed, mag=6.0):
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randn(N, 2, 4, 5, generator=g) * mag
    return torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True).numpy().astype(np.float32)


def flow_aug64(flow, c1, c2, ch, cw):
    """fp64 statement of flow_augmentation_layer.cu:36-61 with nearest-source sampling (round half up, as (int)(p + 0.5) for p >= -0.5)."""
    N, _, H, W = flow.shape
    out = np.zeros((N, 2, ch, cw))
    ys, xs = np.mgrid[0:ch, 0:cw].astype(np.float64)
    for n in range(N):
        M1 = matrix64(c1[n][0], c1[n][1], c1[n][2], c1[n][3], np.exp(np.float64(c1[n][4])), np.exp(np.float64(c1[n][5])), cw, ch, W, H)
        M2 = np.linalg.inv(matrix64(c2[n][0], c2[n][1], c2[n][2], c2[n][3], np.exp(np.float64(c2[n][4])), np.exp(np.float64(c2[n][5])), cw, ch, W, H))
        x1 = M1[0, 0] * xs + M1[0, 1] * ys + M1[0, 2]
        y1 = M1[1, 0] * xs + M1[1, 1] * ys + M1[1, 2]
        xi, yi = np.trunc(x1 + 0.5).astype(int), np.trunc(y1 + 0.5).astype(int)
        assert xi.min() >= 0 and yi.min() >= 0 and xi.max() < W and yi.max() < H, "test transform leaves the image"
        x2, y2 = x1 + flow[n, 0][yi, xi], y1 + flow[n, 1][yi, xi]
        out[n, 0] = M2[0, 0] * x2 + M2[0, 1] * y2 + M2[0, 2] - xs
        out[n, 1] = M2[1, 0] * x2 + M2[1, 1] * y2 + M2[1, 2] - ys
    return out


def aug_inputs(N, H, W, ch, cw, seed):
    rng = np.random.default_rng(seed)
    def coeffs():
        return np.stack([coeff_array(float(rng.random() < 0.5), rng.uniform(-0.04, 0.04), rng.uniform(-0.04, 0.04), rng.uniform(-0.15, 0.15),
                                     rng.uniform(0.95, 1.25), rng.uniform(0.95, 1.25)) for _ in range(N)])
    return smooth_flow(N, H, W, seed), coeffs(), coeffs()


@pytest.mark.parametrize("shape", [(2, 48, 64, 32, 44), (3, 96, 128, 64, 96)])
def test_oracle_flow_augmentation_vs_fp64(shape):
    N, H, W, ch, cw = shape
    flow, c1, c2 = aug_inputs(N, H, W, ch, cw, 5)
    got = oracle.flow_augmentation_forward(flow, c1, c2, ch, cw)
    want = flow_aug64(flow, c1, c2, ch, cw)
    # a float rounding of the sampling position can pick the neighbouring source pixel; the flow is smooth, so that costs < 0.5 px
    err = np.abs(got - want)
    assert np.quantile(err, 0.999) < 2e-3 and err.max() < 0.6


def test_flow_augmentation_invariants_and_errors():
    N, H, W = 2, 40, 56
    flow = smooth_flow(N, H, W, 7)
    ident = np.stack([coeff_array()] * N)
    # identity coefficients, crop = image: the flow comes back (up to the rounding of (x + u) - x)
    top = oracle.flow_augmentation_forward(flow, ident, ident, H, W)
    assert np.abs(top - flow).max() < 1e-4
    # the same transform on both images and no motion: no flow
    c = np.stack([coeff_array(1, 0.03, -0.02, 0.1, 1.1, 1.2)] * N)
    top = oracle.flow_augmentation_forward(np.zeros_like(flow), c, c, 30, 40)
    assert np.abs(top).max() < 1e-3
    # a centred crop without any transform: the window of the flow field
    top = oracle.flow_augmentation_forward(flow, ident, ident, 20, 30)
    assert np.abs(top - flow[:, :, 10:30, 13:43]).max() < 1e-4
    with pytest.raises(ValueError):
        oracle.flow_augmentation_forward(flow, ident, ident, 0, 10)
    with pytest.raises(flownet2_amd.Fn2Error):
        ops.augmentation_matrix(coeff_array(), 0, 10, 10, 10)
    with pytest.raises(ValueError, match="no CPU path"):
        ops.flow_augmentation_forward(torch.zeros(1, 2, 4, 4), ident[:1], ident[:1], 2, 2)


def test_flow_augmentation_layer_mirror_checks():
    from flownet2_amd.layers import Blob, CheckError, LayerParameter, LayerRegistry
    b = [Blob(2, 2, 8, 8, device="cpu"), Blob(2, 42, 1, 1, device="cpu"), Blob(2, 42, 1, 1, device="cpu")]
    layer = LayerRegistry.CreateLayer(LayerParameter(type="FlowAugmentation", augmentation_param=dict(crop_width=6, crop_height=4)))
    top = [Blob(device="cpu")]
    layer.SetUp(b, top)
    assert top[0].shape() == [2, 2, 4, 6] and not layer.AllowBackward() and not layer.layer_param_.reshape_every_iter
    for bottoms, ap, msg in [(b, dict(crop_height=4), "Please enter crop width"), (b, dict(crop_width=6), "Please enter crop height"),
                             (b[:2], dict(crop_width=6, crop_height=4), "takes three input blobs"),
                             ([Blob(2, 3, 8, 8, device="cpu")] + b[1:], dict(crop_width=6, crop_height=4), "two channels")]:
        with pytest.raises(CheckError, match=msg):
            LayerRegistry.CreateLayer(LayerParameter(type="FlowAugmentation", augmentation_param=ap)).SetUp(bottoms, [Blob(device="cpu")])


# ---------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------
def _close_flow(a, b, what):
    """Positions are compared in pixels: identical arithmetic up to the contraction of a*b + c*d + e, which may pick the neighbouring
    source pixel for a handful of outputs of a smooth field."""
    err = np.abs(a - b)
    assert np.quantile(err, 0.999) <= 2e-4 and err.max() < 0.6, (what, float(np.quantile(err, 0.999)), float(err.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 48, 64, 32, 44), (3, 96, 128, 64, 96), (70, 24, 32, 16, 20), (8, 384, 512, 320, 448)])
def test_hip_flow_augmentation_matches_oracle(shape):
    N, H, W, ch, cw = shape
    flow, c1, c2 = aug_inputs(N, H, W, ch, cw, 11)
    got = ops.flow_augmentation_forward(torch.from_numpy(flow).cuda(), c1, torch.from_numpy(c2).cuda(), ch, cw).cpu().numpy()
    _close_flow(got, oracle.flow_augmentation_forward(flow, c1, c2, ch, cw), "hip vs oracle")


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("shape", [(2, 48, 64, 32, 44), (4, 96, 128, 64, 96)])
def test_reference_flow_augmentation_equals_oracle_and_hip(shape):
    N, H, W, ch, cw = shape
    flow, c1, c2 = aug_inputs(N, H, W, ch, cw, 13)
    want = ref.flow_augmentation(flow, c1, c2, ch, cw)
    _close_flow(oracle.flow_augmentation_forward(flow, c1, c2, ch, cw), want, "oracle vs reference")
    got = ops.flow_augmentation_forward(torch.from_numpy(flow).cuda(), c1, c2, ch, cw).cpu().numpy()
    _close_flow(got, want, "hip vs reference")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(GOLD), reason="golden vectors not generated yet")
@pytest.mark.parametrize("i", range(len(MG.FLOW_AUG)))
def test_hip_flow_augmentation_layer_matches_reference_golden(i):
    from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
    gold = np.load(GOLD)
    if f"flowaug{i}" not in gold:
        pytest.skip("golden arrays for FlowAugmentation not generated")
    flow, c1, c2, ch, cw = MG.flow_aug_inputs(i)
    layer = LayerRegistry.CreateLayer(LayerParameter(type="FlowAugmentation", augmentation_param=dict(crop_width=cw, crop_height=ch)))
    bottom = [Blob.from_tensor(torch.from_numpy(x).cuda()) for x in (flow, c1.reshape(len(c1), 42, 1, 1), c2.reshape(len(c2), 42, 1, 1))]
    top = [Blob()]
    layer.SetUp(bottom, top)
    layer.Forward(bottom, top)
    _close_flow(top[0].cpu_data(), gold[f"flowaug{i}"], "layer mirror vs reference golden")


@pytest.mark.skipif(not os.path.exists(GOLD), reason="golden vectors not generated yet")
@pytest.mark.parametrize("i", range(len(MG.FLOW_AUG)))
def test_oracle_flow_augmentation_matches_reference_golden(i):
    gold = np.load(GOLD)
    if f"flowaug{i}" not in gold:
        pytest.skip("golden arrays for FlowAugmentation not generated")
    flow, c1, c2, ch, cw = MG.flow_aug_inputs(i)
    _close_flow(oracle.flow_augmentation_forward(flow, c1, c2, ch, cw), gold[f"flowaug{i}"], "oracle vs reference golden")


# ---------------------------------------------------------------------------------------------------------
# DataAugmentation (for given coefficients)
# ---------------------------------------------------------------------------------------------------------
def _close_img(a, b, what, q=2e-5, worst=0.05):
    """Image values are in [0,1]; the sampling position may differ by an ulp (contraction), which for a smooth image is invisible;
    the colour transforms go through powf, which differs by an ulp between libm and the device library."""
    err = np.abs(a - b)
    assert a.shape == b.shape and np.quantile(err, 0.999) <= q and err.max() < worst, (what, float(np.quantile(err, 0.999)), float(err.max()))


def _oracle_data_aug(i):
    img, co, mean3, ch, cw = MG.data_aug_inputs(i)
    return oracle.data_augmentation_forward(img, co, ch, cw, mean=mean3, mean_mode=1 if mean3 is not None else 0, max_multiplier=255.0,
                                            chromatic_eigvec=MG.EIGVEC)


def test_oracle_spatial_augmentation_vs_fp64_bilinear():
    img, co, mean3, ch, cw = MG.data_aug_inputs(0)
    N, C, H, W = img.shape
    got = oracle.data_augmentation_forward(img, co, ch, cw)
    ys, xs = np.mgrid[0:ch, 0:cw].astype(np.float64)
    for n in range(N):
        c = co[n]
        M = matrix64(c[0], c[1], c[2], c[3], np.exp(np.float64(c[4])), np.exp(np.float64(c[5])), cw, ch, W, H)
        xp = np.clip(M[0, 0] * xs + M[0, 1] * ys + M[0, 2], 0, W - 1.05)
        yp = np.clip(M[1, 0] * xs + M[1, 1] * ys + M[1, 2], 0, H - 1.05)
        x0, y0 = np.floor(xp).astype(int), np.floor(yp).astype(int)
        fx, fy = xp - x0, yp - y0
        for ch_ in range(C):
            p = img[n, ch_].astype(np.float64)
            want = (1 - fx) * (1 - fy) * p[y0, x0] + fx * fy * p[y0 + 1, x0 + 1] + (1 - fx) * fy * p[y0 + 1, x0] + fx * (1 - fy) * p[y0, x0 + 1]
            assert np.abs(got[n, ch_] - want).max() < 2e-4


def test_data_augmentation_invariants_and_errors():
    img, co, mean3, ch, cw = MG.data_aug_inputs(1)
    N, C, H, W = img.shape
    ident = np.zeros((N, 42), np.float32)
    # default coefficients: a centred crop (bilinear at integer + 0.0 offsets when the margins are even), no colour change
    top = oracle.data_augmentation_forward(img, ident, 32, 40)
    assert np.abs(top - img[:, :, 4:36, 8:48]).max() < 1e-6
    assert np.array_equal(oracle.data_augmentation_forward(img, None, 32, 40), top)
    # no crop size: the bottom is copied whatever the coefficients say; then the mean
    m = np.array([0.1, 0.2, 0.3], np.float32)
    out = oracle.data_augmentation_forward(img, co, 0, 0, mean=m, mean_mode=1)
    assert np.array_equal(out, img - m.reshape(1, 3, 1, 1))
    pm = np.random.default_rng(0).random((3, 32, 40)).astype(np.float32)
    assert np.array_equal(oracle.data_augmentation_forward(img, ident, 32, 40, mean=pm, mean_mode=2), top - pm[None])
    # one sample with colour coefficients drags the whole batch through the colour kernel (batch flags, .cu:456-476): the sample with
    # DEFAULT colour coefficients changes too (brightness compensation mean_in / (mean_out + 0.01), clamp to [0,1])
    with_col = oracle.data_augmentation_forward(img, co, ch, cw)
    only_spatial = co.copy(); only_spatial[:, 6:] = 0
    without = oracle.data_augmentation_forward(img, only_spatial, ch, cw)
    assert np.abs(with_col[1] - without[1]).max() > 1e-4 and np.abs(with_col[1] - without[1]).max() < 0.02
    # the noise effect (counter-based stream): only the sample with noise > 0 changes beyond what the effects' clamp does to the batch;
    # reproducible from (seed, stream), different for another stream -- distribution tests: tests/test_augmentation_random.py
    noisy = co.copy(); noisy[0, 41] = np.log(1.0) + 0.1
    n1 = oracle.data_augmentation_forward(img, noisy, ch, cw, noise_seed=5, noise_stream=1)
    assert np.array_equal(n1, oracle.data_augmentation_forward(img, noisy, ch, cw, noise_seed=5, noise_stream=1))
    n2 = oracle.data_augmentation_forward(img, noisy, ch, cw, noise_seed=5, noise_stream=2)
    assert not np.array_equal(n1[0], n2[0]) and np.array_equal(n1[1], n2[1])
    with pytest.raises(ValueError):                      # crop greater than original
        oracle.data_augmentation_forward(img, co, H + 1, W)
    with pytest.raises(ValueError):                      # colour transforms need 3 channels
        oracle.data_augmentation_forward(img[:, :2], co, ch, cw)
    with pytest.raises(ValueError, match="no CPU path"):
        ops.data_augmentation_forward(ops.data_aug_params(8, 8), torch.zeros(1, 3, 16, 16))


def test_data_augmentation_layer_mirror_checks():
    from flownet2_amd.layers import Blob, CheckError, LayerParameter, LayerRegistry
    img, co = Blob(2, 3, 16, 24, device="cpu"), Blob(2, 42, 1, 1, device="cpu")
    layer = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", augmentation_param=dict(crop_width=20, crop_height=12, mean=[1, 2, 3], mean_per_pixel=False)))
    top = [Blob(device="cpu"), Blob(device="cpu")]
    layer.SetUp([img, co], top)
    assert top[0].shape() == [2, 3, 12, 20] and top[1].shape() == [2, 42, 1, 1] and layer.mean_mode_ == ops.MEAN_PER_CHANNEL and not layer.AllowBackward()
    layer = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation"))
    top = [Blob(device="cpu")]
    layer.SetUp([img], top)
    assert top[0].shape() == [2, 3, 16, 24] and not layer.do_cropping_
    for bottoms, ntop, ap, msg in [([img, co, co], 1, {}, "one or two input blobs"), ([img], 3, {}, "one or two output blobs"),
                                   ([img], 1, dict(crop_width=30, crop_height=8), "crop width greater"), ([img], 1, dict(crop_width=8, crop_height=30), "crop height greater")]:
        with pytest.raises(CheckError, match=msg):
            LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", augmentation_param=ap)).SetUp(bottoms, [Blob(device="cpu") for _ in range(ntop)])


@pytest.mark.skipif(not os.path.exists(GOLD), reason="golden vectors not generated yet")
@pytest.mark.parametrize("i", range(len(MG.DATA_AUG)))
def test_oracle_data_augmentation_matches_reference_golden(i):
    gold = np.load(GOLD)
    if f"dataaug{i}" not in gold:
        pytest.skip("golden arrays for DataAugmentation not generated")
    _close_img(_oracle_data_aug(i), gold[f"dataaug{i}"], "oracle vs reference golden")


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(MG.DATA_AUG)))
def test_hip_data_augmentation_matches_oracle_and_reference_golden(i):
    from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
    img, co, mean3, ch, cw = MG.data_aug_inputs(i)
    ap = dict(max_multiplier=255.0, chromatic_eigvec=list(MG.EIGVEC))
    if ch:
        ap.update(crop_width=cw, crop_height=ch)
    if mean3 is not None:
        ap.update(mean=mean3.tolist(), mean_per_pixel=False)
    layer = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", augmentation_param=ap))
    bottom = [Blob.from_tensor(torch.from_numpy(img).cuda())]
    if co is not None:
        bottom.append(Blob.from_tensor(torch.from_numpy(co.reshape(len(co), 42, 1, 1)).cuda()))
    top = [Blob()]
    layer.SetUp(bottom, top)
    layer.Forward(bottom, top)
    got = top[0].cpu_data()
    _close_img(got, _oracle_data_aug(i), "hip vs oracle")
    if os.path.exists(GOLD) and f"dataaug{i}" in np.load(GOLD):
        _close_img(got, np.load(GOLD)[f"dataaug{i}"], "hip vs reference golden", q=5e-6)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_reference_data_augmentation_equals_oracle_and_hip_at_training_size():
    """A FlyingChairs batch: 8 x [3,384,512] -> 320x448 with spatial + colour + shadow coefficients."""
    rng = np.random.default_rng(77)
    N, H, W, ch, cw = 8, 384, 512, 320, 448
    img = torch.nn.functional.interpolate(torch.from_numpy(rng.random((N, 3, 13, 17)).astype(np.float32)), size=(H, W), mode="bilinear", align_corners=True).numpy()
    co = np.zeros((N, 42), np.float32)
    co[:, :6] = np.stack([[float(rng.random() < 0.5), rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-0.1, 0.1),
                           np.log(rng.uniform(1.0, 1.2)), np.log(rng.uniform(1.0, 1.2))] for _ in range(N)])
    co[:, 6:12] = rng.uniform(-0.1, 0.1, (N, 6))
    co[0, 38:41] = [1.0, 3.0, 0.2]            # (no chromatic-eigen coefficients: the reference's batch statistics are racy, see make_golden.py)
    mean3 = np.array([0.41, 0.43, 0.45], np.float32)
    want = ref.data_augmentation(img, co, ch, cw, 255.0, MG.EIGVEC, mean3)
    _close_img(oracle.data_augmentation_forward(img, co, ch, cw, mean=mean3, mean_mode=1, chromatic_eigvec=MG.EIGVEC), want, "oracle vs reference")
    p = ops.data_aug_params(cw, ch, 255.0, MG.EIGVEC, ops.MEAN_PER_CHANNEL)
    got = ops.data_augmentation_forward(p, torch.from_numpy(img).cuda(), co, torch.from_numpy(mean3).cuda()).cpu().numpy()
    _close_img(got, want, "hip vs reference", q=5e-6)


# ---------------------------------------------------------------------------------------------------------
# Drawing coefficients (host logic; the random stream itself is numpy's, not boost's: see flownet2_amd/augment.py)
# ---------------------------------------------------------------------------------------------------------
TRAIN_AUG = dict(   # the shape of the FlowNet2 training prototxts' augmentation_param (memory): first image, then the relative one
    mirror=dict(rand_type="bernoulli", prob=0.5), translate=dict(rand_type="uniform_bernoulli", exp=False, mean=0, spread=0.4, prob=1.0),
    rotate=dict(rand_type="uniform_bernoulli", exp=False, mean=0, spread=0.4, prob=1.0), zoom=dict(rand_type="uniform_bernoulli", exp=True, mean=0.2, spread=0.4, prob=1.0),
    squeeze=dict(rand_type="uniform_bernoulli", exp=True, mean=0, spread=0.3, prob=1.0), gamma=dict(rand_type="uniform_bernoulli", exp=True, mean=0, spread=0.02, prob=1.0),
    brightness=dict(rand_type="gaussian_bernoulli", exp=False, mean=0, spread=0.02, prob=1.0), contrast=dict(rand_type="uniform_bernoulli", exp=True, mean=0, spread=0.4, prob=1.0),
    color=dict(rand_type="gaussian_bernoulli", exp=True, mean=0, spread=0.02, prob=1.0), lmult_pow=dict(rand_type="uniform_bernoulli", exp=True, mean=-0.2, spread=0.4, prob=1.0),
    col_rotate=dict(rand_type="uniform_bernoulli", exp=False, mean=0, spread=1.0, prob=1.0))
REL_AUG = dict(translate=dict(rand_type="gaussian_bernoulli", exp=False, mean=0, spread=0.03, prob=1.0), rotate=dict(rand_type="gaussian_bernoulli", exp=False, mean=0, spread=0.03, prob=1.0),
               zoom=dict(rand_type="gaussian_bernoulli", exp=True, mean=0, spread=0.03, prob=1.0), gamma=dict(rand_type="gaussian_bernoulli", exp=True, mean=0, spread=0.02, prob=1.0))


def test_coefficient_arrays_round_trip_and_compose_in_the_log_domain():
    from flownet2_amd import augment as A
    assert A.NUM_PARAMS == ops.AUG_NUM_PARAMS == 42
    c = A.default_coeff()
    assert np.array_equal(A.coeff_to_array(c), np.zeros(42, np.float32))           # defaults: 0, or log 1
    c.update(mirror=1.0, dx=0.05, angle=-0.2, zoom_x=1.3, zoom_y=0.8, gamma=1.1, col_angle=0.3, shadow_strength=0.2)
    arr = A.coeff_to_array(c)
    assert np.array_equal(arr[:6], coeff_array(1.0, 0.05, 0.0, -0.2, 1.3, 0.8)[:6])   # the layout the pinned C functions read
    back = A.array_to_coeff(arr)
    assert all(abs(back[k] - c[k]) < 1e-6 for k in c)
    # add_coeff_to_array sums arrays: additive fields add, multiplicative (log-stored) fields multiply
    out = arr.copy()
    A.add_coeff_to_array(dict(A.default_coeff(), dx=0.01, zoom_x=1.1, gamma=0.9), out)
    z = A.array_to_coeff(out)
    assert abs(z["dx"] - 0.06) < 1e-6 and abs(z["zoom_x"] - 1.43) < 1e-5 and abs(z["gamma"] - 0.99) < 1e-5 and z["mirror"] == 1.0


def test_drawn_spatial_coefficients_keep_the_crop_inside_the_image():
    """generate_valid_spatial_coeffs (augmentation_layer_base.cpp:101-169): every accepted draw maps the four crop corners into the
    image -- checked with the PINNED matrix helper, not with the generator's own test."""
    from flownet2_amd import augment as A
    rng = np.random.default_rng(3)
    W, H, cw, ch = 512, 384, 448, 320
    first = A.draw_batch(rng, TRAIN_AUG, 64, W, H, cw, ch, discount=A.discount_coeff(10 ** 6))
    second = A.draw_batch(rng, REL_AUG, 64, W, H, cw, ch, in_params=first, mode="add")
    assert first.shape == second.shape == (64, 42) and np.isfinite(first).all() and np.isfinite(second).all()
    assert 10 < first[:, 0].sum() < 54                                                    # mirror: Bernoulli(0.5)
    assert np.array_equal(second[:, 0], first[:, 0])                                      # the relative draw has no mirror: it is inherited
    assert np.abs(second[:, 1:6] - first[:, 1:6]).max() < 0.2 and np.abs(second[:, 1:6] - first[:, 1:6]).max() > 0
    for blob in (first, second):
        for n in range(64):
            if np.array_equal(blob[n, :6], np.zeros(6)):                                  # 50 failed tries: falls back to the incoming values
                continue
            m = oracle.augmentation_matrix(blob[n], cw, ch, W, H)
            for x in (0, cw - 1):
                for y in (0, ch - 1):
                    xs, ys = m[0] * x + m[2] * y + m[4], m[1] * x + m[3] * y + m[5]
                    assert -1e-3 <= xs <= W - 1 + 1e-3 and -1e-3 <= ys <= H - 1 + 1e-3, (n, x, y, xs, ys)
    # schedule: spread grows from 0 with the iteration count (data_augmentation_layer.cu:366-368)
    assert A.discount_coeff(0, dict(half_life=50000, initial_coeff=0.5, final_coeff=1)) == 0.5
    assert abs(A.discount_coeff(50000, dict(half_life=50000, initial_coeff=0.5, final_coeff=1)) - 0.75) < 1e-3
    zero = A.draw_batch(rng, dict(translate=dict(rand_type="uniform", spread=0.3)), 4, W, H, cw, ch, discount=0.0)
    assert np.array_equal(zero, np.zeros((4, 42), np.float32))


def test_augmentation_layer_mirrors_draw_only_in_the_training_phase():
    from flownet2_amd.layers import Blob, CheckError, LayerParameter, LayerRegistry
    img, aug_img = Blob(4, 3, 384, 512, device="cpu"), Blob(4, 3, 320, 448, device="cpu")
    ap = dict(TRAIN_AUG, crop_width=448, crop_height=320, seed=7)
    # DataAugmentation draws its own coefficients in TRAIN, keeps the defaults in TEST (data_augmentation_layer.cu:375-387)
    for phase, drawn in (("TEST", False), ("TRAIN", True)):
        layer = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", phase=phase, augmentation_param=ap,
                                                         coeff_schedule_param=dict(half_life=50000, initial_coeff=0.5, final_coeff=1.0)))
        layer.SetUp([img], [Blob(device="cpu"), Blob(device="cpu")])
        layer.num_iter_ = 1
        co = layer._draw([img])
        assert (co is not None) == drawn
        if drawn:
            assert co.shape == (4, 42) and np.isfinite(co).all() and np.abs(co[:, 1:12]).max() > 0 and np.array_equal(co[:, 34:], np.zeros((4, 8), np.float32))
    # GenerateAugmentationParameters, three bottoms, mode add: relative coefficients on top of the given ones
    params0 = Blob.from_tensor(torch.from_numpy(co).view(4, 42, 1, 1))
    gen = LayerRegistry.CreateLayer(LayerParameter(type="GenerateAugmentationParameters", phase="TRAIN", augmentation_param=dict(REL_AUG, mode="add", seed=9)))
    top = [Blob(device="cpu")]
    gen.SetUp([params0, img, aug_img], top)
    assert top[0].shape() == [4, 42, 1, 1] and (gen.cropped_width_, gen.cropped_height_, gen.bottomwidth_, gen.bottomheight_) == (448, 320, 512, 384)
    gen.Forward_gpu([params0, img, aug_img], top)
    out = top[0].data.numpy().reshape(4, 42)
    assert np.array_equal(out[:, 0], co[:, 0]) and 0 < np.abs(out[:, 1:6] - co[:, 1:6]).max() < 0.3
    assert np.abs(out[:, 7:12] - co[:, 7:12]).max() < 1e-6 and np.abs(out[:, 6] - co[:, 6]).max() > 0          # only gamma is re-drawn
    # outside the training phase nothing is drawn: the coefficients pass through (exp / log round trip of the multiplicative fields)
    gen_test = LayerRegistry.CreateLayer(LayerParameter(type="GenerateAugmentationParameters", phase="TEST", augmentation_param=dict(REL_AUG, mode="add")))
    gen_test.SetUp([params0, img, aug_img], top)
    gen_test.Forward_gpu([params0, img, aug_img], top)
    assert np.abs(top[0].data.numpy().reshape(4, 42) - co).max() < 1e-6
    # one image bottom: mode forced to "regenerate", sizes from the blob and the prototxt
    gen1 = LayerRegistry.CreateLayer(LayerParameter(type="GenerateAugmentationParameters", phase="TRAIN", augmentation_param=dict(TRAIN_AUG, crop_width=448, crop_height=320)))
    gen1.SetUp([img], top)
    assert gen1.mode_ == "regenerate" and (gen1.bottomwidth_, gen1.bottomheight_) == (512, 384)
    gen1.Forward_gpu([img], top)
    assert np.isfinite(top[0].data.numpy()).all()
    for bottoms, apx, msg in [([img, img], dict(crop_width=4, crop_height=4), "takes one .* or three"), ([params0], dict(crop_width=4, crop_height=4), "Need bottomwidth"),
                              ([img], {}, "Need crop_width")]:
        with pytest.raises(CheckError, match=msg):
            LayerRegistry.CreateLayer(LayerParameter(type="GenerateAugmentationParameters", augmentation_param=apx)).SetUp(bottoms, [Blob(device="cpu")])


def test_rng_generate_follows_caffe_rng_generate_semantics():
    """rng.cpp:8-114: spread scaled by the schedule only with apply_schedule, zero spread returns the mean, exp after the draw,
    *_bernoulli returns prob0_value untouched when the coin fails, discretize rounds, multiplier applies last."""
    from flownet2_amd import augment as A
    rng = np.random.default_rng(11)
    g = lambda p, **k: A.rng_generate(rng, p, **k)
    xs = np.array([g(dict(rand_type="uniform", mean=2.0, spread=0.5)) for _ in range(4000)])
    assert 1.5 <= xs.min() < 1.55 and 2.45 < xs.max() <= 2.5 and abs(xs.mean() - 2.0) < 0.02
    xs = np.array([g(dict(rand_type="uniform", mean=2.0, spread=0.5), discount=0.1) for _ in range(500)])
    assert 1.95 <= xs.min() and xs.max() <= 2.05
    xs = np.array([g(dict(rand_type="uniform", mean=2.0, spread=0.5, apply_schedule=False), discount=0.1) for _ in range(2000)])
    assert xs.min() < 1.6 and xs.max() > 2.4
    assert g(dict(rand_type="gaussian", mean=0.3, spread=0.0)) == np.float32(0.3) and g(dict(rand_type="uniform", mean=0.0, spread=0.0, exp=True)) == 1.0
    xs = np.array([g(dict(rand_type="gaussian", mean=-1.0, spread=0.2)) for _ in range(4000)])
    assert abs(xs.mean() + 1.0) < 0.02 and abs(xs.std() - 0.2) < 0.02
    xs = np.array([g(dict(rand_type="gaussian", mean=0.0, spread=0.3, exp=True)) for _ in range(4000)])
    assert xs.min() > 0 and abs(np.log(xs).mean()) < 0.03
    bs = [g(dict(rand_type="bernoulli", prob=0.25), as_bool=True) for _ in range(4000)]
    assert all(isinstance(b, bool) for b in bs) and abs(np.mean(bs) - 0.25) < 0.03
    assert g(dict(rand_type="bernoulli", prob=0.0)) == 0.0
    # the coin of *_bernoulli: failing returns prob0_value as is (no exp, no multiplier); without one it continues from 0
    assert g(dict(rand_type="uniform_bernoulli", mean=5.0, spread=1.0, prob=0.0, exp=True, multiplier=3.0), prob0_value=1.0) == 1.0
    assert g(dict(rand_type="gaussian_bernoulli", mean=5.0, spread=1.0, prob=0.0, exp=True, multiplier=3.0)) == 3.0       # exp(0) * 3
    xs = np.array([g(dict(rand_type="uniform_bernoulli", mean=5.0, spread=1.0, prob=0.5), prob0_value=0.0) for _ in range(2000)])
    assert abs((xs == 0).mean() - 0.5) < 0.05 and xs[xs != 0].min() >= 4.0
    xs = np.array([g(dict(rand_type="uniform", mean=0.0, spread=3.0, discretize=True, multiplier=0.5)) for _ in range(500)])
    assert np.all(np.abs(xs * 2 - np.round(xs * 2)) < 1e-6) and set(np.unique(xs)) <= {-1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5}
    with pytest.raises(ValueError, match="Unknown random type"):
        g(dict(rand_type="poisson"))
