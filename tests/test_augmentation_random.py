"""Random side of the training augmentation: the reference's boost / cuRAND STREAMS cannot be reproduced, so the
DISTRIBUTIONS are pinned -- Kolmogorov-Smirnov tests of caffe_rng_generate's restatement (src/caffe/util/rng.cpp:8-114) against the
closed forms for every rand_type / exp / discretize / schedule combination, of the counter-based generator itself (Philox4x32-10:
Random123's known-answer vectors), of the device-side noise effect (data_augmentation_layer.cu:578-587) and of the drawn spatial
coefficients; plus the layer state of `recompute_mean` (.cu:593-621) and the prefetch thread."""
import math

import numpy as np
import pytest
import torch
from scipy import stats

import oracle
from flownet2_amd import augment as A

N = 20000
P_MIN = 1e-4          # a correct sampler fails a test at this level once in 10,000 runs; the seeds are fixed


def draws(param, n=N, seed=1, **kw):
    rng = A.make_rng(seed, 0)
    return np.array([A.rng_generate(rng, param, **kw) for _ in range(n)], np.float64)


def test_philox_known_answers_and_stream_layout():
    # Random123's kat_vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        assert tuple(oracle.philox4x32_10(c, k)) == want
        assert tuple(int(v) for v in A.PhiloxStream.words([c[0]], [c[1]], [c[2]], [c[3]], k[0], k[1])[0]) == want
    # a stream = key (seed), counter (block, 0, stream): the first words are block 0's
    r = A.PhiloxStream(seed=(7 << 32) | 5, stream=(2 << 32) | 9)
    first = [r._word() for _ in range(8)]
    assert first[:4] == oracle.philox4x32_10((0, 0, 9, 2), (5, 7)) and first[4:] == oracle.philox4x32_10((1, 0, 9, 2), (5, 7))
    # reproducible, and independent of how many draws other streams made
    a = [A.make_rng(3, 10).normal() for _ in range(3)]
    assert a == [A.make_rng(3, 10).normal() for _ in range(3)] and a != [A.make_rng(3, 11).normal() for _ in range(3)]


def test_uniform_and_gaussian_base_draws_pass_ks():
    r = A.make_rng(11, 0)
    assert stats.kstest([r.random() for _ in range(N)], "uniform").pvalue > P_MIN
    assert stats.kstest([r.uniform(-2.0, 5.0) for _ in range(N)], stats.uniform(-2.0, 7.0).cdf).pvalue > P_MIN
    z = np.array([r.normal(1.5, 0.25) for _ in range(N)])
    assert stats.kstest(z, stats.norm(1.5, 0.25).cdf).pvalue > P_MIN
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 0.03                     # Box-Muller pairs are independent


@pytest.mark.parametrize("param,cdf", [
    (dict(rand_type="uniform", mean=0.3, spread=0.2), stats.uniform(0.1, 0.4).cdf),
    (dict(rand_type="gaussian", mean=-1.0, spread=0.5), stats.norm(-1.0, 0.5).cdf),
    (dict(rand_type="uniform", mean=0.0, spread=0.4, exp=True), stats.loguniform(math.exp(-0.4), math.exp(0.4)).cdf),
    (dict(rand_type="gaussian", mean=0.1, spread=0.3, exp=True), stats.lognorm(0.3, scale=math.exp(0.1)).cdf),
    (dict(rand_type="uniform", mean=1.0, spread=2.0, multiplier=3.0), stats.uniform(-3.0, 12.0).cdf)])
def test_continuous_rand_types_follow_their_closed_form(param, cdf):
    assert stats.kstest(draws(param), cdf).pvalue > P_MIN


def test_schedule_discount_scales_the_spread_and_apply_schedule_false_does_not():
    p = dict(rand_type="gaussian", mean=0.0, spread=1.0)
    assert stats.kstest(draws(p, discount=0.25), stats.norm(0, 0.25).cdf).pvalue > P_MIN
    assert stats.kstest(draws(dict(p, apply_schedule=False), discount=0.25), stats.norm(0, 1.0).cdf).pvalue > P_MIN
    assert np.all(draws(dict(rand_type="uniform", mean=0.7, spread=0.0), n=50) == np.float32(0.7))        # spread 0: the mean, no draw


def test_bernoulli_and_mixture_types():
    b = draws(dict(rand_type="bernoulli", prob=0.3))
    assert set(np.unique(b)) == {0.0, 1.0} and abs(b.mean() - 0.3) < 4 * math.sqrt(0.3 * 0.7 / N)
    # uniform_bernoulli: with probability prob a uniform draw, else prob0_value (returned as is) or 0
    m = draws(dict(rand_type="uniform_bernoulli", prob=0.6, mean=2.0, spread=1.0), prob0_value=-7.0)
    off = m == -7.0
    assert abs(off.mean() - 0.4) < 4 * math.sqrt(0.24 / N)
    assert stats.kstest(m[~off], stats.uniform(1.0, 2.0).cdf).pvalue > P_MIN
    g = draws(dict(rand_type="gaussian_bernoulli", prob=0.5, mean=0.0, spread=0.2, exp=True))             # no prob0_value: exp(0) = 1 when off
    off = g == 1.0
    assert abs(off.mean() - 0.5) < 4 * math.sqrt(0.25 / N)
    assert stats.kstest(np.log(g[~off]), stats.norm(0, 0.2).cdf).pvalue > P_MIN
    d = draws(dict(rand_type="uniform", mean=0.0, spread=2.49, discretize=True))
    vals, counts = np.unique(d, return_counts=True)
    assert list(vals) == [-2.0, -1.0, 0.0, 1.0, 2.0]
    expect = np.array([0.99, 1.0, 1.0, 1.0, 0.99]) / 4.98 * N                                               # round() of U(-2.49, 2.49)
    assert stats.chisquare(counts, expect).pvalue > P_MIN
    as_bool = [A.rng_generate(A.make_rng(2, i), dict(rand_type="bernoulli", prob=0.5), as_bool=True) for i in range(50)]
    assert set(as_bool) == {True, False}


def test_drawn_spatial_coefficients_are_valid_and_distributed_as_configured():
    aug = dict(mirror=dict(rand_type="bernoulli", prob=0.5), translate=dict(rand_type="uniform_bernoulli", prob=1.0, mean=0.0, spread=0.05),
               rotate=dict(rand_type="uniform_bernoulli", prob=1.0, mean=0.0, spread=0.1), zoom=dict(rand_type="uniform_bernoulli", prob=1.0, mean=0.1, spread=0.2, exp=True))
    blob = np.concatenate([A.draw_batch(A.make_rng(5, it), aug, 8, 512, 384, 448, 320) for it in range(400)])
    co = [A.array_to_coeff(r) for r in blob]
    assert all(A.corners_inside(c, 512, 384, 448, 320) for c in co)                      # the rejection loop of generate_valid_spatial_coeffs
    assert abs(np.mean([c["mirror"] for c in co]) - 0.5) < 0.05
    ang = np.array([c["angle"] for c in co])
    assert ang.min() >= -0.1001 and ang.max() <= 0.1001 and abs(ang.mean()) < 0.01      # (rejection favours small angles: not uniform any more)
    z = np.array([c["zoom_x"] for c in co])
    assert z.min() >= math.exp(-0.1) - 1e-4 and np.allclose(z, [c["zoom_y"] for c in co])
    # a pure function of (seed, iteration): what the prefetch thread relies on
    assert np.array_equal(A.draw_batch(A.make_rng(5, 17), aug, 8, 512, 384, 448, 320), blob[17 * 8:18 * 8])


def test_prefetcher_hands_out_iterations_in_order_and_surfaces_errors():
    aug = dict(translate=dict(rand_type="uniform_bernoulli", prob=1.0, mean=0.0, spread=0.02))
    draw = lambda it: A.draw_batch(A.make_rng(9, it), aug, 4, 64, 48, 56, 40, A.discount_coeff(it, dict(half_life=10.0, initial_coeff=0.5, final_coeff=1.0)))
    pre = A.CoefficientPrefetcher(draw, depth=3)
    try:
        for it in range(8):
            assert np.array_equal(pre.get(), draw(it))
    finally:
        pre.close()

    def bad(it):
        if it == 2:
            raise RuntimeError("boom")
        return it
    pre = A.CoefficientPrefetcher(bad, depth=2)
    assert pre.get() == 0 and pre.get() == 1
    with pytest.raises(RuntimeError, match="boom"):
        pre.get()
    pre.close()


def _draw_for_process(it, seed):                 # module level: the worker process pickles it
    aug = dict(translate=dict(rand_type="uniform_bernoulli", prob=1.0, mean=0.0, spread=0.02))
    return A.draw_batch(A.make_rng(seed, it), aug, 4, 64, 48, 56, 40, 1.0)


def _bad_for_process(it):
    if it == 1:
        raise RuntimeError("boom")
    return it


def test_prefetcher_in_a_worker_process():
    """process=True: the draws run in a spawned interpreter (no competition for the consumer's interpreter lock); same order, same
    values, errors surface in get()."""
    import functools
    pre = A.CoefficientPrefetcher(functools.partial(_draw_for_process, seed=9), depth=3, process=True)
    try:
        for it in range(6):
            assert np.array_equal(pre.get(), _draw_for_process(it, 9))
    finally:
        pre.close()
    pre = A.CoefficientPrefetcher(_bad_for_process, depth=2, process=True)
    try:
        assert pre.get() == 0
        with pytest.raises(RuntimeError, match="boom"):
            pre.get()
    finally:
        pre.close()


def test_oracle_noise_effect_is_gaussian_with_the_samples_sigma():
    img = np.full((2, 3, 96, 128), 0.5, np.float32)
    co = np.zeros((2, 42), np.float32)
    co[0, 41], co[1, 41] = 0.05, 0.0                                  # noise: default 0 -> stored as is (no log)
    out = oracle.data_augmentation_forward(img, co, 96, 128, max_multiplier=1.0, noise_seed=123, noise_stream=4)
    d = (out[0] - 0.5).astype(np.float64)
    assert stats.kstest(d.ravel(), stats.norm(0, 0.05).cdf).pvalue > P_MIN
    assert np.array_equal(out[1], img[1])                              # the sample without noise is untouched
    assert abs(np.corrcoef(d[0].ravel(), d[1].ravel())[0, 1]) < 0.02 and abs(np.corrcoef(d[0, :, :-1].ravel(), d[0, :, 1:].ravel())[0, 1]) < 0.02


@pytest.mark.gpu
def test_device_noise_effect_matches_the_twin_and_is_gaussian():
    from flownet2_amd import ops
    img = np.random.default_rng(3).random((3, 3, 80, 112)).astype(np.float32) * 0.5 + 0.25
    co = np.zeros((3, 42), np.float32)
    co[:, 41] = [0.02, 0.0, 0.1]
    p = ops.data_aug_params(112, 80, 4.0, None, ops.MEAN_NONE, noise_seed=(9 << 32) | 77, noise_stream=12345)
    got = ops.data_augmentation_forward(p, torch.from_numpy(img).cuda(), co).cpu().numpy()
    ref = oracle.data_augmentation_forward(img, co, 80, 112, max_multiplier=4.0, noise_seed=(9 << 32) | 77, noise_stream=12345)
    assert np.abs(got - ref).max() <= 2e-6 * (1 + np.abs(ref).max()) + 1e-5 * 0.1      # same words; logf / cosf differ in the last ulps
    co0 = co.copy(); co0[:, 41] = 0
    co0[0, 38:41] = [0.3, 1e6, 1e-6]              # a shadow nobody sees keeps the effects pass (and its clamp) on for the whole batch
    clean = ops.data_augmentation_forward(p, torch.from_numpy(img).cuda(), co0).cpu().numpy()
    assert np.array_equal(got[1], clean[1])                                              # the sample without noise is untouched by it
    for n, sigma in ((0, 0.02), (2, 0.1)):
        assert stats.kstest((got[n] - clean[n]).astype(np.float64).ravel(), stats.norm(0, sigma).cdf).pvalue > P_MIN
    again = ops.data_augmentation_forward(p, torch.from_numpy(img).cuda(), co).cpu().numpy()
    assert np.array_equal(got, again)
    p2 = ops.data_aug_params(112, 80, 4.0, None, ops.MEAN_NONE, noise_seed=(9 << 32) | 77, noise_stream=12346)
    other = ops.data_augmentation_forward(p2, torch.from_numpy(img).cuda(), co).cpu().numpy()
    assert not np.array_equal(got[0], other[0]) and abs(np.corrcoef((got[0] - clean[0]).ravel(), (other[0] - clean[0]).ravel())[0, 1]) < 0.03


@pytest.mark.gpu
def test_two_augmentation_layers_add_independent_noise():
    """The reference draws the noise of every layer instance from the process-wide generator (data_augmentation_layer.cu:578-587): the
    two frames of a pair (img0s_aug / img1s_aug, identical default parameters, same iteration) must not receive the same noise field."""
    from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
    img = np.full((2, 3, 64, 96), 0.5, np.float32)
    co = np.zeros((2, 42, 1, 1), np.float32)
    co[:, 41] = 0.05
    fields = []
    for name in ("img0s_aug", "img1s_aug", "img0s_aug"):
        layer = LayerRegistry.CreateLayer(LayerParameter(name=name, type="DataAugmentation", phase="TRAIN",
                                                         augmentation_param=dict(crop_width=96, crop_height=64, max_multiplier=1.0)))
        bottom, top = [Blob(2, 3, 64, 96), Blob(2, 42, 1, 1)], [Blob()]
        bottom[0].data, bottom[1].data = torch.from_numpy(img).cuda(), torch.from_numpy(co)
        layer.SetUp(bottom, top)
        layer.Forward(bottom, top)
        fields.append((top[0].data.cpu().numpy() - 0.5).astype(np.float64))
    assert fields[0].std() > 0.04 and fields[1].std() > 0.04
    assert abs(np.corrcoef(fields[0].ravel(), fields[1].ravel())[0, 1]) < 0.03          # two layers: independent
    assert np.array_equal(fields[0], fields[2])                                         # the same layer name and iteration: reproducible


@pytest.mark.gpu
def test_recompute_mean_is_layer_state_like_the_reference():
    """data_augmentation_layer.cu:593-621: over the first `recompute_mean` iterations the per-pixel mean is the running average of the
    batch means of the augmented images, the per-channel mean its average over the area; afterwards it is frozen; every iteration
    subtracts the current mean."""
    from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
    rng = np.random.default_rng(8)
    batches = [rng.random((4, 3, 24, 32)).astype(np.float32) for _ in range(5)]
    for per_pixel in (True, False):
        layer = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", phase="TRAIN",
                                                         augmentation_param=dict(crop_width=32, crop_height=24, recompute_mean=3, mean_per_pixel=per_pixel)))
        bottom, top = [Blob(4, 3, 24, 32)], [Blob()]
        layer.SetUp(bottom, top)
        mean = np.zeros((3, 24, 32), np.float64)
        for it, b in enumerate(batches, 1):
            bottom[0].data = torch.from_numpy(b).cuda()
            layer.Forward(bottom, top)
            aug = oracle.data_augmentation_forward(b, None, 24, 32)       # the centre crop itself (its last row / column are the edge clamp's)
            if it <= 3:
                mean = (mean * (it - 1) + aug.astype(np.float64).mean(0)) / it
            want = aug - (mean[None] if per_pixel else mean.mean((1, 2)).reshape(1, 3, 1, 1))
            np.testing.assert_allclose(top[0].data.cpu().numpy(), want, atol=2e-6)
        np.testing.assert_allclose(layer.mean_channel_.cpu().numpy(), mean.mean((1, 2)), atol=1e-6)
