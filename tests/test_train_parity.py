"""BASELINE config 4 at its own size: one FlowNetC training step at batch 8 @448x320 with PRODUCTION routing (own forward kernels inside
autograd, own weight-gradient / transposed-convolution / Winograd data-gradient kernels, fused bias + ReLU backward, flow-head
backward kernels, correlation backward) against the pinned fp64 comparator oracle/fp64_graph.py: loss and EVERY parameter gradient.

The comparator itself is pinned on the CPU (no GPU): its correlation against the independent fp64 re-derivation of tests/ref_torch64.py
incl. autograd gradients, and the whole fp64 graph against the C oracle's forward / backward restatements of the reference kernels
(the differentiable CPU backend of tests/test_parallel.py) at a size the host finishes in seconds."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import fp64_graph  # noqa: E402


def _batch(N, H, W, seed, nan_frac=0.05):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (N, 3, H, W)).astype(np.float32)
    b = np.clip(np.roll(a, (3, -5), (2, 3)) + rng.normal(0, 2, a.shape), 0, 255).astype(np.float32)
    gt = (rng.standard_normal((N, 2, H, W)) * 5).astype(np.float32)
    gt[np.broadcast_to(rng.random((N, 1, H, W)) < nan_frac, gt.shape)] = np.nan            # config 4: 5 % of the pixels without ground truth
    return torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(gt)


def test_fp64_correlation_matches_the_independent_rederivation():
    import ref_torch64
    g = torch.Generator().manual_seed(0)
    for (N, C, H, W, pad, md, s2) in [(2, 5, 9, 11, 4, 4, 2), (1, 3, 8, 8, 3, 3, 1), (1, 4, 10, 12, 6, 4, 2)]:
        b0 = torch.randn(N, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        b1 = torch.randn(N, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        top = fp64_graph._Corr64.apply(b0, b1, pad, md, s2)
        ref = ref_torch64.correlation(b0, b1, pad, 1, md, 1, s2)
        assert top.shape == ref.shape and float((top - ref).abs().max()) < 1e-13
        w = torch.randn(top.shape, generator=g, dtype=torch.float64)
        g0, g1 = torch.autograd.grad((top * w).sum(), (b0, b1))
        r0, r1 = torch.autograd.grad((ref * w).sum(), (b0, b1))
        assert float((g0 - r0).abs().max()) < 1e-13 and float((g1 - r1).abs().max()) < 1e-13


def test_fp64_graph_is_pinned_by_the_oracle_restatements():
    """The fp64 comparator vs the fp32 CPU graph built from the C oracle's restatements of the reference's forward AND backward kernels
    (correlation_layer.cu:45-249, l1loss_layer.cu:67-190, downsample_layer.cu:15-72) + torch-CPU fp32 convolutions: same loss, every
    parameter gradient within fp32 rounding (the fp32 side's; measured 1e-6 .. 2e-5 per parameter)."""
    from test_parallel import _cpu_train_backend
    from flownet2_amd import nets
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    P = nets.init_params("C", seed=3)
    a, b, gt = _batch(1, 128, 128, 11)
    loss64, g64 = fp64_graph.flownetc_train_reference(P, a, b, gt, device="cpu")
    P32 = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    be = _cpu_train_backend()
    pre = [(im * (1.0 / 255.0)) - 0.43 for im in (a, b)]
    with fp64_graph.record_relu_branches() as rec:
        loss32 = nets.multiscale_loss(nets.flownet_c_core(P32, pre[0], pre[1], be), gt, be)
    loss32.backward()
    assert abs(float(loss32.detach()) - loss64) <= 2e-6 * max(1.0, abs(loss64))
    agree = fp64_graph.grad_agreement({k: v.grad for k, v in P32.items()}, g64)
    assert set(g64) == set(P) and agree["all"] <= 2e-5 and agree["worst"] <= 2e-4, (agree["all"], agree["worst_name"], agree["worst"])
    # the same on the piecewise-linear branch the fp32 run took (16 leaky ReLUs, recorded in execution order): rounding only
    assert [n for n, _ in rec.branches] == ["conv1", "conv2", "conv3", "conv_redir", "relu", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1",
                                            "conv6", "conv6_1", "deconv5", "deconv4", "deconv3", "deconv2"]
    loss_p, g_p = fp64_graph.flownetc_train_reference(P, a, b, gt, device="cpu", masks=rec.branches)
    pinned = fp64_graph.grad_agreement({k: v.grad for k, v in P32.items()}, g_p)
    assert abs(float(loss32.detach()) - loss_p) <= 2e-6 * max(1.0, abs(loss_p))
    assert pinned["all"] <= min(1e-5, agree["all"] * 1.01) and pinned["worst"] <= 5e-5, (pinned["all"], agree["all"], pinned["worst_name"], pinned["worst"])


@pytest.mark.gpu
def test_flownetc_training_step_at_config4_size_matches_fp64():
    """Batch 8 @448x320, production routing (what `bench.py --mode train` executes) against the fp64 comparator.
    (1) On the piecewise-linear branch the fp32 run took (oracle/fp64_graph.record_relu_branches: the same sign for every leaky ReLU):
        loss within 1e-6 relative, EVERY one of the 48 parameter gradients within 1e-5 in relative L2 (measured: worst 2.1e-6, conv1's
        weights -- the library's fp32 kernels: 2.1e-6), all 39 M values together within 3e-6 (measured 7.2e-7; library 7.4e-7).
    (2) Against the plain fp64 graph (its own ReLU signs) the few units whose pre-activation is within rounding of zero dominate: reported
        next to the same figure for the library's fp32 kernels (MIOpen / rocBLAS through torch), bounded by 2e-3 per parameter and by twice the
        library's error over all values.
    (3) No gradient of the step is computed by a library kernel."""
    from flownet2_amd import functional as Fn, nets
    dev = torch.device("cuda:0")
    P = nets.init_params("C", seed=0)
    a, b, gt = _batch(8, 320, 448, 4)
    Pd = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}

    def run():
        for v in Pd.values():
            v.grad = None
        pre = [(im.to(dev) * (1.0 / 255.0)) - 0.43 for im in (a, b)]
        loss = nets.multiscale_loss(nets.flownet_c_core(Pd, pre[0], pre[1], Fn), gt.to(dev), Fn)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach())
    fallbacks = Fn.LIBRARY_FALLBACKS[0]
    run()                                                   # first use: the kernels time their tile variants (same bits whichever wins)
    assert Fn.LIBRARY_FALLBACKS[0] == fallbacks, "a layer of the training step left the own kernels"
    os.environ["FN2_TRACE_BWD"] = "1"
    try:
        import contextlib, io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), fp64_graph.record_relu_branches() as rec:
            loss = run()
        routes = [l for l in buf.getvalue().splitlines() if l.startswith("bwd on the library")]
    finally:
        os.environ.pop("FN2_TRACE_BWD", None)
    grads = {k: v.grad.detach().clone() for k, v in Pd.items()}
    assert len(rec.branches) == 16 and [n for n, _ in rec.branches][:6] == ["conv1", "conv2", "conv3", "conv_redir", "relu", "conv3_1"]
    loss_p, g_p = fp64_graph.flownetc_train_reference(P, a, b, gt, device=dev, masks=rec.branches)
    loss64, g64 = fp64_graph.flownetc_train_reference(P, a, b, gt, device=dev)
    assert set(grads) == set(g64) == set(g_p) and len(g64) == len(P)
    pinned, plain = fp64_graph.grad_agreement(grads, g_p), fp64_graph.grad_agreement(grads, g64)
    # the yardstick for the plain comparison: the same graph with every convolution, deconvolution, activation and their gradients done by the
    # LIBRARY in fp32 -- the comparator's own torch graph in float32 (oracle/fp64_graph.py, dtype: none of the product's kernels; the product
    # carries no second backend to switch to), against both comparators
    with fp64_graph.record_relu_branches() as rec_lib:
        _, g_lib = fp64_graph.flownetc_train_reference(P, a, b, gt, device=dev, dtype=torch.float32)
    lib_plain = fp64_graph.grad_agreement(g_lib, g64)
    lib_pinned = fp64_graph.grad_agreement(g_lib, fp64_graph.flownetc_train_reference(P, a, b, gt, device=dev, masks=rec_lib.branches)[1])
    flips, _units = fp64_graph.relu_sign_flips(rec.branches, rec_lib.branches)
    lines = ["relative L2 error of the parameter gradients, FlowNetC training step, batch 8 @448x320 (tests/test_train_parity.py)",
             "parameter                     own | same-branch fp64   own | plain fp64   library fp32 | same-branch   library fp32 | plain"]
    for k in sorted(pinned["per_param"], key=lambda q: -pinned["per_param"][q]):
        lines.append("%-28s  %.2e               %.2e         %.2e                   %.2e" %
                     (k, pinned["per_param"][k], plain["per_param"][k], lib_pinned["per_param"][k], lib_plain["per_param"][k]))
    lines.append("all parameters together       %.2e               %.2e         %.2e                   %.2e" % (pinned["all"], plain["all"], lib_pinned["all"], lib_plain["all"]))
    lines.append("median parameter              %.2e               %.2e         %.2e                   %.2e" % (pinned["median"], plain["median"], lib_pinned["median"], lib_plain["median"]))
    lines.append("loss: own %.9g, same-branch fp64 %.9g, plain fp64 %.9g; ReLU units on different sides in the own and the library run: %d of %d; "
                 "library backward calls of the own path: %d" % (loss, loss_p, loss64, flips, sum(m.numel() for _, m in rec.branches), len(routes)))
    report = "\n".join(lines)
    print(report)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "train_parity_config4.txt"), "w").write(report + "\n")
    # (1) rounding only
    assert abs(loss - loss_p) <= 1e-6 * max(1.0, abs(loss_p)), (loss, loss_p)
    bad = {k: v for k, v in pinned["per_param"].items() if v > 1e-5}
    assert not bad, bad
    assert pinned["all"] <= 3e-6 and pinned["median"] <= 3e-6, (pinned["all"], pinned["median"])
    # (2) the plain fp64 graph
    assert abs(loss - loss64) <= 1e-5 * max(1.0, abs(loss64)), (loss, loss64)
    assert plain["worst"] <= 2e-3 and plain["all"] <= 2.0 * lib_plain["all"] + 1e-5, (plain["worst_name"], plain["worst"], plain["all"], lib_plain["all"])
    # (3) row a12
    assert not routes, routes


@pytest.mark.gpu
def test_fused_optimizer_updates_reach_the_packed_weights():
    """torch.optim.Adam(fused=True) writes the parameters without touching `_version` (the key of the packed-operand caches): until round
    4 every training step after the first ran on the operands packed in step 1.  Trainable parameters are repacked on every use now:
    the losses of four steps equal those of a run that drops every cache before each forward, and they move."""
    from flownet2_amd import functional as Fn, nets

    def run(invalidate):
        P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
        opt = torch.optim.Adam(list(P.values()), lr=1e-3, fused=True)
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.rand(2, 3, 128, 192, device="cuda", generator=g)
        b = torch.rand(2, 3, 128, 192, device="cuda", generator=g)
        gt = torch.randn(2, 2, 128, 192, device="cuda", generator=g)
        out = []
        for _ in range(4):
            if invalidate:
                Fn.invalidate_weight_caches()
            opt.zero_grad(set_to_none=True)
            loss = nets.multiscale_loss(nets.flownet_c_core(P, a - 0.43, b - 0.43, Fn), gt, Fn)
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        # and the frozen copy of the trained net computes with the trained weights
        with torch.no_grad():
            Pf = {k: v.detach() for k, v in P.items()}
            out.append(float(nets.multiscale_loss(nets.flownet_c_core(Pf, a - 0.43, b - 0.43, Fn), gt, Fn)))
        return out

    cached, fresh = run(False), run(True)
    assert cached == fresh, (cached, fresh)
    assert len(set(cached)) == len(cached), cached


@pytest.mark.gpu
def test_weight_gradients_on_the_second_stream_are_the_same_bits():
    """functional.set_wgrad_side_stream (round 5; parallel.GradientExchange switches it on for a single-rank job): the weight gradients of
    every layer but the stem run on a second HIP stream beside the data-gradient chain and the main stream joins when the backward pass
    ends.  Same kernels on the same operands: every gradient must have the bits of the one-stream pass, pass after pass (a missing
    dependency between the streams shows up as a gradient that differs from run to run), through four fused-Adam steps, and also when the
    gradients are ACCUMULATED by a second backward pass (which must stay on the main stream)."""
    from flownet2_amd import functional as Fn, nets

    def run(pixels, accumulate=False):
        Fn.set_wgrad_side_stream(pixels)
        try:
            P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=2).items()}
            opt = torch.optim.Adam(list(P.values()), lr=1e-3, fused=True)
            g = torch.Generator(device="cuda").manual_seed(3)
            a = torch.rand(4, 3, 192, 256, device="cuda", generator=g) - 0.43
            b = torch.rand(4, 3, 192, 256, device="cuda", generator=g) - 0.43
            gt = torch.randn(4, 2, 192, 256, device="cuda", generator=g) * 3
            grads, losses = [], []
            for it in range(4):
                opt.zero_grad(set_to_none=True)
                for _ in range(2 if accumulate else 1):
                    loss = nets.multiscale_loss(nets.flownet_c_core(P, a, b, Fn), gt, Fn)
                    loss.backward()
                grads.append({k: v.grad.clone() for k, v in P.items()})
                opt.step()
                losses.append(float(loss.detach()))
            return grads, losses
        finally:
            Fn.set_wgrad_side_stream(0)

    ref_g, ref_l = run(0)
    for attempt in range(3):
        got_g, got_l = run(36000)
        assert got_l == ref_l, (attempt, got_l, ref_l)
        for it, (gg, rg) in enumerate(zip(got_g, ref_g)):
            bad = [k for k in rg if not torch.equal(gg[k], rg[k])]
            assert not bad, (attempt, it, bad)
    acc_ref, _ = run(0, accumulate=True)
    acc_got, _ = run(36000, accumulate=True)
    for gg, rg in zip(acc_got, acc_ref):
        assert all(torch.equal(gg[k], rg[k]) for k in rg)


@pytest.mark.gpu
@pytest.mark.parametrize("trainable", [("deconv2.w", "deconv2.b", "upsample_flow3to2.w", "upsample_flow3to2.b", "Convolution5.w", "Convolution5.b"),
                                       ("upsample_flow5to4.w",), ("deconv4.b", "Convolution3.w")])
def test_partial_freeze_whose_first_trainable_tensors_are_a_stage_s_own_parameters(trainable):
    """A fine-tune that freezes everything in front of a refinement stage: the stage's deconvolution / flow up-sampling weights are then the
    first tensors that require grad, and the in-place Concat route must still put them into the graph (round-5 advisor finding: the stage
    looked only at its INPUTS and fell through to the inference path, so those parameters silently got no gradient).  The gradients of the
    trainable subset equal those of the fully trainable net -- same kernels, same operands."""
    from flownet2_amd import functional as Fn, nets
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.rand(2, 3, 128, 192, device="cuda", generator=g) - 0.43
    b = torch.rand(2, 3, 128, 192, device="cuda", generator=g) - 0.43
    gt = torch.randn(2, 2, 128, 192, device="cuda", generator=g) * 3

    def grads(names):
        P = {k: v.cuda().requires_grad_(names is None or k in names) for k, v in nets.init_params("C", seed=7).items()}
        loss = nets.multiscale_loss(nets.flownet_c_core(P, a, b, Fn), gt, Fn)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: (None if v.grad is None else v.grad.clone()) for k, v in P.items()}

    l_all, g_all = grads(None)
    l_part, g_part = grads(set(trainable))
    assert l_all == l_part
    for k, v in g_part.items():
        if k in trainable:
            assert v is not None, f"{k} is trainable but got no gradient"
            assert torch.equal(v, g_all[k]), (k, float((v - g_all[k]).abs().max()))
        else:
            assert v is None, k


@pytest.mark.gpu
def test_relu_chain_conv1_conv2_gives_the_gradients_of_the_separate_passes():
    """Round 6: ReLUBackward of conv1 folded into the epilogue of conv2's data gradient (fn2_conv_backward_data_masked) and conv1's bias
    gradient out of its weight-gradient kernel (fn2_conv_backward_weights_bias) against the graph in which every layer undoes its own ReLU in
    a pass of its own.  The masked top_diff has the same bits either way (the same product g * (y > 0 ? 1 : slope)), so every WEIGHT gradient
    and every other bias gradient is bit-identical; conv1's bias gradient is summed in another order (inside the weight-gradient kernel):
    equal to fp32 rounding.  Run twice: bit-reproducible."""
    from flownet2_amd import functional as Fn, nets
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.rand(2, 3, 128, 192, device="cuda", generator=g) - 0.43
    b = torch.rand(2, 3, 128, 192, device="cuda", generator=g) - 0.43
    gt = torch.randn(2, 2, 128, 192, device="cuda", generator=g) * 3

    def grads(chain):
        nets.RELU_CHAIN[0] = chain
        try:
            P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=3).items()}
            loss = nets.multiscale_loss(nets.flownet_c_core(P, a, b, Fn), gt, Fn)
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.detach()), {k: v.grad.clone() for k, v in P.items()}
        finally:
            nets.RELU_CHAIN[0] = True

    l0, g0 = grads(False)
    l1, g1 = grads(True)
    l2, g2 = grads(True)
    assert l0 == l1 == l2
    for k in g0:
        if k == "conv1.b":
            rel = float((g1[k] - g0[k]).abs().max() / g0[k].abs().max())
            assert rel <= 2e-6, (k, rel)
        else:
            assert torch.equal(g1[k], g0[k]), (k, float((g1[k] - g0[k]).abs().max()))
        assert torch.equal(g1[k], g2[k]), k
