"""Wiring checks of the FlowNet2 stack that do not depend on kernels (CPU): the units the sub-networks hand to the fusion net."""
import torch

from flownet2_amd import nets


class _Stub:
    """Backend stand-in: identity-like ops that record what they were fed."""

    def __init__(self):
        self.nearest_inputs = []

    def resample(self, x, h, w, type=2, antialias=True):
        if type == 1:
            self.nearest_inputs.append(x)
        return torch.nn.functional.interpolate(x, size=(h, w), mode="nearest")

    def flow_warp(self, img, flow):
        return img

    def channel_norm(self, x):
        return x.pow(2).sum(1, keepdim=True).sqrt()


def test_fusion_inputs_carry_pixels(monkeypatch):
    """FlowNetCSS predicts px/20 (x FLOW_SCALE), FlowNet-SD predicts px/0.05 (x SD_FLOW_SCALE = 0.05, NOT x 20 x 0.05): with
    every sub-network returning ones, the two NEAREST Resample inputs of the fusion stage are 20 and 0.05."""
    ones = lambda n, h, w: torch.ones(n, 2, h // 4, w // 4)
    monkeypatch.setattr(nets, "flownet_c_core", lambda P, a, b, be, towers=None: {2: ones(a.shape[0], a.shape[2], a.shape[3])})
    monkeypatch.setattr(nets, "flownet_s_core", lambda P, x, be=None: {2: ones(x.shape[0], x.shape[2], x.shape[3])})
    monkeypatch.setattr(nets, "flownet_sd_core", lambda P, x, be: ones(x.shape[0], x.shape[2], x.shape[3]))
    seen = {}

    def fusion(P, x, be):
        seen["x"] = x
        return torch.zeros(x.shape[0], 2, x.shape[2], x.shape[3])

    monkeypatch.setattr(nets, "fusion_core", fusion)
    be = _Stub()
    img = torch.rand(1, 3, 64, 64) * 255
    nets.flownet2_deploy_forward({}, img, img, be)
    css, sd = be.nearest_inputs
    assert torch.allclose(css, torch.full_like(css, nets.FLOW_SCALE)) and nets.FLOW_SCALE == 20.0
    assert torch.allclose(sd, torch.full_like(sd, 0.05)) and nets.SD_FLOW_SCALE == 0.05
    x = seen["x"]                                   # [img0(3), flow_sd(2), flow_css(2), |flow_sd|, |flow_css|, err_sd, err_css]
    assert x.shape[1] == 11
    assert torch.allclose(x[:, 3:5], torch.full_like(x[:, 3:5], 0.05)) and torch.allclose(x[:, 5:7], torch.full_like(x[:, 5:7], 20.0))


def test_training_graph_helpers_have_the_gradients_of_plain_slices_and_cat():
    """nets._SplitTowers / _StackedAndFirstTower / _ConcatInPlace (round 5: the training graph's tower split and in-place Concats) against the
    torch ops they replace -- same values, same gradients (CPU, float64)."""
    import torch
    from flownet2_amd import nets
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 3, 4, 5, generator=g, dtype=torch.float64, requires_grad=True)
    wa, wb = torch.randn(3, 3, 4, 5, generator=g, dtype=torch.float64), torch.randn(3, 3, 4, 5, generator=g, dtype=torch.float64)
    a, b = nets._SplitTowers.apply(x * 1.0)
    (a * wa).sum().add((b * wb).sum()).backward()
    got = x.grad.clone(); x.grad = None
    y = x * 1.0
    ((y[:3] * wa).sum() + (y[3:] * wb).sum()).backward()
    assert torch.equal(got, x.grad); x.grad = None
    # the stacked batch for the next layer AND its first tower for the skip connection
    full, first = nets._StackedAndFirstTower.apply(x * 1.0)
    wf = torch.randn(6, 3, 4, 5, generator=g, dtype=torch.float64)
    ((full * wf).sum() + (first * wa).sum()).backward()
    got = x.grad.clone(); x.grad = None
    y = x * 1.0
    ((y * wf).sum() + (y[:3] * wa).sum()).backward()
    assert torch.allclose(got, x.grad, rtol=0, atol=1e-15); x.grad = None
    # Concat of producers that wrote their slices of one blob
    blob = torch.zeros(2, 7, 3, 3, dtype=torch.float64)

    class Into(torch.autograd.Function):            # a producer in the style of functional._OwnForwardConv(into=...)
        @staticmethod
        def forward(ctx, t, c0):
            blob.data[:, c0:c0 + t.shape[1]] = t * 2.0        # like the kernels: a write autograd's version counter does not see
            return blob[:, c0:c0 + t.shape[1]]

        @staticmethod
        def backward(ctx, gg):
            return gg * 2.0, None
    p = torch.randn(2, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    q = torch.randn(2, 4, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    wc = torch.randn(2, 7, 3, 3, generator=g, dtype=torch.float64)
    out = nets._ConcatInPlace.apply(blob, Into.apply(p, 0), Into.apply(q, 3))
    assert torch.equal(out, torch.cat([p.detach() * 2, q.detach() * 2], 1))
    (out * wc).sum().backward()
    gp, gq = p.grad.clone(), q.grad.clone()
    p.grad = q.grad = None
    (torch.cat([p * 2.0, q * 2.0], 1) * wc).sum().backward()
    assert torch.equal(gp, p.grad) and torch.equal(gq, q.grad)
