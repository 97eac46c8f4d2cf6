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
