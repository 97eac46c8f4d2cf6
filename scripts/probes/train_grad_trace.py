"""Where along the backward chain do the own path's gradients leave fp64?  Hooks on the output of every Convolution / Deconvolution of
nets.flownet_c_core record d(loss)/d(output) in the production graph and in the fp64 comparator (oracle/fp64_graph.py); relative L2 per layer,
in backward order.  (Round 4: the coarse levels' parameter gradients were 30 x further from fp64 than the library's.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flownet2_amd import functional as Fn, nets  # noqa: E402
from oracle import fp64_graph  # noqa: E402
from test_train_parity import _batch  # noqa: E402

store = {}
acts = {}
order = []


def patch(tag):
    oc, od = nets._conv, nets._deconv

    def conv(x, P, name, stride, pad, act=True, backend=None):
        y = oc(x, P, name, stride, pad, act, backend)
        if y.requires_grad:
            acts[(tag, name)] = y.detach().double().cpu()
            y.register_hook(lambda g, n=name: (store.__setitem__((tag, n), g.detach().double().cpu()), order.append(n) if tag == "own" else None) and None)
        return y

    def deconv(x, P, name, act=True, backend=None):
        y = od(x, P, name, act, backend)
        if y.requires_grad:
            acts[(tag, name)] = y.detach().double().cpu()
            y.register_hook(lambda g, n=name: (store.__setitem__((tag, n), g.detach().double().cpu()), order.append(n) if tag == "own" else None) and None)
        return y
    nets._conv, nets._deconv = conv, deconv
    return oc, od


dev = torch.device("cuda:0")
P = nets.init_params("C", seed=0)
a, b, gt = _batch(8, 320, 448, 4)
Pd = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
oc, od = patch("own")
pre = [(im.to(dev) * (1.0 / 255.0)) - 0.43 for im in (a, b)]
loss = nets.multiscale_loss(nets.flownet_c_core(Pd, pre[0], pre[1], Fn), gt.to(dev), Fn)
loss.backward()
nets._conv, nets._deconv = oc, od
patch("f64")
fp64_graph.flownetc_train_reference(P, a, b, gt, device=dev)
nets._conv, nets._deconv = oc, od
rel = lambda x, y: float((x - y).norm() / y.norm())
print("%-22s %-12s %-12s" % ("layer (backward order)", "grad of out", "activation"))
for n in order:
    if ("f64", n) in store:
        print("%-22s %.2e     %.2e" % (n, rel(store[("own", n)], store[("f64", n)]), rel(acts[("own", n)], acts[("f64", n)])))
