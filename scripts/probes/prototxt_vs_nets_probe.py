"""First layer at which the prototxt executor and nets.py disagree: both paths call nets.conv_forward / deconv_forward / the flow-head
kernels / conv_mfma_relu in the same order; their outputs are recorded and compared in call order."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flownet2_amd import functional as Fn, net as fnet, nets, templates  # noqa: E402


def record(log):
    saved = []

    def wrap(mod, name):
        f = getattr(mod, name)
        saved.append((mod, name, f))

        def g(*a, **k):
            r = f(*a, **k)
            if torch.is_tensor(r):
                t = r
                if k.get("out") is not None:
                    c0 = k.get("out_c0", 0)
                    n = a[1].shape[0] if name == "conv_mfma_relu" else (a[3] if name == "deconv_gemm_relu" else 2)
                    t = r[:, c0:c0 + n]
                log.append((name + " " + "x".join(str(v) for v in a[0].shape), t.detach().clone()))
            return r
        setattr(mod, name, g)
    for nm in ("conv_mfma_relu", "conv_k7s2_relu", "conv_gemm_relu", "deconv_gemm_relu", "predict_flow_conv", "upsample_flow_deconv", "lib_conv2d",
               "lib_conv_transpose2d", "correlation", "correlation_relu_into", "resample", "flow_warp", "channel_norm", "scale_shift"):
        wrap(Fn, nm)
    return saved


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "S"
    B, H, W = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (2, 128, 192)))
    P = nets.init_params_flownet2(0) if kind == "2" else nets.init_params(kind, 0)
    Pd = {k: v.cuda() for k, v in P.items()}
    rng = np.random.default_rng(7)
    i0 = torch.from_numpy(rng.integers(0, 256, (B, 3, H, W)).astype(np.float32)).cuda()
    i1 = torch.from_numpy(np.roll(i0.cpu().numpy(), (2, -3), (2, 3)).copy()).cuda()
    Fn.set_batch_invariant(True)
    logs = ([], [])
    saved = record(logs[0])
    with torch.no_grad():
        want = nets.flownet2_deploy_forward(Pd, i0, i1, Fn) if kind == "2" else nets.deploy_forward(kind, Pd, i0, i1, Fn)
    for mod, name, f in saved:
        setattr(mod, name, f)
    saved = record(logs[1])
    n = fnet.from_template(open(templates.template_path(kind)).read(), W, H, batch=B, device="cuda")
    n.load_param_dict(Pd)
    got = n.forward(img0=i0, img1=i1)["predict_flow_final"]
    for mod, name, f in saved:
        setattr(mod, name, f)
    print("final equal:", torch.equal(got, want), "max |d|", float((got - want).abs().max()), "calls", len(logs[0]), len(logs[1]))
    skip = ("scale_shift", "resample", "flow_warp", "channel_norm", "correlation")       # the executor's custom layers call ops directly
    logs = tuple([e for e in lg if not e[0].startswith(skip)] for lg in logs)
    for i, ((la, ta), (lb, tb)) in enumerate(zip(*logs)):
        same = ta.shape == tb.shape and torch.equal(ta, tb)
        print("%3d %-40s | %-40s %s" % (i, la, lb, "ok" if same else ("DIFF %.3e" % float((ta - tb).abs().max()) if ta.shape == tb.shape else "SHAPE")))
        if not same and la == lb:
            break


if __name__ == "__main__":
    main()
