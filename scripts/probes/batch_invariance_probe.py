"""Where does a sample's result start to depend on its batch?  Runs FlowNet2 (or C / S) in batch-invariant mode on a batch and on
one of its samples alone, records the output of every layer call (in call order) and prints the first ones that differ in bits."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flownet2_amd import functional as Fn, nets  # noqa: E402


def record(log):
    """Wrap the layer-level entry points of nets / functional so that every call appends (label, output)."""
    saved = {}

    def wrap(mod, name, label_of):
        f = getattr(mod, name)
        saved[(mod, name)] = f

        def g(*a, **k):
            r = f(*a, **k)
            t = r[1] if isinstance(r, tuple) else r
            if torch.is_tensor(t):
                if name == "conv_mfma_relu" and k.get("out") is not None:       # written into a Concat blob: only its own channel slice is defined yet
                    c0 = k.get("out_c0", 0)
                    t = t[:, c0:c0 + a[1].shape[0]]
                if name in ("deconv_gemm_relu", "upsample_flow_deconv") and k.get("out") is not None:
                    c0 = k.get("out_c0", 0)
                    t = t[:, c0:c0 + (a[3] if name == "deconv_gemm_relu" else 2)]
                log.append((label_of(a, k), t.detach().clone()))
            return r
        setattr(mod, name, g)

    wrap(nets, "_conv_routed", lambda a, k: "conv " + a[2])
    wrap(nets, "_deconv", lambda a, k: "deconv " + a[2])
    wrap(nets, "_conv_into_concat", lambda a, k: "conv_into " + a[2])
    for nm in ("correlation", "flow_warp", "resample", "channel_norm", "predict_flow_conv", "upsample_flow_deconv", "deconv_gemm_relu", "conv_gemm_relu",
               "conv_mfma_relu", "conv_k7s2_relu"):
        wrap(Fn, nm, lambda a, k, nm=nm: nm + " " + "x".join(str(v) for v in a[0].shape))
    return saved


def main():
    net = sys.argv[1] if len(sys.argv) > 1 else "2"
    B, H, W = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (4, 384, 768)))
    P = nets.init_params_flownet2(0) if net == "2" else nets.init_params(net, 0)
    Pd = {k: v.cuda() for k, v in P.items()}
    rng = np.random.default_rng(54)
    i0 = torch.from_numpy(rng.integers(0, 256, (B, 3, H, W)).astype(np.float32)).cuda()
    i1 = torch.from_numpy(np.roll(i0.cpu().numpy(), (3, -5), (2, 3)).copy()).cuda()
    run = (lambda a, b: nets.flownet2_deploy_forward(Pd, a, b, Fn)) if net == "2" else (lambda a, b: nets.deploy_forward(net, Pd, a, b, Fn))
    Fn.set_batch_invariant(True)
    logs = ([], [])
    saved = record(logs[0])
    with torch.no_grad():
        whole = run(i0, i1)
    for (mod, name), f in saved.items():
        setattr(mod, name, f)
    saved = record(logs[1])
    with torch.no_grad():
        one = run(i0[1:2], i1[1:2])
    for (mod, name), f in saved.items():
        setattr(mod, name, f)
    print("calls:", len(logs[0]), len(logs[1]), "final equal:", torch.equal(whole[1:2], one))
    bad = 0
    for (la, ta), (lb, tb) in zip(*logs):
        n = ta.shape[0] // one.shape[0] if ta.shape[0] % B == 0 else 0
        # towers stack 2N samples: sample 1 of the batch sits at rows 1 and N + 1
        if ta.shape[0] == 2 * B:
            sel = torch.stack([ta[1], ta[B + 1]])
        elif ta.shape[0] == B:
            sel = ta[1:2]
        else:
            sel = ta
        same = sel.shape == tb.shape and torch.equal(sel, tb)
        if not same and bad < 12:
            d = float((sel - tb).abs().max()) if sel.shape == tb.shape else float("nan")
            print("DIFF %-48s %-24s max|d| %.3e %s" % (la, tuple(ta.shape), d, "" if la == lb else "(label mismatch: %s)" % lb))
            bad += 1
    print("first differences listed above" if bad else "every recorded layer output is bit-identical")


if __name__ == "__main__":
    main()
