#!/usr/bin/env python
"""py3 re-authoring of the reference's inference driver (scripts/run-flownet.py:1-127) on the MI355X path.

    python scripts/run_flownet.py model.caffemodel deploy.prototxt.template img0 img1 out.flo     # the reference's argument order
    python scripts/run_flownet.py [--net C|S|2] [--weights w.caffemodel|w.caffemodel.h5|w.npz] img0 img1 out.flo   # built-in graph (flownet2_amd/nets.py)

First form: the template's $TARGET_WIDTH$ ... $SCALE_HEIGHT$ variables are substituted exactly as run-flownet.py:38-58 does, the net
is built layer by layer from the prototxt through the layer registry (flownet2_amd.net.Net = caffe.Net of run-flownet.py:64) and the
weights are copied by layer name (Net::CopyTrainedLayersFrom).  `model.caffemodel` may be `seed:S` / `seed:C` / `seed:2` for the seeded
synthetic weights of nets.init_params (no released model can be downloaded here); templates for the three nets ship in
flownet2_amd/prototxt_templates/.

Same behaviour as the reference script: images are read as RGB, fed as BGR raw 0..255 floats, the net runs at the
ADAPTED (x64) size and the flow is resampled / rescaled back to the TARGET size, and the result is written as .flo.
Differences: no prototxt template / .caffemodel (neither is in the reference tree; the graph is flownet2_amd/nets.py,
weights are a .caffemodel / .caffemodel.h5 read by flownet2_amd.caffemodel, a name->array .npz in Caffe blob layout, or seeded random), and no "retry up to 5x on NaN" loop -- the
reference needs it for a race in its kernels (run-flownet.py:72-96); these kernels are deterministic."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flownet2_amd import flo, nets  # noqa: E402
from flownet2_amd import functional as Fn  # noqa: E402


def read_image(path):
    from PIL import Image
    a = np.asarray(Image.open(path))
    if a.ndim == 2:
        return a[np.newaxis, np.newaxis].astype(np.float32)                               # run-flownet.py:30
    return a[np.newaxis].transpose(0, 3, 1, 2)[:, [2, 1, 0]].astype(np.float32)          # :31 RGB -> BGR


def load_params(net, weights, device):
    """-> (params on the device, mean or None).  `weights`: a .caffemodel (matched by layer name like Net::CopyTrainedLayersFrom,
    net.cpp:752-800; the DataAugmentation layers' per-channel mean comes with it) or a name -> array .npz in Caffe blob layout."""
    P = nets.init_params_flownet2(0) if net == "2" else nets.init_params(net, 0)
    mean = None
    if weights and (weights.endswith(".caffemodel") or weights.endswith(".h5")):       # ".h5": the HDF5 form, dispatched like net.cpp:804-811
        from flownet2_amd import caffemodel
        found, means, ignored = caffemodel.to_params(caffemodel.load_file(weights), P)
        for k, v in found.items():
            P[k] = torch.from_numpy(np.ascontiguousarray(v))
        missing = [k for k in P if k not in found]
        print(f"{weights}: {len(found)} blobs copied, {len(missing)} parameters keep their seeded values, ignored source layers: {ignored}")
        if means:
            first = next(iter(means.values()))
            mean = torch.from_numpy(first[:3].copy()).to(device)        # img0 / img1 layers hold the same data-set mean
    elif weights:
        blob = np.load(weights)
        for k in P:
            if k in blob:
                assert tuple(blob[k].shape) == tuple(P[k].shape), f"{k}: shape mismatch"      # net.cpp:783-799
                P[k] = torch.from_numpy(blob[k].astype(np.float32))
    return {k: v.to(device) for k, v in P.items()}, mean


def infer(net, P, img0, img1, mean=None):
    with torch.no_grad():
        if net == "2":
            return nets.flownet2_deploy_forward(P, img0, img1, Fn, mean=mean)
        return nets.deploy_forward(net, P, img0, img1, Fn, mean=mean)


def infer_prototxt(model, template_path, img0, img1, device):
    """run-flownet.py:38-98: substitute, build, load, forward; returns predict_flow_final [N,2,H,W]."""
    from flownet2_amd import net as fnet, prototxt
    H, W = img0.shape[2], img0.shape[3]
    text = prototxt.substitute(open(template_path).read(), prototxt.deploy_vars(W, H))
    n = fnet.Net(text, phase="TEST", device=device)
    if img0.shape[0] != 1:
        n.reshape_inputs(img0.shape[0])
    if model.startswith("seed:"):
        kind = model.split(":", 1)[1]
        P = nets.init_params_flownet2(0) if kind == "2" else nets.init_params(kind, 0)
        missing = [m for m in n.load_param_dict(P) if m != "scale_conv1"]
        if missing:
            raise SystemExit("no seeded weights for layers: " + ", ".join(missing))
    else:
        ignored = n.CopyTrainedLayersFrom(model)
        if ignored:
            print("Ignoring source layers:", ", ".join(ignored))
    out = n.forward(**{n.inputs[0]: img0, n.inputs[1]: img1})
    if "predict_flow_final" not in n.blobs:
        raise SystemExit("the net has no blob 'predict_flow_final' (run-flownet.py:98)")
    return n.blobs["predict_flow_final"].data


def _positionals(argv):
    """Positional arguments of either form: the values of the value-taking options are not positionals."""
    takes_value = {"--gpu", "--weights", "--net"}
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
        elif a in takes_value:
            skip = True
        elif not a.startswith("-"):
            out.append(a)
    return out


def main():
    # the reference's argument form (run-flownet.py:12-18: caffemodel deployproto img0 img1 out) is recognised by its second positional,
    # the deploy prototxt (template) -- not by counting tokens: `--weights w.npz --gpu 1 a.png b.png out.flo` has five non-dash tokens too
    pos = _positionals(sys.argv[1:])
    if len(pos) >= 5 and pos[1].endswith((".prototxt", ".template")) and "--net" not in sys.argv:
        ap = argparse.ArgumentParser()
        ap.add_argument("caffemodel", help="path to model (or seed:S / seed:C / seed:2)")
        ap.add_argument("deployproto", help="path to deploy prototxt template")
        ap.add_argument("img0"); ap.add_argument("img1"); ap.add_argument("out")
        ap.add_argument("--gpu", type=int, default=0)
        ap.add_argument("--verbose", action="store_true")
        ap.add_argument("--no-batch-invariant", action="store_true")
        a = ap.parse_args()
        if not a.caffemodel.startswith("seed:") and not os.path.exists(a.caffemodel):
            raise SystemExit("caffemodel does not exist: " + a.caffemodel)                   # run-flownet.py:20
        if not os.path.exists(a.deployproto):
            raise SystemExit("deploy-proto does not exist: " + a.deployproto)                # :21
        for f in (a.img0, a.img1):
            if not os.path.exists(f):
                raise SystemExit("image does not exist: " + f)
        dev = torch.device("cuda", a.gpu)
        torch.cuda.set_device(dev)
        Fn.set_batch_invariant(not a.no_batch_invariant)
        i0, i1 = torch.from_numpy(read_image(a.img0)).to(dev), torch.from_numpy(read_image(a.img1)).to(dev)
        flow = infer_prototxt(a.caffemodel, a.deployproto, i0, i1, dev)
        flo.write_flo(a.out, flow[0].cpu().numpy())
        print("wrote", a.out, tuple(flow.shape[2:]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("img0"); ap.add_argument("img1"); ap.add_argument("out")
    ap.add_argument("--net", choices=["C", "S", "2"], default="C")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--no-batch-invariant", action="store_true", help="let kernel selection follow the work size (see run_flownet_many.py)")
    a = ap.parse_args()
    for f in (a.img0, a.img1):
        if not os.path.exists(f):
            raise SystemExit("image does not exist: " + f)
    dev = torch.device("cuda", a.gpu)
    # same arithmetic as run_flownet_many.py on any batching / sharding of a list: a pair's .flo does not depend on how it was computed
    Fn.set_batch_invariant(not a.no_batch_invariant)
    P, mean = load_params(a.net, a.weights, dev)
    i0, i1 = torch.from_numpy(read_image(a.img0)).to(dev), torch.from_numpy(read_image(a.img1)).to(dev)
    flow = infer(a.net, P, i0, i1, mean)
    flo.write_flo(a.out, flow[0].cpu().numpy())                                         # predict_flow_final -> (H,W,2)
    print("wrote", a.out, tuple(flow.shape[2:]))


if __name__ == "__main__":
    main()
