"""TRAIN-phase prototxt stepping (round 5): Net::Init with InsertSplits and the need-backward flags, Net::Backward through the Caffe-style
Backward_gpu of every layer mirror (stock layers: stock_layers.py; custom layers: layers.py), parameter diffs accumulated like the
reference -- against the training graph this repository benchmarks (nets.flownet_c_core + multiscale_loss under autograd)."""
import numpy as np
import pytest
import torch

from flownet2_amd import net as fnet, nets, templates


def test_insert_splits_and_backward_flags_on_the_flownetc_train_net():
    """util/insert_splits.cpp:12-86 + net.cpp:164-256 on the generated FlowNetC TRAIN prototxt (host logic; layers are set up on the CPU)."""
    n = fnet.Net(templates.flownet_c_train_prototxt(2, 128, 192), phase="TRAIN", device="cpu")
    splits = [nm for nm, l in zip(n.layer_names, n.layers) if l.type() == "Split"]
    # every blob with two consumers has its Split, named <blob>_<producer>_<top index>_split (SplitLayerName); a prediction feeds the
    # next stage's upsampling, its Downsample (as the size reference) and its loss: three tops
    assert "conv3_1_ReLU9_0_split" in splits and "predict_flow6_Convolution1_0_split" in splits and "flow_gt_scaled_Eltwise1_0_split" in splits
    i = n.layer_names.index("predict_flow6_Convolution1_0_split")
    assert [len(n.tops_[i])] == [3] and n.layers[i].layer_param_.top[2] == "predict_flow6_Convolution1_0_split_2"
    flags = dict(zip(n.layer_names, zip(n.layer_need_backward_, n.bottom_need_backward_)))
    assert flags["conv1a"] == (True, [False])                       # parameters to learn, but nothing to propagate into the images
    assert flags["conv2a"] == (True, [True]) and flags["flow_loss6"] == (True, [True, False])       # propagate_down: true / false of the prototxt
    assert flags["Downsample6"][0] is False and flags["Eltwise1"][0] is False                        # the ground-truth side is under no parameter
    assert len(n.learnable_) == 48                                  # the siamese towers share their blobs (ParamSpec names)
    assert n.outputs == ["flow_loss%d" % k for k in (6, 5, 4, 3, 2)]
    # a deploy net keeps its blobs unsplit (the forward results and the folded-ReLU / in-place paths are untouched)
    d = fnet.Net(templates.flownet_c_train_prototxt(1, 64, 64), phase="TEST", device="cpu")
    assert not any(l.type() == "Split" for l in d.layers)
    with pytest.raises(Exception):
        d.Backward()


def _inputs(N, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    img0 = torch.rand((N, 3, H, W), generator=g) - 0.43
    img1 = torch.rand((N, 3, H, W), generator=g) - 0.43
    gt = torch.randn((N, 2, H, W), generator=g) * 5.0
    gt[torch.rand((N, 1, H, W), generator=g).expand(-1, 2, -1, -1) < 0.05] = float("nan")      # 5 % NaN pixels (the loss mask)
    return img0.cuda(), img1.cuda(), gt.cuda()


@pytest.mark.gpu
def test_train_prototxt_forward_backward_matches_the_autograd_graph():
    from flownet2_amd import functional as Fn
    N, H, W = 2, 128, 192
    P = {k: v.cuda() for k, v in nets.init_params("C", seed=5).items()}
    img0, img1, gt = _inputs(N, H, W, 7)
    net = fnet.Net(templates.flownet_c_train_prototxt(N, H, W), phase="TRAIN", device="cuda")
    assert net.load_param_dict(P) == []
    net.ClearParamDiffs()
    loss = net.ForwardBackward(img0_nomean=img0, img1_nomean=img1, flow_gt=gt)
    # the same step through nets.py + autograd (what bench.py --mode train runs)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    flows = nets.flownet_c_core(Pg, img0, img1, Fn)
    want = nets.multiscale_loss(flows, gt, Fn)
    want.backward()
    assert abs(float(loss) - float(want.detach())) <= 1e-6 * abs(float(want.detach()))
    worst = 0.0
    for name, layer in zip(net.layer_names, net.layers):
        if layer.type() not in ("Convolution", "Deconvolution"):
            continue
        key = name[:-1] if name in ("conv1a", "conv2a", "conv3a", "conv1b", "conv2b", "conv3b") else name
        for blob, suffix in zip(layer.blobs_, (".w", ".b")):
            got, ref = blob.mutable_gpu_diff(), Pg[key + suffix].grad
            rel = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
            worst = max(worst, rel)
            # the towers: nets.py runs both images as one stacked batch (one weight-gradient launch over 2N samples), the prototxt has two layers
            # sharing a blob (two launches accumulated): another summation order.  Everything else runs the same kernels in the same order.
            # the same kernels; the summation orders differ where nets.py stacks the towers into one batch (one weight-gradient launch over 2N
            # samples against two accumulated ones) and where the five loss layers run as one launch: measured 6e-8 (last head) .. 2e-6 (conv1)
            assert rel <= 5e-6, (name, suffix, rel)
    # a second iteration ACCUMULATES into the diffs unless they are cleared (net.cpp:949-967 is the solver's job)
    w = net.layer_by_name("conv4").blobs_[0]
    once = w.mutable_gpu_diff().clone()
    net.ForwardBackward(img0_nomean=img0, img1_nomean=img1, flow_gt=gt)
    assert torch.allclose(w.mutable_gpu_diff(), 2 * once, rtol=1e-6, atol=0)
    print("worst relative L2 of a parameter gradient vs the autograd graph: %.2e" % worst)


@pytest.mark.gpu
def test_train_prototxt_sgd_steps_follow_torch_sgd():
    """Three plain SGD iterations through Net.ClearParamDiffs / ForwardBackward / (scale) / Update against torch.optim.SGD on the nets.py graph."""
    from flownet2_amd import functional as Fn
    N, H, W, lr = 1, 128, 128, 1e-3
    P = {k: v.cuda() for k, v in nets.init_params("C", seed=9).items()}
    net = fnet.Net(templates.flownet_c_train_prototxt(N, H, W), phase="TRAIN", device="cuda")
    assert net.load_param_dict(P) == []
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    opt = torch.optim.SGD(list(Pg.values()), lr=lr)
    losses = []
    for it in range(3):
        img0, img1, gt = _inputs(N, H, W, 20 + it)
        net.ClearParamDiffs()
        l = net.ForwardBackward(img0_nomean=img0, img1_nomean=img1, flow_gt=gt)
        for b in net.learnable_:
            b.mutable_gpu_diff().mul_(lr)               # SGDSolver::ComputeUpdateValue with momentum 0 (sgd_solver.cpp:207-240): diff *= local_rate
        net.Update()
        opt.zero_grad(set_to_none=True)
        want = nets.multiscale_loss(nets.flownet_c_core(Pg, img0, img1, Fn), gt, Fn)
        want.backward()
        opt.step()
        losses.append((float(l), float(want.detach())))
    for a, b in losses:
        assert abs(a - b) <= 2e-5 * abs(b), losses
    w = net.layer_by_name("conv5_1").blobs_[0].data
    assert float((w - Pg["conv5_1.w"]).norm() / Pg["conv5_1.w"].norm()) <= 1e-6
