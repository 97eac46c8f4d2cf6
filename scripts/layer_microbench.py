#!/usr/bin/env python
"""Micro-benchmarks of the custom layers at the SURVEY section 8(d) sizes: time, algorithmic bytes, achieved GB/s
(fraction of the 8 TB/s HBM peak).  Inputs are resident in HBM; >= 40 ms of warm-up launches, then 5 back-to-back rounds of 100 timed
launches, median."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops
dev = "cuda"
def timeit(f, iters=100, rounds=5):
    """Steady clocks: the chip needs ~25 ms of load to reach them (rounds 1-2 timed 100 launches after 20 warm-ups, i.e. the clock ramp), so
    the op first runs for >= 40 ms, and the rounds follow back to back without a synchronisation in between."""
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1) / 20, 1e-3)                      # ms per launch, first estimate
    for _ in range(min(20000, int(40.0 / per) + 1)): f()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(rounds + 1)]
    marks[0].record()
    for r in range(rounds):
        for _ in range(iters): f()
        marks[r + 1].record()
    torch.cuda.synchronize()
    return statistics.median(marks[r].elapsed_time(marks[r + 1]) / iters * 1e3 for r in range(rounds))
rows = []
ONLY = [w for w in os.environ.get("FN2_MB_ONLY", "").split(",") if w]      # substrings: run only the matching workloads
def add(name, f, nbytes):
    if ONLY and not any(w in name for w in ONLY):
        return
    t = timeit(f)
    rows.append((name, t, nbytes, nbytes / t / 1e3))
g = torch.Generator(device=dev).manual_seed(0)
# FlowWarp: config B images and a 256-channel feature case
for (N, C, H, W) in [(4, 3, 384, 768), (4, 256, 48, 96)]:
    img = torch.rand(N, C, H, W, device=dev, generator=g); flow = torch.randn(N, 2, H, W, device=dev, generator=g) * 8
    gout = torch.randn(N, C, H, W, device=dev, generator=g)
    add(f"FlowWarp fwd [{N},{C},{H},{W}] iid flow", lambda: ops.flow_warp_forward(img, flow), 4 * N * H * W * (2 * C + 2))
    add(f"FlowWarp bwd [{N},{C},{H},{W}] iid flow", lambda: ops.flow_warp_backward(img, flow, gout), 4 * N * H * W * (3 * C + 4))
    # a smooth field of the same magnitude (what a network predicts): neighbouring pixels sample neighbouring addresses
    sm = torch.nn.functional.interpolate(torch.randn(N, 2, max(H // 64, 2), max(W // 64, 2), device=dev, generator=g) * 8, size=(H, W),
                                         mode="bilinear", align_corners=True).contiguous()
    add(f"FlowWarp fwd [{N},{C},{H},{W}] smooth flow", lambda: ops.flow_warp_forward(img, sm), 4 * N * H * W * (2 * C + 2))
    add(f"FlowWarp bwd [{N},{C},{H},{W}] smooth flow", lambda: ops.flow_warp_backward(img, sm, gout), 4 * N * H * W * (3 * C + 4))
# Resample: flow x4 up-sampling (config B), config A final flow, image identity
for (N, C, H, W, Ho, Wo) in [(4, 2, 96, 192, 384, 768), (8, 2, 80, 112, 320, 448), (4, 3, 384, 768, 384, 768), (4, 3, 436, 1024, 448, 1024)]:
    x = torch.randn(N, C, H, W, device=dev, generator=g)
    add(f"Resample LINEAR [{N},{C},{H},{W}]->[{Ho},{Wo}]", lambda: ops.resample_forward(x, Ho, Wo), 4 * N * C * (H * W + Ho * Wo))
# L1Loss at the training scales of config A, ChannelNorm / Downsample at config B
for (h, w) in [(80, 112), (5, 7)]:
    a = torch.randn(8, 2, h, w, device=dev, generator=g); b = torch.randn(8, 2, h, w, device=dev, generator=g)
    p = ops.l1_params(l2_per_location=True, normalize_by_num_entries=True)
    ws = ops.l1loss_workspace(a)
    add(f"L1Loss fwd [8,2,{h},{w}]", lambda: ops.l1loss_forward(p, a, b, ws), 4 * 2 * 8 * 2 * h * w)
    ops.l1loss_forward(p, a, b, ws)
    add(f"L1Loss bwd [8,2,{h},{w}]", lambda: ops.l1loss_backward(p, a, b, 1.0, ws), 4 * 4 * 8 * 2 * h * w)
# the five loss layers of config A's training net in ONE launch per direction (fn2_l1loss_forward_multi / _backward_multi)
sc = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
pa = [torch.randn(8, 2, h, w, device=dev, generator=g) for h, w in sc]; pb = [torch.randn(8, 2, h, w, device=dev, generator=g) for h, w in sc]
p = ops.l1_params(l2_per_location=True, normalize_by_num_entries=True)
lw = [0.005, 0.01, 0.02, 0.08, 0.32]
nb = sum(4 * 2 * 8 * 2 * h * w for h, w in sc)
add("L1Loss fwd, all 5 scales [8,2,80,112]..[8,2,5,7], one launch", lambda: ops.l1loss_forward_multi(p, pa, pb, lw), nb)
_, _, wsm = ops.l1loss_forward_multi(p, pa, pb, lw)
one = torch.ones((), device=dev)
add("L1Loss bwd, all 5 scales, one launch", lambda: ops.l1loss_backward_multi(p, pa, pb, lw, one, wsm), 2 * nb)
x = torch.randn(4, 3, 384, 768, device=dev, generator=g)
add("ChannelNorm fwd [4,3,384,768]", lambda: ops.channel_norm_forward(x), 4 * 4 * 384 * 768 * 4)
gt = torch.randn(8, 2, 320, 448, device=dev, generator=g)
add("Downsample [8,2,320,448]->[80,112]", lambda: ops.downsample_forward(gt, 80, 112), 4 * 8 * 2 * (320 * 448 + 80 * 112))
y = torch.randn(16, 128, 80, 112, device=dev, generator=g); bb = torch.randn(128, device=dev, generator=g)
add("bias + leaky ReLU in place [16,128,80,112]", lambda: ops.bias_leaky_relu_(y, bb, 0.1), 2 * 4 * y.numel())
xi = torch.randn(8, 512, 20, 28, device=dev, generator=g)
add("im2col 3x3/2 [8,512,20,28]", lambda: ops.im2col_forward(xi, 3, 1, 2), 4 * xi.numel() + 4 * 8 * 512 * 9 * 140)
col = torch.randn(8, 64 * 16, 40 * 56, device=dev, generator=g); b64 = torch.randn(64, device=dev, generator=g)
add("col2im 4x4/2 + bias + ReLU -> [8,64,80,112]", lambda: ops.col2im_bias_relu_forward(col, b64, 8, 64, 80, 112, 4, 1, 2), 4 * col.numel() + 4 * 8 * 64 * 80 * 112)
# FlowAugmentation at the training crop of config A (FlyingChairs 512x384 -> 448x320), Correlation1D with DispNetCorr1D's parameters
import numpy as np
flow_a = torch.randn(8, 2, 384, 512, device=dev, generator=g) * 8
c_id = np.zeros((8, 42), np.float32); c_id[:, :6] = [0, 0.01, -0.02, 0.05, 0.1, 0.08]
add("FlowAugmentation [8,2,384,512]->[320,448]", lambda: ops.flow_augmentation_forward(flow_a, c_id, c_id, 320, 448), 8 * 320 * 448 * 16)
img_a = torch.rand(8, 3, 384, 512, device=dev, generator=g)
c_col = c_id.copy(); c_col[:, 6:12] = 0.05; c_col[:, 12:34] = 0.02
mean_a = torch.tensor([0.41, 0.43, 0.45], device=dev)
pa = ops.data_aug_params(448, 320, 255.0, (0.51, 0.56, 0.65, 0.79, 0.01, -0.62, 0.35, -0.83, 0.44), ops.MEAN_PER_CHANNEL)
add("DataAugmentation spatial + mean [8,3,384,512]->[320,448]", lambda: ops.data_augmentation_forward(pa, img_a, c_id, mean_a), 4 * 8 * 3 * (384 * 512 + 320 * 448))
add("DataAugmentation spatial + eigen + colour + mean (statistics pass first)", lambda: ops.data_augmentation_forward(pa, img_a, c_col, mean_a), 4 * 8 * 3 * (2 * 384 * 512 + 320 * 448))
a1 = torch.randn(4, 256, 48, 96, device=dev, generator=g); b1 = torch.randn(4, 256, 48, 96, device=dev, generator=g)
p1d = ops.corr_params(40, 1, 40, 1, 1, single_direction=-1)
add("Correlation1D fwd [4,256,48,96] md 40 left (MFMA kernel)", lambda: ops.correlation1d_forward(p1d, a1, b1), 4 * 4 * 48 * 96 * (2 * 256 + 41))
# CustomData sample decode: a batch of 8 FlyingChairs samples (512x384): 10.125 B/pixel of packed bytes in, 9 fp32 planes out
from flownet2_amd import sample_format as SF
Hs, Ws, Ns = 384, 512, 8
nb = SF.sample_bytes(9, Hs, Ws, SF.FLOW_SAMPLE_SLICE_POINTS, SF.FLOW_SAMPLE_ENCODINGS)
packed = torch.randint(0, 256, (Ns, (nb + 15) // 16 * 16), device=dev, dtype=torch.uint8, generator=g)
add(f"CustomData decode [{Ns} x 9ch {Ws}x{Hs}] (one launch)", lambda: SF.decode_batch(packed, 9, Hs, Ws, SF.FLOW_SAMPLE_SLICE_POINTS, SF.FLOW_SAMPLE_ENCODINGS),
    Ns * (nb + 4 * 9 * Hs * Ws))
mean_t = torch.randn(9 * Hs * Ws, device=dev, generator=g)
add(f"CustomData decode [{Ns} x 9ch {Ws}x{Hs}] with a mean blob", lambda: SF.decode_batch(packed, 9, Hs, Ws, SF.FLOW_SAMPLE_SLICE_POINTS, SF.FLOW_SAMPLE_ENCODINGS, mean=mean_t, scale=1 / 255.),
    Ns * (nb + 4 * 9 * Hs * Ws) + 4 * 9 * Hs * Ws)
print("| layer | us | algorithmic MB | GB/s | % of 8 TB/s |\n|---|---|---|---|---|")
for name, t, nb, gbs in rows:
    print("| %s | %.1f | %.2f | %.0f | %.1f |" % (name, t, nb / 1e6, gbs, gbs / 80.0))

# host -> device leg of the sample decode (the CPU decode baselines live in tests/cpu_baselines.py: only tests/ may use the oracle)
import time
host = packed.cpu().numpy()
pinned = torch.from_numpy(host).pin_memory()
src = torch.from_numpy(host)
t0 = time.time()
for _ in range(20):
    pinned.copy_(src); dev_copy = pinned.to(dev, non_blocking=True); torch.cuda.synchronize()
print("host -> device copy of the packed batch through a persistent pinned buffer (%.1f MB): %.2f ms" % (host.nbytes / 1e6, (time.time() - t0) / 20 * 1e3))
