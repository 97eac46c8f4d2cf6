#!/usr/bin/env python
"""bench.py -- FlowNetC forward throughput on MI355X (BASELINE.json configs[1]).

A "step" = one FlowNetC deploy forward over one batch of 8 synthetic 448x320 image pairs per GPU
(weak scaling: every rank owns its own batch, no data-path collective -- SURVEY.md section 8e).
`--mode train` times fwd + bwd + RCCL gradient all-reduce + Adam instead (configs[3]).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is launched through
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from flownet2_amd import functional as Fn   # noqa: E402
from flownet2_amd import nets, ops, parallel   # noqa: E402

CONV_STACK_NOTE = ("own fp32 MFMA kernels for every layer, no library GEMM or convolution in the forward step: direct 5x5/2 (conv2, conv3), "
                   "Winograd F(2x2,3x3) for 3x3/1 (conv3_1, conv4_1), small-map kernel with deterministic split-K (conv4 .. conv6_1), 1x1 "
                   "(conv_redir; the weight^T x bottom GEMM of the 4x4/2 deconvolutions + own col2im/bias/ReLU pass into the concat blob), "
                   "7x7/2 stem, flow heads; no Concat copies, no element-wise glue kernels; train mode: own weight-gradient (fp32 MFMA, "
                   "deterministic split) and data-gradient kernels (Winograd, transposed 5x5/2 and 3x3/2, 4x4/2 as convolution) for every "
                   "layer incl. the stem, the 2-channel heads and the smallest maps")


def LIB_FALLBACKS():
    """Convolution / Deconvolution calls of this process that left the own kernels (functional.lib_conv2d / lib_conv_transpose2d)."""
    from flownet2_amd import functional as Fn
    return int(Fn.LIBRARY_FALLBACKS[0])

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md "Chip-level parameters" (spec; 6290 measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3    # same table: dense f32-input MFMA peak (= f32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: after the synchronisation in front of the timed region the chip needs ~10 steps (25 ms) to come back to its steady
    # clocks (2.85, 2.70, 2.59 ... 2.31 ms per step, also under hipGraph replay); 100 steps keep that ramp below 1 % of the mean
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="image pairs per GPU per step")
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=448)
    ap.add_argument("--mode", choices=["fwd", "train"], default="fwd")
    ap.add_argument("--net", choices=["C", "2"], default="C", help="C = FlowNetC (headline, configs[1]); 2 = full FlowNet2 stack (configs[2]: use --batch 4 --height 384 --width 768)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip the FlowNet2 768x384 / 1024x448 and train-step legs reported under `extra`")
    ap.add_argument("--graph", action="store_true", help="capture a forward step into a hipGraph and replay it (measured: no gain, the step is not launch-bound)")
    ap.add_argument("--corr-iters", type=int, default=200)
    ap.add_argument("--bucket-mb", type=int, default=48, help="gradient all-reduce bucket size (train mode)")
    ap.add_argument("--wgrad-side-pixels", type=int, default=36000, help="train mode, one rank: weight gradients of maps up to this many pixels on a second HIP stream (0 = off)")
    ap.add_argument("--sd-stream", choices=["auto", "on", "off"], default="auto", help="FlowNet2: FlowNet-SD on a second stream beside the CSS stack (nets.set_sd_side_stream)")
    return ap.parse_args()


def synth_batch(batch, h, w, seed, device):
    """uint8-valued BGR images as scripts/run-flownet.py:28-35 feeds them (raw 0..255 floats)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (batch, 3, h, w)).astype(np.float32)
    b = np.roll(a, (3, -5), (2, 3)) + rng.normal(0, 2, a.shape).astype(np.float32)
    return torch.from_numpy(a).to(device), torch.from_numpy(np.clip(b, 0, 255).astype(np.float32)).to(device)


def corr_roofline(device, batch, h, w, iters):
    """Time the correlation kernel alone (HIP events on the launch stream) at the conv3 shape of the workload."""
    C, H, W, D2 = 256, h // 8, w // 8, 441
    g = torch.Generator(device=device).manual_seed(0)
    a = torch.randn(batch, C, H, W, device=device, generator=g)
    b = torch.randn(batch, C, H, W, device=device, generator=g)
    p = ops.corr_params(20, 1, 20, 1, 2)
    out = torch.empty(batch, D2, H, W, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # ~60 ms of the same launch first: the chip needs ~25 ms of load to reach its steady clocks (the first ~500 launches run at 44.8 us
    # on average, the rest at 41.3: DESIGN.md 3.1), and the timed launches follow without a synchronisation in between
    for _ in range(1500):
        ops.correlation_forward(p, a, b, out=out)
    e0.record()
    for _ in range(iters):
        ops.correlation_forward(p, a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters                       # seconds per launch
    # cold caches: the 68 MB working set fits the 256 MiB Infinity Cache, so back-to-back launches (and the launch inside
    # the net, whose inputs conv3 has just written) find their inputs on chip; with a 1 GiB fill in between they come from HBM
    flush = torch.empty(256 << 20, dtype=torch.float32, device=device)
    cold = []
    for _ in range(8):
        flush.fill_(1.0)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        ops.correlation_forward(p, a, b, out=out)
        c1.record()
        torch.cuda.synchronize()
        cold.append(c0.elapsed_time(c1) * 1e3)
    del flush
    cold_us = sorted(cold)[len(cold) // 2]
    # the backward of the same layer (both bottom diffs; one launch since round 6), timed the same way: 2 x the forward's flops
    gtop = torch.randn(batch, D2, H, W, device=device, generator=g)
    for _ in range(300):
        ops.correlation_backward(p, a, b, gtop)
    b0e, b1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0e.record()
    for _ in range(max(1, iters // 2)):
        ops.correlation_backward(p, a, b, gtop)
    b1e.record()
    torch.cuda.synchronize()
    t_bwd = b0e.elapsed_time(b1e) * 1e-3 / max(1, iters // 2)
    del gtop
    # per-launch events (includes launch gaps) vs back-to-back average: take the back-to-back average
    alg_bytes = 4.0 * batch * H * W * (2 * C + D2)               # read both maps once + write top once
    alg_flops = 2.0 * C * D2 * batch * H * W
    tf = alg_flops / t / 1e12
    gbps = alg_bytes / t / 1e9
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes,
    # calibrated -- profiles/rNN_rocprof_summary.md).  Counters cannot be read from inside this process, so
    # the figure comes from the newest committed profile of exactly this kernel and shape.
    traffic, traffic_detail, profiled = None, None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_corr_hbm.json")), reverse=True):
        try:
            pj = json.load(open(f))
            if pj.get("kernel", "").endswith("[%d,%d,%d,%d]" % (batch, C, H, W)):
                traffic = round(pj["traffic_bytes_per_launch"])
                traffic_detail = {"unit": "bytes per launch", "source": "profiles/" + os.path.basename(f),
                                  "vs_algorithmic": round(pj["traffic_bytes_per_launch"] / alg_bytes, 3)}
                # the same fraction from the committed rocprofv3 kernel trace (slower clocks under the profiler), and what the
                # matrix pipes really executed: MFMA instructions x 2048 flop (products against the zero padding are skipped)
                us = pj["avg_us_kernel_trace"]
                profiled = {"source": "profiles/" + os.path.basename(f), "us_per_launch": round(us, 2),
                            "frac": round(alg_flops / (us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                            "executed_flops_per_launch": pj.get("executed_flops_per_launch"),
                            "frac_executed": round(pj["executed_flops_per_launch"] / (us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4) if pj.get("executed_flops_per_launch") else None,
                            "mfma_busy": round(pj["mfma_util"], 4) if pj.get("mfma_util") else None}
                break
        except Exception:
            pass
    # `achieved` / `frac` are what THIS run measured (HIP events on the launch stream around `iters` back-to-back launches).  The same
    # fraction from the newest committed rocprofv3 kernel trace of this exact kernel and shape rides along as frac_profiled (the figure
    # profiles/ can be checked against); `frac_agrees_with_profile` says whether the two are within 3 % of each other -- a committed
    # profile of an older build of the kernel must not speak for this one.
    frac_live = tf / F32_MFMA_PEAK_TFLOPS
    agrees = (abs(profiled["frac"] - frac_live) <= 0.03 * frac_live) if profiled else None
    return {
        "kernel": "corr_fwd (K=1,md=20,s2=2) [%d,%d,%d,%d]" % (batch, C, H, W),
        "bound": "mfma", "achieved": round(tf, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(frac_live, 4),
        "frac_source": "live hipEvent timing of this run (%d back-to-back launches after 1500 untimed ones)" % iters,
        "frac_agrees_with_profile": agrees,
        "achieved_live": round(tf, 3), "frac_live": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
        "frac_profiled": profiled["frac"] if profiled else None, "profiled": profiled,
        "traffic": traffic, "traffic_detail": traffic_detail,
        "us_per_launch": round(t * 1e6, 2), "us_per_launch_cold_caches": round(cold_us, 2),
        "alg_flops_per_launch": alg_flops, "alg_bytes_per_launch": alg_bytes,
        "hbm": {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4)},
        "backward": {"kernel": "corr_bwd (both bottom diffs, one launch)", "us_per_call": round(t_bwd * 1e6, 2), "alg_flops_per_call": 2 * alg_flops,
                     "achieved": round(2 * alg_flops / t_bwd / 1e12, 3), "unit": "TFLOP/s", "frac": round(2 * alg_flops / t_bwd / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                     "alg_bytes_per_call": 4.0 * batch * H * W * (D2 + 4 * C)},
        "note": "exact-fp32 correlation is FMA-bound (59 flop/B > machine balance); hbm = algorithmic bytes / time",
    }


def step_mfma_busy():
    """Matrix-pipe utilisation of the whole FlowNetC step from the newest committed counter pass (profiles/rNN_step_mfma.json, written by
    scripts/summarize_profiles.py from `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python bench.py`):
    counters cannot be read from inside this process."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_mfma.json")), reverse=True):
        try:
            pj = json.load(open(f))
            return {"mfma_busy_step": round(pj["mfma_busy_step"], 4), "mfma_busy_step_source": "profiles/" + os.path.basename(f)}
        except Exception:
            pass
    return {"mfma_busy_step": None, "mfma_busy_step_source": None}


def step_percentiles(marks):
    ts = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1))
    pick = lambda q: round(ts[min(len(ts) - 1, int(q * len(ts)))], 4)
    if os.environ.get("FN2_BENCH_STEPS"):           # debugging aid: the device time of every step, in order
        print("steps ms:", " ".join("%.3f" % marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)), file=sys.stderr)
    return [pick(0.1), pick(0.5), pick(0.9)]


def cpu_baseline(P_cpu, img0, img1, flow_gpu, budget_s=30.0, min_reps=9):
    """Oracle leg: the same graph on the host (C oracle ops + torch-CPU fp32 conv), bounded sample: one untimed pass (page-in, thread pools),
    then >= 9 timed repetitions (fewer only if the budget runs out); value = pairs / MEDIAN repetition, p10 / p90 beside it -- a shared
    256-cpu host makes single repetitions vary by 2x."""
    import oracle
    from oracle import backend as cpu_backend
    i0, i1 = img0.cpu(), img1.cpu()
    n = i0.shape[0]
    with torch.no_grad():
        flow_cpu = nets.deploy_forward("C", P_cpu, i0, i1, cpu_backend)
        times, t_all = [], time.time()
        while len(times) < min_reps and (time.time() - t_all < budget_s or len(times) < 3):
            t0 = time.time()
            nets.deploy_forward("C", P_cpu, i0, i1, cpu_backend)
            times.append(time.time() - t0)
    epe = float(((flow_gpu.cpu() - flow_cpu) ** 2).sum(1).sqrt().mean())
    cores = max(oracle.num_threads(), torch.get_num_threads())
    ts = sorted(times)
    q = lambda f: ts[min(len(ts) - 1, int(f * len(ts)))]
    return {"value": round(n / q(0.5), 3), "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "value_p10_p90": [round(n / q(0.9), 3), round(n / q(0.1), 3)], "repetitions": len(ts),
            "sample": f"median of {len(ts)} x FlowNetC deploy forward (after one untimed pass), batch {n} @{i0.shape[3]}x{i0.shape[2]}: C oracle "
                      f"(restated reference kernels, OpenMP) + torch-CPU fp32 conv; {os.cpu_count()} host cpus"}, epe


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run, one rank per GPU
    (RCCL over xGMI), and pass its JSON line through.  Fails loudly when the box has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: this box has {have} visible GPU(s); refusing to report a {n}-GPU number from fewer ranks")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


class CudaRuntime:
    """What run_workload needs from the device side: the operator backend of the graphs, synchronisation, timing events, the optimizer.
    The product runtime is this one (HIP kernels through flownet2_amd.functional, HIP events, fused Adam); tests/test_bench_ranks.py
    drives the same run_workload -- its rank logic, barriers, max-over-ranks timing, per-rank inputs and the gradient exchange -- with a
    CPU stand-in over the gloo backend."""
    backend = Fn

    def synchronize(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def optimizer(self, plist):
        return torch.optim.Adam(plist, lr=1e-5, fused=True)     # one multi-tensor kernel for the 48 parameter blobs (Caffe's AdamSolver: one kernel per blob)


def check_ranks(world, gpus, device):
    """The launcher must have started exactly --gpus ranks and the collective library must have connected all of them (one all-reduce
    of ones): a number reported for N GPUs from fewer ranks would be wrong by construction."""
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
    seen = parallel.ranks_seen(device)
    if seen != world:
        raise SystemExit(f"bench.py: all-reduce of ones returned {seen}, expected {world}")
    return seen


def rank_seed(rank):
    return 1234 + rank              # every rank its own synthetic batch (weak scaling: per-GPU work fixed)


nets.RELU_CHAIN[0] = os.environ.get("FN2_RELU_CHAIN", "1") == "1"       # A/B hook: 0 = every layer undoes its own ReLU in a pass of its own (rounds 1-5)
AHEAD = [os.environ.get("FN2_BENCH_TARGETS_AHEAD", "1") == "1"]      # A/B hook: 0 = Downsample(GT) between forward and backward (rounds 1-5)
WGRAD_SIDE_PIXELS = [36000]       # --wgrad-side-pixels: GradientExchange's second stream for the weight gradients (single-rank jobs)


def run_workload(net, mode, B, H, W, steps, warmup, device, world, rank, bucket_mb=48, graph=False, settle_s=1.0, rt=None, local_grads=False,
                 force_collective=False):
    """W untimed warm-up steps (+ untimed settling steps until `settle_s` seconds of back-to-back stepping have passed: the chip
    needs ~25 ms of load to come back to its steady clocks, and a short run otherwise sits inside that ramp), then EXACTLY `steps`
    timed steps between barrier + synchronize on both sides.  Returns the measurements and what the caller needs for the oracle leg."""
    rt = rt or CudaRuntime()
    be = rt.backend
    P_cpu = nets.init_params("C", seed=0) if net == "C" else nets.init_params_flownet2(seed=0)   # same weights on every rank
    P = {k: v.to(device) for k, v in P_cpu.items()}
    img0, img1 = synth_batch(B, H, W, seed=rank_seed(rank), device=device)

    if mode == "train":
        for v in P.values():
            v.requires_grad_(True)
        plist = list(P.values())
        opt = rt.optimizer(plist)
        gt = torch.randn(B, 2, H, W, device=device) * 5
        gt[torch.rand(B, 1, H, W, device=device).expand(-1, 2, -1, -1) < 0.05] = float("nan")
        parallel.broadcast_params(plist, src=0)
        # the ONE exchange of the path: sum-all-reduce of the fp32 gradients (39.18 M floats = 156.7 MB) over RCCL, scaled by
        # 1/world (parallel.cpp:377), in reverse-order buckets launched from gradient hooks while backward is still running;
        # identical Adam step on every rank
        # local_grads: the SAME step without the collective (every rank keeps its own gradients) -- the comparison leg from which
        # train_leg() reports how much of the all-reduce is not hidden behind backward
        NEG_MEAN = [torch.full((3,), -0.43, device=device, dtype=torch.float32)]
        exchange = parallel.GradientExchange([P[k] for k in P], bucket_bytes=bucket_mb << 20, local_only=local_grads, wgrad_side_pixels=WGRAD_SIDE_PIXELS[0],
                                             force_collective=force_collective)

        def step():
            exchange.zero_grad()
            tg = nets.loss_targets_ahead(gt, be) if AHEAD[0] else None      # the ground-truth pyramid: issued first, on the second stream
            if hasattr(be, "scale_shift") and img0.is_cuda:
                # Eltwise{1/255} + mean subtraction of both images written straight into the stacked tower batch (product and difference
                # rounded separately: the bits of `im * (1 / 255) - 0.43`; two launches instead of four element-wise kernels and a concat)
                towers = torch.empty((2 * B, 3, H, W), device=device, dtype=torch.float32)
                be.scale_shift(img0, 1.0 / 255.0, NEG_MEAN[0], out=towers[:B])
                be.scale_shift(img1, 1.0 / 255.0, NEG_MEAN[0], out=towers[B:])
                flows = nets.flownet_c_core(P, None, None, be, towers=towers)
            else:
                pre = [(im * (1.0 / 255.0)) - 0.43 for im in (img0, img1)]
                flows = nets.flownet_c_core(P, pre[0], pre[1], be)
            loss = nets.multiscale_loss(flows, gt, be, targets=tg)
            loss.backward()
            exchange.finish()
            opt.step()
            return loss
    else:
        def step():
            with torch.no_grad():
                if net == "2":
                    return nets.flownet2_deploy_forward(P, img0, img1, be)
                return nets.deploy_forward("C", P, img0, img1, be)

    out = None
    for _ in range(warmup):
        out = step()
    use_graph = mode == "fwd" and graph
    if use_graph:
        # One step = ~130 kernels: captured once (hipGraph through torch's CUDAGraph; same kernels, order and buffers)
        # and replayed, which removes the host launch path.  The warm-up above has run every lazy initialisation.
        rt.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        g.replay()
        step_fn = g.replay
    else:
        step_fn = step
    # Everything the host has to do before the timed region happens BEFORE the settling steps (event objects, a full garbage collection: a
    # few milliseconds in which the device would sit idle and fall back to its low clocks), so that the synchronisation in front of the
    # timed region is followed by the first timed launch at once.
    marks = [rt.event() for _ in range(steps + 1)]
    # The cyclic collector is paused for the K timed steps (config.python_gc): a generation-2 pass over the interpreter's objects
    # takes milliseconds -- two steps of the batch-1 configuration -- and has nothing to collect here (no reference cycles in a step).
    import gc
    gc_was_on = gc.isenabled() and os.environ.get("FN2_BENCH_GC", "off") != "on"
    if gc_was_on:
        gc.collect()
        gc.disable()
    # cold figure: the same K steps timed straight after the W warm-up steps, with NO settling steps in between (the clock ramp of the
    # first ~25 ms of load is inside it) -- reported as value_cold next to the steady-clock headline; rank-local, no barrier
    rt.synchronize()
    t_c = time.perf_counter()
    for _ in range(steps):
        step_fn()
    rt.synchronize()
    elapsed_cold = time.perf_counter() - t_c
    # settling: untimed steps, back to back, until the chip has been under this load for settle_s seconds
    rt.synchronize()
    settle_steps, t_s = 0, time.perf_counter()
    while warmup > 0 and time.perf_counter() - t_s < settle_s and settle_steps < 2000:
        for _ in range(16):
            step_fn()
        settle_steps += 16
        rt.synchronize()
    settling_s = time.perf_counter() - t_s
    if world > 1:
        dist.barrier()
    rt.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        r = step_fn()
        if r is not None:
            out = r
        marks[i + 1].record()                  # per-step spread (device time); the headline stays the host clock around all K steps
    rt.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    elapsed = parallel.max_over_ranks(elapsed, device)
    elapsed_cold = parallel.max_over_ranks(elapsed_cold, device)
    res = {"elapsed": elapsed, "params": P, "marks": marks, "out": out, "P_cpu": P_cpu, "img0": img0, "img1": img1, "use_graph": use_graph,
           "gc_paused": gc_was_on, "settle_steps": settle_steps, "settling_s": settling_s, "elapsed_cold": elapsed_cold}
    if mode == "train":
        res["n_buckets"] = len(exchange.buckets)
        res["buckets_launched_inside_backward"] = exchange.launched_in_backward      # of the last timed step
        res["exchange_world"] = exchange.world
        res["exchange_collective"] = exchange.collective
        res["gradient_bytes_copied_into_buckets"] = exchange.copied_bytes          # of the last step: what was NOT produced in its bucket slot
        exchange.remove()
    return res


def flownet2_epe_vs_cpu(P_cpu, img0, img1, flow_gpu):
    """Oracle leg for the FlowNet2 extras: the same stacked graph on the host, once; (EPE, seconds)."""
    from oracle import backend as cpu_backend
    t0 = time.time()
    with torch.no_grad():
        ref = nets.flownet2_deploy_forward(P_cpu, img0.cpu(), img1.cpu(), cpu_backend)
    return float(((flow_gpu.cpu() - ref) ** 2).sum(1).sqrt().mean()), time.time() - t0


def _sd_on(args, shape):
    if args.sd_stream != "auto":
        return args.sd_stream == "on"
    picks = [v for k, v in nets.sd_side_stream_picks().items() if k[1] == tuple(shape)]
    return bool(picks[-1]) if picks else True


def extras(device, args):
    """BASELINE configs 3, 5 (per-GPU leg) and 4 on the same GPU, after the headline: reported under `extra`, never as `value`."""
    ex = {}
    # config 5's per-GPU leg twice: one pair per step (the reference's runner, scripts/run-flownet-many.py:38-81, forwards pair by pair) and
    # four pairs per step -- what scripts/run_flownet_many.py does by default (it batches up to --batch 8 equal-sized pairs per GPU; the
    # .flo bytes do not depend on the batch: batch-invariant mode)
    for key, (B, H, W, steps) in {"flownet2_768x384": (4, 384, 768, 20), "flownet2_1024x448": (1, 448, 1024, 30),
                                  "flownet2_1024x448_batch4": (4, 448, 1024, 12)}.items():
        # one pair per step is ~340 launches for 5 ms of GPU work on two streams: on a box with a slow host the step becomes launch-bound (5.65 ms
        # seen once against 4.98): that leg replays a captured step (same kernels, same arguments; 4.97 against 4.98-5.00 ms on a fast host)
        m = run_workload("2", "fwd", B, H, W, steps, 5, device, 1, 0, graph=(B == 1))
        p = step_percentiles(m["marks"])
        ex[key] = {"metric": "image-pairs/sec FlowNet2 (CSS+SD+fusion) forward at %dx%d" % (W, H), "value": round(B * steps / m["elapsed"], 2),
                   "unit": "image-pairs/s", "batch": B, "steps": steps, "warmup": 5, "ms_per_step": round(m["elapsed"] / steps * 1e3, 4),
                   "ms_per_step_p10_p50_p90": p, "dtype": "f32",
                   "conv_tflops": round(nets.flownet2_conv_flops(H, W) * B * steps / m["elapsed"] / 1e12, 2),
                   "streams": ("FlowNet-SD on a second HIP stream beside the FlowNetC -> S -> S stack" if _sd_on(args, (B, 3, H, W)) else "one")
                              + (" (picked by timing both layouts on the first calls)" if args.sd_stream == "auto" else ""),
                   "launch": "hipGraph replay" if m["use_graph"] else "host launches"}
        tm = [v for k, v in nets.sd_side_stream_timings().items() if k[1] == (B, 3, H, W)]
        if tm:
            ex[key]["stream_layouts_timed_ms"] = tm[-1]            # what "auto" measured on THIS box when it picked (one synchronised forward each)
        if not args.no_cpu_baseline and not key.endswith("_batch4"):
            epe, secs = flownet2_epe_vs_cpu(m["P_cpu"], m["img0"], m["img1"], m["out"])
            ex[key]["epe_vs_cpu_oracle"] = epe
            ex[key]["cpu_oracle_seconds_per_batch"] = round(secs, 2)
        del m
        torch.cuda.empty_cache()
    ex["caffe_adapter_forward"] = caffe_adapter_leg()
    ex["train_448x320"] = train_leg(device, 1, 0, args.bucket_mb)
    if not args.no_cpu_baseline:
        ex["train_448x320"].update(train_parity(device))
    return ex


def caffe_adapter_leg(B=8, H=320, W=448, iterations=20):
    """The Caffe-side number: a FlowNetC core forward (pre-processed pair -> predict_flow2) chained from LayerRegistry-created layers -- the
    plug-ins of flownet2_amd/csrc/caffe_adapter for Convolution / Deconvolution / Correlation, the reference's own in-place ReLU and Concat
    between them -- timed the way `caffe time` does (tools/caffe.cpp:346-366) by the adapter test library.  What a maintainer who follows
    INTEGRATION.md sees; the gap to the fused graph of the headline is the separate ReLU passes and Concat copies (and hipMalloc'ed Caffe
    blobs instead of one arena).  Checker-side code (oracle/ref.py drives the C shim): after the timed region, never the headline."""
    try:
        from oracle import ref
        if not ref.adapter_available():
            return {"error": "adapter test library not built"}
        ref.use("adapter")
        try:
            if not hasattr(ref.lib(), "fn2ref_flownetc_time"):
                return {"error": "adapter shim built without the reference's ReLU / Concat sources"}
            cached = ref.flownetc_time(B, H, W, warmup=5, iterations=iterations, use_cache=True)
            fresh = ref.flownetc_time(B, H, W, warmup=3, iterations=iterations, use_cache=False)
        finally:
            ref.use("ref")
    except Exception as e:      # noqa: BLE001 -- a leg of `extra`: report, do not lose the bench line
        return {"error": str(e)[:300]}
    relu = sum(ms for n, ms in cached["layers"] if n.endswith("_relu"))
    cat = sum(ms for n, ms in cached["layers"] if n.startswith(("concat", "blob20")))
    return {"metric": "ms per FlowNetC core forward through LayerRegistry-created plug-ins (`caffe time` style), batch %d @%dx%d" % (B, W, H),
            "ms_per_forward": round(cached["total_ms"], 4), "value": round(B / cached["total_ms"] * 1e3, 2), "unit": "image-pairs/s",
            "ms_per_forward_repacking_every_forward": round(fresh["total_ms"], 4),
            "layers": len(cached["layers"]), "weight_packs_in_timed_region": cached["packs"], "weight_pack_reuses": cached["pack_reuses"],
            "separate_relu_layers_ms": round(relu, 4), "concat_layers_ms": round(cat, 4),
            "slowest_layers_ms": [[n, round(ms, 4)] for n, ms in sorted(cached["layers"], key=lambda t: -t[1])[:6]],
            "note": "per-layer times are synchronised per layer (caffe time's per-layer Timer) and do not add up to ms_per_forward; the head "
                    "(scale, Resample) and tail (x20, Resample) of the deploy net are not part of this chain"}


def train_leg(device, world, rank, bucket_mb=48, B=8, H=320, W=448, steps=20, warmup=5, settle_s=1.0, rt=None):
    """BASELINE config 4 (FlowNetC fwd + bwd + gradient all-reduce + Adam, batch 8 per GPU) on EVERY rank of the job: with world > 1 this is
    the leg that exercises the one collective of the path (parallel.GradientExchange: reverse-order buckets launched from gradient hooks over
    RCCL; the reference: P2PSync::on_gradients_ready, parallel.cpp:325-380, effective batch x world, docs/multigpu.md:11).  Collective
    calls inside: every rank must call it.  ms_per_step is the max over ranks; with world > 1 the same step is timed once more with
    world-local gradients (no collective), and allreduce_ms_exposed = the difference = the part of the exchange backward does not hide."""
    m = run_workload("C", "train", B, H, W, steps, warmup, device, world, rank, bucket_mb=bucket_mb, settle_s=settle_s, rt=rt)
    ms = m["elapsed"] / steps * 1e3
    mb = 4e-6 * nets.num_params(m["P_cpu"])
    if world > 1:
        what, coll = "fwd+bwd+allreduce+Adam", "one fp32 sum-all-reduce of %.1f MB per step in %d buckets <= %d MB over RCCL, scaled by 1/%d" % (mb, m["n_buckets"], bucket_mb, world)
    else:
        what, coll = "fwd+bwd+Adam", "none (world 1): no collective is issued in a single-rank job; %.1f MB of gradients stay where the kernels wrote them" % mb
    leg = {"metric": "image-pairs/sec FlowNetC %s at %dx%d" % (what, W, H), "value": round(world * B * steps / m["elapsed"], 2),
           "unit": "image-pairs/s", "n_gpus": world, "batch": B, "global_batch": B * world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
           "ms_per_step_p10_p50_p90": step_percentiles(m["marks"]), "dtype": "f32", "loss": float(m["out"].detach()),
           "conv_tflops": round(nets.conv_flops("C", H, W) * B * 3 * steps / m["elapsed"] / 1e12, 2),
           "parallelism": "dp%d" % world, "collective": coll,
           "grad_buckets": m["n_buckets"], "buckets_launched_inside_backward": m["buckets_launched_inside_backward"],
           "gradient_bytes_copied_into_buckets": m["gradient_bytes_copied_into_buckets"],
           "library_conv_fallbacks": LIB_FALLBACKS(),
           "streams": ("weight gradients of maps <= %d px on a second HIP stream beside the data-gradient chain%s" % (
                           WGRAD_SIDE_PIXELS[0], "; a bucket's all-reduce is ordered behind that stream" if world > 1 else ""))
                      if (WGRAD_SIDE_PIXELS[0] > 0 and device.type == "cuda") else "one"}
    del m
    if device.type == "cuda":
        torch.cuda.empty_cache()
    if world > 1:
        ml = run_workload("C", "train", B, H, W, steps, warmup, device, world, rank, bucket_mb=bucket_mb, settle_s=settle_s, rt=rt, local_grads=True)
        ms_local = ml["elapsed"] / steps * 1e3
        leg["ms_per_step_local_gradients"] = round(ms_local, 4)
        leg["allreduce_ms_exposed"] = round(ms - ms_local, 4)
        leg["ranks_seen_by_rccl"] = parallel.ranks_seen(device)
        del ml
        if device.type == "cuda":
            torch.cuda.empty_cache()
    elif device.type == "cuda" and rt is None and not (dist.is_available() and dist.is_initialized()):
        leg["rccl_bucket_path_world1"] = rccl_world1_leg(device, bucket_mb, B, H, W, steps, warmup, settle_s, ms)
    return leg


def rccl_world1_leg(device, bucket_mb, B, H, W, steps, warmup, settle_s, ms_local):
    """The exchange as a multi-rank job runs it, on the one GPU this box has: a process group of ONE rank over the "nccl" (= RCCL) backend and
    GradientExchange forced through its full bucket path (gradients produced in the flat bucket slots, hooks, one all_reduce per bucket on
    RCCL's stream ordered behind the weight-gradient stream, wait).  An all-reduce over one rank moves no data between GPUs: the figure is
    the cost of the PATH (flattened gradients, collective launches, stream ordering) against the local step -- not a scaling number."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=device)
    except Exception as e:       # noqa: BLE001 -- a box without a usable RCCL: say so instead of failing the bench line
        return {"error": "init_process_group('nccl', world_size=1) failed: %s" % str(e)[:200]}
    try:
        m = run_workload("C", "train", B, H, W, steps, warmup, device, 1, 0, bucket_mb=bucket_mb, settle_s=settle_s, force_collective=True)
        ms = m["elapsed"] / steps * 1e3
        return {"ms_per_step": round(ms, 4), "ms_per_step_local_gradients": round(ms_local, 4), "overhead_ms": round(ms - ms_local, 4),
                "overhead_frac": round(ms / ms_local - 1.0, 4), "grad_buckets": m["n_buckets"],
                "buckets_launched_inside_backward": m["buckets_launched_inside_backward"],
                "gradient_bytes_copied_into_buckets": m["gradient_bytes_copied_into_buckets"],
                "ranks_seen_by_rccl": parallel.ranks_seen(device), "librccl_mapped": "librccl" in open("/proc/self/maps").read(),
                "note": "backend nccl (= RCCL), world_size 1, same process: the full bucket path of parallel.GradientExchange, no inter-GPU traffic"}
    finally:
        dist.destroy_process_group()
        if device.type == "cuda":
            torch.cuda.empty_cache()



def train_parity(device, B=8, H=320, W=448):
    """Checker leg of the training configuration (after the timing, never inside it): loss and every parameter gradient of ONE training
    step with production routing against the fp64 comparator oracle/fp64_graph.py (float64 autograd graph, none of the product's
    kernels), at the configuration's own size.  grad_rel_l2_* = ||g - g64|| / ||g64|| over all 39 M gradient values."""
    from oracle import fp64_graph
    P = nets.init_params("C", seed=0)
    Pd = {k: v.to(device).requires_grad_(True) for k, v in P.items()}
    img0, img1 = synth_batch(B, H, W, seed=4321, device=device)
    g = torch.Generator().manual_seed(9)
    gt = torch.randn(B, 2, H, W, generator=g) * 5
    gt[(torch.rand(B, 1, H, W, generator=g) < 0.05).expand(-1, 2, -1, -1)] = float("nan")
    pre = [(im * (1.0 / 255.0)) - 0.43 for im in (img0, img1)]
    with fp64_graph.record_relu_branches() as rec:
        loss = nets.multiscale_loss(nets.flownet_c_core(Pd, pre[0], pre[1], Fn), gt.to(device), Fn)
    loss.backward()
    torch.cuda.synchronize()
    t0 = time.time()
    grads = {k: v.grad for k, v in Pd.items()}
    loss_p, g_p = fp64_graph.flownetc_train_reference(P, img0, img1, gt, device=device, masks=rec.branches)
    with fp64_graph.record_relu_branches() as rec64:
        loss64, g64 = fp64_graph.flownetc_train_reference(P, img0, img1, gt, device=device)
    # yardstick: the SAME torch graph in fp32 on the library's kernels (none of ours) against the plain fp64 graph
    _, g_lib = fp64_graph.flownetc_train_reference(P, img0, img1, gt, device=device, dtype=torch.float32)
    a, b, lib = fp64_graph.grad_agreement(grads, g_p), fp64_graph.grad_agreement(grads, g64), fp64_graph.grad_agreement(g_lib, g64)
    flips, units = fp64_graph.relu_sign_flips(rec.branches, rec64.branches)
    return {"grad_rel_l2_vs_fp64_same_relu_branch": a["all"], "grad_rel_l2_vs_fp64_same_relu_branch_worst_param": [a["worst_name"], a["worst"]],
            "grad_rel_l2_vs_fp64_same_relu_branch_median_param": a["median"],
            "loss_rel_err_vs_fp64_same_relu_branch": abs(float(loss.detach()) - loss_p) / max(1.0, abs(loss_p)),
            "grad_rel_l2_vs_plain_fp64": b["all"], "grad_rel_l2_vs_plain_fp64_worst_param": [b["worst_name"], b["worst"]],
            "relu_sign_flips_vs_plain_fp64": flips, "relu_units": units,
            "library_fp32_grad_rel_l2_vs_plain_fp64": lib["all"], "library_fp32_grad_rel_l2_vs_plain_fp64_worst_param": [lib["worst_name"], lib["worst"]],
            "ref": "oracle/fp64_graph.py: float64 autograd graph on the same inputs (none of the product's kernels).  ..._same_relu_branch: every leaky "
                   "ReLU of the fp64 graph on the branch the fp32 run took = the rounding error of the product's kernels; ..._vs_plain_fp64: the fp64 "
                   "graph with its own ReLU signs -- the relu_sign_flips units (pre-activations within rounding of zero) dominate that figure, "
                   "for the library's fp32 kernels (library_fp32_...: the same torch graph in float32) as for ours (%.1f s)" % (time.time() - t0)}


def _claim_stdout():
    """The driver reads ONE JSON line from stdout.  Native libraries print there too (RCCL's version banner goes to the C stdout at
    communicator init and is flushed at exit, BEHIND the line): file descriptor 1 is pointed at stderr for the life of the process and the
    line is written to a private duplicate of the original stdout."""
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    line_out = _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); cuda not available")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" == RCCL on ROCm
    ranks_seen = check_ranks(world, args.gpus, device)  # what RCCL actually connected (one all-reduce of ones)

    from flownet2_amd import nets as _nets
    _nets.set_sd_side_stream(args.sd_stream)
    WGRAD_SIDE_PIXELS[0] = args.wgrad_side_pixels
    B, H, W = args.batch, args.height, args.width
    m = run_workload(args.net, args.mode, B, H, W, args.steps, args.warmup, device, world, rank, args.bucket_mb, args.graph)
    elapsed, marks, out, P_cpu, img0, img1 = m["elapsed"], m["marks"], m["out"], m["P_cpu"], m["img0"], m["img1"]

    headline = args.mode == "fwd" and args.net == "C" and (B, H, W) == (8, 320, 448)
    multi_rank_train = None
    if world > 1 and headline and not args.no_extras:
        # The multi-rank line carries the leg that runs the path's one collective: BASELINE config 4 on all ranks, after the headline's
        # timed region (every rank calls it: barriers and all-reduces inside).  The forward headline above stays embarrassingly parallel.
        keep = {k: m[k] for k in ("settle_steps", "settling_s", "elapsed_cold", "use_graph", "gc_paused")}
        m = dict(keep)
        torch.cuda.empty_cache()
        multi_rank_train = train_leg(device, world, rank, args.bucket_mb)
    if rank == 0:
        pairs = world * B * args.steps
        conv_gf = (nets.conv_flops("C", H, W) if args.net == "C" else nets.flownet2_conv_flops(H, W)) * B / 1e9
        res = {
            "metric": "image-pairs/sec " + ("FlowNetC " if args.net == "C" else "FlowNet2 (CSS+SD+fusion) ") + ("forward" if args.mode == "fwd" else ("fwd+bwd+allreduce+Adam" if world > 1 else "fwd+bwd+Adam")) + " at %dx%d" % (W, H),
            "value": round(pairs / elapsed, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            # the same K steps timed straight after the W warm-up steps (no settling steps: clock ramp included), and what came between
            "value_cold": round(pairs / m["elapsed_cold"], 2), "ms_per_step_cold": round(m["elapsed_cold"] / args.steps * 1e3, 4),
            "settling_s": round(m["settling_s"], 3), "untimed_steps_before_timed_region": args.warmup + args.steps + m["settle_steps"],
            "ms_per_step_p10_p50_p90": step_percentiles(marks), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("FlowNetC" if args.net == "C" else "FlowNet2") + " %s (correlation max_disp=20 stride_2=2), batch %d/GPU @%dx%d, synthetic uint8-valued "
                                   "pairs, seeded random-init weights (%.2f M params)" % ("deploy forward" if args.mode == "fwd" else "train step", B, W, H, nets.num_params(P_cpu) / 1e6),
                       "global_batch": B * world, "parallelism": "replicas x%d (no data-path collective)" % world if args.mode == "fwd" else ("dp%d (RCCL all-reduce)" % world if world > 1 else "dp1 (collective: none, world 1)"),
                       "conv_stack": CONV_STACK_NOTE + " (%.1f GFLOP/step/GPU)" % conv_gf,
                       "library_conv_fallbacks": LIB_FALLBACKS(),
                       "launch": "hipGraph replay" if m["use_graph"] else "host launches", "python_gc": "paused for the timed steps" if m["gc_paused"] else "on",
                       "untimed_settling_steps_after_warmup": m["settle_steps"],
                       "ranks_seen_by_rccl": ranks_seen if world > 1 else "no process group (world 1)"},
            "conv_tflops": round(conv_gf * (3 if args.mode == "train" else 1) * args.steps / elapsed / 1e3, 2),
        }
        if args.mode == "fwd" and args.net == "C" and (B, H, W) == (8, 320, 448):
            res.update(step_mfma_busy())
        if world == 1:
            res["roofline"] = corr_roofline(device, B, H, W, args.corr_iters)
            if args.mode == "fwd" and args.net == "C" and not args.no_cpu_baseline:
                cb, epe = cpu_baseline(P_cpu, img0, img1, out)
                res["cpu_baseline"] = cb
                res["epe_vs_cpu_oracle"] = epe
            elif args.mode == "fwd" and args.net == "2" and not args.no_cpu_baseline:
                res["epe_vs_cpu_oracle"], _ = flownet2_epe_vs_cpu(P_cpu, img0, img1, out)
            if headline and not args.no_extras:
                del m, out
                torch.cuda.empty_cache()
                res["extra"] = extras(device, args)
        if multi_rank_train is not None:
            res["extra"] = {"train_448x320": multi_rank_train}
        line_out.write(json.dumps(res) + "\n")
        line_out.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
