#!/usr/bin/env python
"""Round 5: the unit kernel (csrc/correlation_units.hip) against corr_fwd_pair in ONE process -- bit-identity first (every task policy,
BASELINE shapes + ragged ones + the fused ReLU / Concat-slice form), then interleaved timing at steady clocks (a warm-up second, then
rounds of N back-to-back launches per variant, order reversed every other round).  impl codes: fn2_debug_set_correlation_impl."""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops, _lib

L = _lib.lib()
P = ops.corr_params(20, 1, 20, 1, 2)


def mk(shape, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, device="cuda", generator=g), torch.randn(*shape, device="cuda", generator=g)


def check(impls):
    bad = 0
    for shape in [(8, 256, 40, 56), (4, 256, 48, 96), (1, 256, 56, 128), (2, 64, 16, 24), (3, 32, 11, 20), (1, 32, 5, 8), (16, 32, 9, 12), (5, 96, 30, 44), (8, 64, 40, 56), (8, 128, 24, 32), (2, 64, 40, 56), (1, 64, 40, 56), (4, 192, 40, 56)]:
        x, y = mk(shape)
        N, C, H, W = shape
        L.fn2_debug_set_correlation_impl(19)
        want = ops.correlation_forward(P, x, y)
        if shape[1] % 32 != 0:
            continue
        wide = torch.full((N, 441 + 40, H, W), 7.0, device="cuda")
        want_f = ops.correlation_forward(P, x, y, out=wide.clone(), out_c0=13, relu=True, negative_slope=0.1)
        for i in impls:
            L.fn2_debug_set_correlation_impl(i)
            got = torch.full_like(want, float("nan"))
            ops.correlation_forward(P, x, y, out=got)
            got_f = ops.correlation_forward(P, x, y, out=wide.clone(), out_c0=13, relu=True, negative_slope=0.1)
            torch.cuda.synchronize()
            same = torch.equal(got.view(torch.int32), want.view(torch.int32)) and torch.equal(got_f.view(torch.int32), want_f.view(torch.int32))
            if not same:
                bad += 1
                d = (got - want)
                nanc = int(torch.isnan(got).sum())
                print("MISMATCH impl %d shape %s: max abs %.3e, nan %d, differing %d / %d" % (i, shape, float(torch.nan_to_num(d).abs().max()), nanc,
                                                                                           int((got.view(torch.int32) != want.view(torch.int32)).sum()), got.numel()), flush=True)
    L.fn2_debug_set_correlation_impl(0)
    print("bit-identity vs corr_fwd_pair: %s" % ("OK" if bad == 0 else "%d FAILURES" % bad), flush=True)
    return bad


def timing(impls, shape, rounds=6, iters=1500):
    x, y = mk(shape)
    N, C, H, W = shape
    out = torch.empty(N, 441, H, W, device="cuda")

    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.correlation_forward(P, x, y, out=out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    L.fn2_debug_set_correlation_impl(impls[0])
    t0 = time.time()
    while time.time() - t0 < 1.0:
        run(500)
    res = {i: [] for i in impls}
    for rnd in range(rounds):
        for i in (impls if rnd % 2 == 0 else impls[::-1]):
            L.fn2_debug_set_correlation_impl(i)
            run(200)
            res[i].append(run(iters))
    L.fn2_debug_set_correlation_impl(0)
    fl = 2.0 * C * 441 * N * H * W
    rows = {}
    for i in impls:
        med = statistics.median(res[i])
        rows[i] = {"min_us": round(min(res[i]), 2), "median_us": round(med, 2), "max_us": round(max(res[i]), 2), "tflops_alg": round(fl / med / 1e6, 1),
                   "frac_fp32_mfma": round(fl / med / 1e6 / 157.3, 4)}
        print("shape %s impl %3d: min %.2f median %.2f max %.2f us  -> %.1f TFLOP/s alg = %.3f of 157.3" % (shape, i, min(res[i]), med, max(res[i]), fl / med / 1e6, fl / med / 1e6 / 157.3), flush=True)
    return rows


if __name__ == "__main__":
    impls = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [19, 0]
    implsB = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [19, 0]
    implsD = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [19, 0, 28, 26]
    out = {"check_failures": check(sorted(set(i for i in impls + implsB + implsD if i != 19)))}
    out["A"] = timing(impls, (8, 256, 40, 56))
    out["B"] = timing(implsB, (4, 256, 48, 96), rounds=4, iters=1000)
    out["D"] = timing(implsD, (1, 256, 56, 128), rounds=4, iters=1000)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/corr_units_ab.json", "w"), indent=1)
