#!/usr/bin/env python
"""gpurun_out/<tag>/ (rocprofv3 CSVs from scripts/profile_round.sh) -> profiles/<tag>_*.{md,json}."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = os.path.join(ROOT, "gpurun_out", tag)
OUT = os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)


def kernel_stats(path):
    return list(csv.DictReader(open(path)))


def counters(path, match):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name, n=100):
    return name if len(name) <= n else name[: n - 3] + "..."


lines = [f"# rocprofv3 summary, round tag {tag}", "",
         "Source: `bash scripts/profile_round.sh %s` on one MI355X (gfx950), ROCm 7.2; raw CSVs stay in gpurun_out/ (scratch)." % tag, ""]

# --- correlation micro-benchmark: kernel trace
ks = kernel_stats(os.path.join(R, "corr", "corr_kernel_stats.csv"))
lines += ["## `rocprofv3 --kernel-trace --stats -- python scripts/corr_microbench.py --iters 6000 --backward`", "",
          "(6,000 forward / 1,200 backward launches: the average includes the ~25 ms clock ramp of the first ~500 launches; rounds 1-2 profiled 200 launches, i.e. the ramp only)", "",
          "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
corr_avg_us = None
for r in ks[:6]:
    lines.append("| `%s` | %s | %.2f | %.2f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    if ("corr_fwd_units" in r["Name"] or "corr_fwd_pair" in r["Name"] or "corr_fwd_glds" in r["Name"] or "corr_fwd_mfma" in r["Name"]):
        corr_avg_us = float(r["AverageNs"]) / 1e3
lines += ["", "stdout of the same run:", "```", open(os.path.join(R, "corr_stdout.txt")).read().strip(), "```", ""]

# --- HBM counters + calibration
cal_txt = open(os.path.join(R, "cal_stdout.txt")).read().strip().splitlines()
known = {}
for l in cal_txt:
    p = l.split()
    known[p[0]] = (int(p[2]), int(p[4]))
cf = {k: counters(os.path.join(R, "cal_fetch", "cal_counter_collection.csv"), k) for k in ("channel_norm_fwd", "flow_warp_fwd")}
cw = {k: counters(os.path.join(R, "cal_write", "cal_counter_collection.csv"), k) for k in ("channel_norm_fwd", "flow_warp_fwd")}
fetch_factor = known["channel_norm_fwd"][0] / (cf["channel_norm_fwd"]["FETCH_SIZE"] * 1024.0)
write_factor = known["flow_warp_fwd"][1] / (cw["flow_warp_fwd"]["WRITE_SIZE"] * 1024.0)
f = counters(os.path.join(R, "pmc_fetch", "corr_counter_collection.csv"), "corr_fwd_")
w = counters(os.path.join(R, "pmc_write", "corr_counter_collection.csv"), "corr_fwd_")
sq = counters(os.path.join(R, "pmc_sq", "corr_counter_collection.csv"), "corr_fwd_")
fetch_bytes = f["FETCH_SIZE"] * 1024.0 * fetch_factor
write_bytes = w["WRITE_SIZE"] * 1024.0 * write_factor
alg = 4.0 * 8 * 40 * 56 * (2 * 256 + 441)
lines += ["## HBM traffic of the correlation forward kernel (`corr_fwd_units`) at [8,256,40,56] (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)", "",
          "Calibration (MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are in KiB and FETCH_SIZE under-reports on gfx950;",
          "calibrate on a known byte count in the same access width): `scripts/hbm_calibrate.py`", "",
          "| kernel | known read B | FETCH_SIZE KiB | bytes / (FETCH_SIZE*1024) | known write B | WRITE_SIZE KiB | bytes / (WRITE_SIZE*1024) |", "|---|---|---|---|---|---|---|"]
for k in ("channel_norm_fwd", "flow_warp_fwd"):
    lines.append("| %s | %d | %.0f | %.3f | %d | %.0f | %.3f |" % (k, known[k][0], cf[k]["FETCH_SIZE"], known[k][0] / (cf[k]["FETCH_SIZE"] * 1024),
                                                                   known[k][1], cw[k]["WRITE_SIZE"], known[k][1] / (cw[k]["WRITE_SIZE"] * 1024)))
lines += ["", "| quantity | value |", "|---|---|",
          "| FETCH_SIZE (KiB, avg per dispatch) | %.0f |" % f["FETCH_SIZE"],
          "| WRITE_SIZE (KiB, avg per dispatch) | %.0f |" % w["WRITE_SIZE"],
          "| read correction factor (4 B/lane coalesced, from channel_norm_fwd) | %.3f |" % fetch_factor,
          "| write correction factor (from flow_warp_fwd) | %.3f |" % write_factor,
          "| corrected HBM read bytes / launch | %.2f MB |" % (fetch_bytes / 1e6),
          "| corrected HBM write bytes / launch | %.2f MB |" % (write_bytes / 1e6),
          "| **traffic / launch** | **%.2f MB** |" % ((fetch_bytes + write_bytes) / 1e6),
          "| algorithmic bytes / launch 4*N*H*W*(2C+441) | %.2f MB |" % (alg / 1e6),
          "| traffic / algorithmic | %.3f |" % ((fetch_bytes + write_bytes) / alg), ""]
gui = sq.get("GRBM_GUI_ACTIVE", 0) / 8.0
lines += ["## SQ counters of the same kernel (`--pmc` pass of their own)", "", "| counter | avg per dispatch |", "|---|---|"]
for k in sorted(sq):
    lines.append("| %s | %.0f |" % (k, sq[k]))
if gui:
    util = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui)
    lines += ["", "GRBM_GUI_ACTIVE / 8 XCDs = %.0f cycles per launch; matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = **%.1f %%**."
              % (gui, 100 * util),
              "MFMA instructions issued: %.0f x 2048 flop = %.2f GFLOP (algorithmic 2*C*441*N*H*W = 4.05 GFLOP counts the products against the zero padding, which the kernel skips)."
              % (sq["SQ_INSTS_MFMA"], sq["SQ_INSTS_MFMA"] * 2048 / 1e9), ""]

# --- bench kernel table (steady state, find-db warmed)
ks = kernel_stats(os.path.join(R, "bench", "bench_kernel_stats.csv"))
lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5` (FlowNetC forward, batch 8 @448x320)", "",
          "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in ks[:34]:
    lines.append("| `%s` | %s | %.2f | %.2f | %s |" % (short(r["Name"], 90), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
# share of the own kernels in the steady-state step: the correlation roofline loop of bench.py (200 + launches of corr_fwd_*) is not part of a step
step_rows = [r for r in ks if "distribution_elementwise" not in r["Name"]]
def _per_step(r):
    calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
    if "corr_fwd_" in r["Name"] and calls > 25:
        return tot / calls * 25
    return tot
tot_ns = sum(_per_step(r) for r in step_rows)
own_ns = sum(_per_step(r) for r in step_rows if "fn2::" in r["Name"])
cat_calls = sum(int(r["Calls"]) for r in ks if "CatArrayBatchedCopy" in r["Name"])
lines += ["", "`fn2::` kernels: **%.1f %%** of the GPU time of the 25 traced steps (the correlation kernel counted once per step); "
          "`CatArrayBatchedCopy` launches: %.1f per step." % (100.0 * own_ns / tot_ns, cat_calls / 25.0)]
# --- matrix-pipe utilisation of the conv stack (counter pass over the same bench command)
pmc_path = os.path.join(R, "pmc_bench", "bench_counter_collection.csv")
if os.path.exists(pmc_path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(pmc_path)):
        per[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r["Kernel_Name"]] += 1
    rows = []
    for name, c in per.items():
        gui_k = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui_k > 0 and c.get("SQ_INSTS_MFMA", 0.0) > 0:
            rows.append((gui_k, name, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui_k), cnt[name]))
    # the correlation roofline loop of bench.py launches corr_fwd_* 1,500 + times in the same process: count the kernel once per step
    # (steps = dispatches of the stem kernel, which runs exactly once per step)
    steps_seen = max([n for _, name, _, n in rows if "conv_k7s2_relu" in name] or [0])
    if steps_seen:
        rows = [((g * steps_seen / n) if ("corr_fwd_" in name and n > steps_seen) else g, name, u, n) for g, name, u, n in rows]
    rows.sort(reverse=True)
    tot_gui = sum(r[0] for r in rows)
    tot_busy = sum(r[0] * r[2] for r in rows)
    lines += ["", "## Matrix-pipe utilisation of the step's MFMA kernels (`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE`, own pass over the same command)", "",
              "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); kernels ordered by their share of the MFMA kernels' cycles.", "",
              "| kernel | dispatches | share of cycles % | MFMA busy % |", "|---|---|---|---|"]
    for gui_k, name, u, n in rows[:16]:
        lines.append("| `%s` | %d | %.1f | %.1f |" % (short(name, 90), n, 100 * gui_k / tot_gui, 100 * u))
    lines += ["", "All MFMA kernels of the run together: **%.1f %%** matrix-pipe utilisation." % (100 * tot_busy / tot_gui)]
    # bench.py reports this figure as `mfma_busy_step` (it cannot read counters from inside its own process)
    json.dump({"tag": tag, "workload": "FlowNetC deploy forward, batch 8 @448x320", "mfma_busy_step": tot_busy / tot_gui,
               "kernels": [{"kernel": short(name, 90), "dispatches": n, "share_of_cycles": gui_k / tot_gui, "mfma_busy": u} for gui_k, name, u, n in rows[:16]],
               "source": "profiles/%s_rocprof_summary.md (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE pass over bench.py)" % tag},
              open(os.path.join(OUT, f"{tag}_step_mfma.json"), "w"), indent=1)

lines += ["", "bench line of the unprofiled run in the same session:", "```", open(os.path.join(R, "bench.json")).read().strip(), "```", ""]
open(os.path.join(OUT, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")

import shutil
for extra in ("bench.json", "bench_flownet2.json", "bench_flownet2_1024.json", "bench_train.json"):
    if os.path.exists(os.path.join(R, extra)) and os.path.getsize(os.path.join(R, extra)) > 0:
        shutil.copyfile(os.path.join(R, extra), os.path.join(OUT, f"{tag}_{extra}"))
if os.path.exists(os.path.join(R, "conv_bench_C.txt")):
    shutil.copyfile(os.path.join(R, "conv_bench_C.txt"), os.path.join(OUT, f"{tag}_conv_bench_flownetc.txt"))
for src, dst in (("conv_plane_bench_C.txt", "conv_plane_bench_flownetc.txt"), ("deconv_bench_C.txt", "deconv_bench_flownetc.txt"),
                 ("train_kernels.txt", "train_kernels.txt"), ("wgrad_bench.txt", "wgrad_bench.txt"), ("tconv_bench.txt", "tconv_bench.txt"),
                 ("wgrad_counters.txt", "wgrad_counters.txt")):
    if os.path.exists(os.path.join(R, src)):
        shutil.copyfile(os.path.join(R, src), os.path.join(OUT, f"{tag}_{dst}"))
for src, dst in (("bench_nonfn2.txt", "nonfn2_flownetc_fwd.txt"), ("bench2_nonfn2.txt", "nonfn2_flownet2_b4.txt"), ("bench2b1_nonfn2.txt", "nonfn2_flownet2_b1.txt"),
                 ("train_nonfn2.txt", "nonfn2_train.txt")):
    if os.path.exists(os.path.join(R, src)):
        shutil.copyfile(os.path.join(R, src), os.path.join(OUT, f"{tag}_{dst}"))
k2 = os.path.join(R, "bench2", "bench2_kernel_stats.csv")
if os.path.exists(k2):
    rows2 = kernel_stats(k2)
    tot2 = sum(float(r["TotalDurationNs"]) for r in rows2)
    own2 = sum(float(r["TotalDurationNs"]) for r in rows2 if "fn2::" in r["Name"])
    with open(os.path.join(OUT, f"{tag}_flownet2_kernels.md"), "w") as fh:
        fh.write("# FlowNet2 (CSS + SD + fusion) deploy forward, batch 4 @768x384: kernel table\n\n"
                 "`rocprofv3 --kernel-trace --stats -- python bench.py --net 2 --batch 4 --height 384 --width 768 --steps 10 --warmup 3` (13 steps in the trace).\n"
                 "`fn2::` kernels: %.1f %% of the GPU time.\n\n| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|\n" % (100.0 * own2 / tot2))
        for r in rows2[:40]:
            fh.write("| `%s` | %s | %.2f | %.2f | %s |\n" % (short(r["Name"], 90), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
# matrix-pipe counters of the convolution kernels (scripts/conv_bench.py under --pmc)
pmc_conv = os.path.join(R, "pmc_conv", "conv_counter_collection.csv")
if os.path.exists(pmc_conv):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc_conv)):
        if "fn2::cv::" in r["Kernel_Name"] or "fn2::wino::" in r["Kernel_Name"]:
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    cl = ["# Matrix-pipe and LDS counters of the convolution kernels (round %s)" % tag, "",
          "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python scripts/conv_bench.py --net C --layers conv2,conv3_1`",
          "(FN2_AUTOTUNE=0: every tile variant is launched by the script itself).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).", "",
          "| kernel | dispatches | GUI cycles / launch | MFMA instructions | MFMA busy % | LDS bank-conflict cycles / LDS active |", "|---|---|---|---|---|---|"]
    for name, c in sorted(per.items()):
        gui_k = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        insts = sum(c["SQ_INSTS_MFMA"]) / len(c["SQ_INSTS_MFMA"])
        conf = sum(c.get("SQ_LDS_BANK_CONFLICT", [0])) / max(1, len(c.get("SQ_LDS_BANK_CONFLICT", [0])))
        act = sum(c.get("SQ_LDS_IDX_ACTIVE", [0])) / max(1, len(c.get("SQ_LDS_IDX_ACTIVE", [0])))
        cl.append("| `%s` | %d | %.0f | %.0f | %.1f | %.3f |" % (short(name.replace("void ", ""), 80), len(c["GRBM_GUI_ACTIVE"]), gui_k, insts,
                                                              100 * busy / (1024.0 * gui_k) if gui_k else 0, conf / act if act else 0))
    open(os.path.join(OUT, f"{tag}_conv_counters.md"), "w").write("\n".join(cl) + "\n")

# the correlation backward (both bottoms, one launch since round 6): kernel-trace average of the same micro-benchmark run + its own counter pass
bwd = None
bwd_rows = [r for r in kernel_stats(os.path.join(R, "corr", "corr_kernel_stats.csv")) if "corr_bwd" in r["Name"]]
if bwd_rows:
    bwd_us = sum(float(r["AverageNs"]) for r in bwd_rows) / 1e3          # (one merged kernel, or the two per-bottom kernels of earlier rounds)
    bwd_flops = 2 * 2.0 * 256 * 441 * 8 * 40 * 56
    bwd = {"kernels": [short(r["Name"].split("(")[0], 80) for r in bwd_rows], "avg_us_kernel_trace_both_bottoms": bwd_us,
           "algorithmic_flops_both_bottoms": bwd_flops, "tflops": bwd_flops / (bwd_us * 1e-6) / 1e12,
           "frac_of_fp32_mfma_peak": bwd_flops / (bwd_us * 1e-6) / 1e12 / 157.3}
    pb = os.path.join(R, "pmc_sq_bwd", "corr_counter_collection.csv")
    if os.path.exists(pb):
        sb = counters(pb, "corr_bwd")
        gb = sb.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if gb:
            bwd.update({"mfma_util": sb["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gb), "sq_insts_mfma": sb.get("SQ_INSTS_MFMA"),
                        "lds_bank_conflict_over_active": (sb.get("SQ_LDS_BANK_CONFLICT", 0.0) / sb["SQ_LDS_IDX_ACTIVE"]) if sb.get("SQ_LDS_IDX_ACTIVE") else None})
    lines_b = ["", "## Correlation backward (`%s`) at [8,256,40,56], both bottoms" % ", ".join(bwd["kernels"]), "",
               "%.2f us per call in the kernel trace above = %.1f TFLOP/s = **%.3f** of the fp32 MFMA peak (8.09 GFLOP algorithmic)%s."
               % (bwd_us, bwd["tflops"], bwd["frac_of_fp32_mfma_peak"],
                  ("; matrix pipes %.1f %% busy, %.0f MFMA instructions, LDS bank-conflict / active cycles %.2f (`--pmc` pass of its own)"
                   % (100 * bwd["mfma_util"], bwd["sq_insts_mfma"], bwd["lds_bank_conflict_over_active"] or 0.0)) if "mfma_util" in bwd else "")]
    md = os.path.join(OUT, f"{tag}_rocprof_summary.md")
    open(md, "a").write("\n".join(lines_b) + "\n")

summary = {"tag": tag, "kernel": "corr_fwd_units [8,256,40,56]", "avg_us_kernel_trace": corr_avg_us, "backward": bwd,
           "FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"], "fetch_correction": fetch_factor,
           "write_correction": write_factor, "hbm_read_bytes": fetch_bytes, "hbm_write_bytes": write_bytes,
           "traffic_bytes_per_launch": fetch_bytes + write_bytes, "algorithmic_bytes_per_launch": alg,
           "mfma_util": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui) if gui else None,
           "sq_insts_mfma": sq.get("SQ_INSTS_MFMA"), "executed_flops_per_launch": sq.get("SQ_INSTS_MFMA", 0.0) * 2048.0,
           "source": "profiles/%s_rocprof_summary.md (kernel-trace avg; --pmc passes of their own)" % tag}
json.dump(summary, open(os.path.join(OUT, f"{tag}_corr_hbm.json"), "w"), indent=1)
print(open(os.path.join(OUT, f"{tag}_rocprof_summary.md")).read()[:6000])
