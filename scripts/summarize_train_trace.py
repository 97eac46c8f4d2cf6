"""Per-step kernel table of a traced training run (rocprofv3 --kernel-trace --stats): steps are counted by the calls of the
stem kernels (one launch per step)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
# steps = calls of a kernel that runs exactly once per training step: the stem's weight gradient (round 4), else the stem's forward kernel
steps = None
for pat in ("stem_wgrad<", "conv_k7s2_relu<3"):       # ("stem_wgrad<": the kernel itself, not its two finalize launches)
    hit = [int(r["Calls"]) for r in rows if pat in r["Name"]]
    if hit:
        steps = max(hit)
        break
steps = steps or 1
tot = sum(float(r["TotalDurationNs"]) for r in rows if "corr_fwd_" not in r["Name"])
cats = {}


def cat(n):
    if "wgrad" in n or "pad_width" in n: return "own wgrad"
    if "tconv" in n: return "own tconv (dgrad)"
    if "igemm_wrw" in n: return "lib wrw"
    if "igemm_bwd" in n: return "lib bwd-data"
    if "igemm_fwd" in n: return "lib fwd"
    if "miopenSp3AsmConv" in n or "Winograd" in n or "winograd" in n: return "lib winograd"
    if "transpose" in n: return "lib transpose"
    if n.startswith("Cijk"): return "lib gemm"
    if "fn2::" in n: return "own other"
    if "multi_tensor_apply" in n: return "torch optimizer"
    if "at::native" in n: return "torch elementwise"
    return "other"


for r in rows:
    if "corr_fwd_" in r["Name"]:
        continue
    c = cat(r["Name"])
    cats[c] = cats.get(c, 0) + float(r["TotalDurationNs"])
print("steps %d; kernel time per step %.3f ms, SUMMED over the streams (correlation forward excluded: its micro-benchmark shares the trace).  Since round 5 the\n"
      "weight gradients run on a second HIP stream beside the data-gradient chain: the durations overlap -- a kernel that shares the chip runs longer than alone (the\n"
      "correlation backward: 90 us alone) -- and their sum exceeds the step time (bench.py --mode train, same round: profiles/<tag>_bench_train.json)." % (steps, tot / 1e6 / steps))
for c, v in sorted(cats.items(), key=lambda x: -x[1]):
    print("  %-22s %8.3f ms/step %5.1f %%" % (c, v / 1e6 / steps, 100 * v / tot))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-104s %6.1f/step %9.1f us avg %8.1f us/step" % (r["Name"][:104], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
