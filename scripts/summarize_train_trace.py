"""Per-step kernel table of a traced training run (rocprofv3 --kernel-trace --stats): steps are counted by the calls of the Adam
multi-tensor kernel group / the correlation forward kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = None
for r in rows:
    if "corr_fwd_pair" in r["Name"]:
        steps = int(r["Calls"])
steps = steps or 1
# the correlation micro-benchmark of bench.py launches the forward kernel (10 + 200 + 8) more times
if steps > 218:
    steps -= 218
tot = sum(float(r["TotalDurationNs"]) for r in rows if "corr_fwd_pair" not in r["Name"])
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
    if "corr_fwd_pair" in r["Name"]:
        continue
    c = cat(r["Name"])
    cats[c] = cats.get(c, 0) + float(r["TotalDurationNs"])
print("steps %d; GPU time per step %.3f ms (correlation forward excluded: its micro-benchmark shares the trace)" % (steps, tot / 1e6 / steps))
for c, v in sorted(cats.items(), key=lambda x: -x[1]):
    print("  %-22s %8.3f ms/step %5.1f %%" % (c, v / 1e6 / steps, 100 * v / tot))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-104s %6.1f/step %9.1f us avg %8.1f us/step" % (r["Name"][:104], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
