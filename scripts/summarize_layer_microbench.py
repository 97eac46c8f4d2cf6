#!/usr/bin/env python
"""profiles/<tag>_layer_microbench.md from `rocprofv3 --kernel-trace --stats -d gpurun_out/layers -o lay -- python scripts/layer_microbench.py`
plus the script's own (host-timed) table in gpurun_out/layer_microbench.txt."""
import csv, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "layers", "lay_kernel_stats.csv"))))
out = [f"# Custom-layer micro-benchmarks (round {tag})", "",
       "`rocprofv3 --kernel-trace --stats -- python scripts/layer_microbench.py` on one MI355X, inputs resident in HBM, SURVEY section 8(d) sizes.",
       "Kernel durations (every launch of the run; a kernel that serves several workloads shows their mix):", "",
       "| kernel | launches | avg us | min us | max us |", "|---|---|---|---|---|"]
for r in rows:
    if "fn2::" in r["Name"]:
        name = r["Name"].split("(")[0].replace("void ", "")
        out.append("| `%s` | %s | %.2f | %.2f | %.2f |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
txt = os.path.join(ROOT, "gpurun_out", "layer_microbench.txt")
if os.path.exists(txt):
    out += ["", "Per-workload table of the same script, timed from the host with HIP events around 100 back-to-back calls (FlowWarp backward = 3 kernels + 2",
            "memsets; blobs of a few MB are bounded below by the ~10-15 us Python launch path, not by the kernel):", ""]
    out += [l.rstrip() for l in open(txt) if l.startswith("|")]
    extra = [l.rstrip() for l in open(txt) if l.startswith(("CPU decode", "reference CustomDataLayer", "host -> device"))]
    if extra:
        out += ["", "Host side of the sample decode on the same box (`oracle/_ref` = the reference's own CustomData layer compiled in place):", ""] + ["* " + l for l in extra]
out += ["", "Notes: FlowWarp with an i.i.d. random flow makes every lane touch its own cache line (texture-path bound); the smooth field",
        "(bilinear up-sampling of a coarse random field, same magnitude) is what a network predicts.  Resample keeps the reference's 25 taps",
        "per output including the zero-weight ones (a NaN there poisons the output in the reference too)."]
open(os.path.join(ROOT, "profiles", f"{tag}_layer_microbench.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
