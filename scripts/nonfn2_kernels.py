#!/usr/bin/env python
"""Which kernels of a rocprofv3 --kernel-trace --stats run are NOT this repository's (no `fn2::` in the name): the library-dispatch audit of
SURVEY row a12.  usage: nonfn2_kernels.py <kernel_stats.csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
# steps: given, or the calls of a kernel that runs once per step (the 3-channel stem of FlowNetC, alone or inside FlowNet2)
once = [int(r["Calls"]) for r in rows if "stem_wgrad" in r["Name"]] or [int(r["Calls"]) for r in rows if "conv_k7s2_relu<3" in r["Name"]]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else float(max(once) if once else 1)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
own = sum(float(r["TotalDurationNs"]) for r in rows if "fn2::" in r["Name"])
print("kernels: %d distinct; GPU time %.3f ms (%.3f ms / step over %g steps); fn2:: %.2f %%" % (len(rows), tot / 1e6, tot / 1e6 / steps, steps, 100 * own / tot))
print("not fn2:: (calls, calls / step, total us, %):")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "fn2::" in r["Name"]:
        continue
    print("  %6d %8.2f %10.1f %6.3f  %s" % (int(r["Calls"]), int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
