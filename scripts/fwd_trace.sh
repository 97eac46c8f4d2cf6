# kernel trace of the FlowNetC forward step:  bash scripts/fwd_trace.sh <tag> [bench args]
set -u
export TMPDIR=/tmp
TAG=$1; shift
R=gpurun_out/$TAG
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/fwd -o f -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --corr-iters 4 "$@" > $R/fwd_profiled.json 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/fwd/f_kernel_stats.csv")))
steps = 25 + 4 * 8       # warm-up + timed + untimed settling steps are not known exactly: normalise by the stem kernel's calls
for r in rows:
    if "conv_k7s2_relu" in r["Name"]: steps = int(r["Calls"])
tot = sum(float(r["TotalDurationNs"]) for r in rows if "corr_fwd_" not in r["Name"])
print("steps %d, GPU time per step %.3f ms (correlation kernel excluded)" % (steps, tot / 1e6 / steps))
for r in rows[:40]:
    print("%-110s %5.1f/step %8.1f us avg %8.1f us/step" % (r["Name"][:110], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
PY
