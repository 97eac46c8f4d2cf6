# kernel trace + SQ counters of the weight-gradient kernel on chosen layers:  bash scripts/wgrad_pmc.sh <tag> <layers> [buffers] [chunk]
set -u
export TMPDIR=/tmp
TAG=$1; LAYERS=$2; DB=${3:-1}; XT=${4:-0}
R=gpurun_out/$TAG
mkdir -p $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -o t -- python scripts/wgrad_bench.py $LAYERS $DB $XT > $R/trace_stdout.txt 2>/dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/pmc1 -o p -- python scripts/wgrad_bench.py $LAYERS $DB $XT > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/pmc2 -o p -- python scripts/wgrad_bench.py $LAYERS $DB $XT > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$R/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fn2" in r["Name"] or "igemm" in r["Name"] or "transpose" in r["Name"]:
            print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
for d in ("pmc1", "pmc2"):
    for f in glob.glob("$R/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "conv_wgrad" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                print("   %-28s %16.0f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
