# Where the wave cycles of the correlation forward go (SQ counters, two passes; counters only, no trace domains):  bash scripts/corr_stall_pmc.sh <tag>
set -u
export TMPDIR=/tmp
TAG=$1
R=gpurun_out/$TAG
mkdir -p $R
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $R/p1 -o c -- python scripts/corr_microbench.py --iters 10 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $R/p2 -o c -- python scripts/corr_microbench.py --iters 10 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$R/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "corr_fwd_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            print("%-26s %16.0f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
