#!/bin/bash
# SQ counter passes over the correlation BACKWARD kernel (round 6: corr_bwd_g4_both; FN2_CORR_IMPL=15 / 16 for the per-bottom kernels of
# generations 3 / 4), each set in its own rocprofv3 run.
export TMPDIR=/tmp
R=gpurun_out/pmc_bwd
mkdir -p $R
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $R/$n -o c -- python scripts/corr_microbench.py --iters 10 --backward --impl ${FN2_CORR_IMPL:-0} > /dev/null 2> $R/$n.err
done
python - <<'PY'
import csv, glob, collections
for which in ("corr_bwd_g4_both", "corr_bwd_g4<0>", "corr_bwd_g4<1>", "corr_bwd_g3<0>", "corr_bwd_g3<1>"):
    rows = []
    for f in sorted(glob.glob("gpurun_out/pmc_bwd/*/c_counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if which in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows += ["  %-28s %.4g" % (k, sum(v) / len(v)) for k, v in acc.items()]
    if rows:
        print(which)
        print("\n".join(rows))
PY
