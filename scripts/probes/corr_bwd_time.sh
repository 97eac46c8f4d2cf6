#!/bin/bash
# per-kernel average of the correlation backward kernels at config A under rocprofv3 (1500 forward / 300 backward launches):
#   bash scripts/probes/corr_bwd_time.sh <tag> [impl]
set -u
TAG=${1:-corrbwd}; IMPL=${2:-0}
R=gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o c -- python scripts/corr_microbench.py --iters 1500 --backward --impl $IMPL > $R/stdout_$IMPL.txt 2>&1
f=$(find $R/t -name "*_kernel_stats.csv" | head -1)
[ -n "$f" ] && python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'corr' in r['Name']: print('impl $IMPL  %-40s calls %s  avg %.2f us' % (r['Name'].split('(')[0].replace('void fn2::',''), r['Calls'], float(r['AverageNs'])/1e3))
"
rm -rf $R/t
