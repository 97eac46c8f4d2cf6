# kernel trace of the FlowNetC training step:  bash scripts/train_trace.sh <tag>   -> gpurun_out/<tag>/train/
set -u
export TMPDIR=/tmp
TAG=$1
R=gpurun_out/$TAG
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/train -o t -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/train_profiled.json 2>/dev/null
python scripts/summarize_train_trace.py $R/train/t_kernel_stats.csv
