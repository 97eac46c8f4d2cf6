set -u
export TMPDIR=/tmp
R=gpurun_out/r04j
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
( time timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/pytest_train_parity.txt 2>&1
( time timeout 600 python bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-extras ) > $R/bench_train.json 2> $R/bench_train.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/train -o t -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-extras --corr-iters 2 > /dev/null 2>&1
python scripts/nonfn2_kernels.py $(find $R/train -name "t_kernel_stats.csv" | head -1) 130 > $R/train_nonfn2.txt 2>&1
python scripts/summarize_train_trace.py $(find $R/train -name "t_kernel_stats.csv" | head -1) > $R/train_kernels.txt 2>&1
grep -B2 -A60 "^relative L2" $R/pytest_train_parity.txt | head -64; tail -3 $R/pytest_train_parity.txt
python -c "
import json;d=json.loads(open('$R/bench_train.json').read().strip().splitlines()[-1]);print('train', d['value'],d['ms_per_step'],d['ms_per_step_cold'])"
head -16 $R/train_nonfn2.txt; head -14 $R/train_kernels.txt
