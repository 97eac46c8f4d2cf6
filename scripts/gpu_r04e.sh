set -u
export TMPDIR=/tmp
R=gpurun_out/r04e
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
# warm the autotune cache for all four configurations (unprofiled)
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python bench.py --net 2 --batch 4 --height 384 --width 768 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python bench.py --net 2 --batch 1 --height 448 --width 1024 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
prof() { # tag, steps-in-trace, args
  local tag=$1 n=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$tag -o t -- python bench.py "$@" --no-cpu-baseline --no-extras --corr-iters 2 > $R/${tag}_bench.json 2>/dev/null
  python scripts/nonfn2_kernels.py $(find $R/$tag -name "t_kernel_stats.csv" | head -1) $n > $R/${tag}_nonfn2.txt 2>&1
}
prof fwdC 50 --steps 10 --warmup 3
prof fn2_b4 26 --net 2 --batch 4 --height 384 --width 768 --steps 10 --warmup 3
prof fn2_b1 26 --net 2 --batch 1 --height 448 --width 1024 --steps 10 --warmup 3
prof train 26 --mode train --steps 10 --warmup 3
for t in fwdC fn2_b4 fn2_b1 train; do echo "== $t"; head -12 $R/${t}_nonfn2.txt; done
( time timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/pytest_train_parity.txt 2>&1
grep -A60 "^parameter" $R/pytest_train_parity.txt | head -56; tail -3 $R/pytest_train_parity.txt
