set -u
export TMPDIR=/tmp
R=gpurun_out/r03a
mkdir -p $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
( time timeout 900 python bench.py ) > $R/bench.json 2> $R/bench.err
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/train -o t -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/train_profiled.json 2>/dev/null
tail -5 $R/pytest.txt; tail -c 1500 $R/bench.json; tail -3 $R/bench.err
