set -u
export TMPDIR=/tmp
R=gpurun_out/r04b
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
( time timeout 600 python -m pytest tests/test_conv_wino.py tests/test_conv_plane.py -m gpu -q -x ) > $R/pytest_wino.txt 2>&1
timeout 300 python scripts/conv_bench.py --net C --layers conv3_1,conv4_1 > $R/conv_bench_wino.txt 2>&1
( time timeout 900 python bench.py --no-extras ) > $R/bench.json 2> $R/bench.err
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_train_parity.py ) > $R/pytest.txt 2>&1
( time timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/pytest_train_parity.txt 2>&1
tail -3 $R/pytest_wino.txt; grep -c "BITS DIFFER" $R/conv_bench_wino.txt; tail -4 $R/pytest.txt; grep "config-4\|passed\|failed" $R/pytest_train_parity.txt | tail -3; tail -c 1200 $R/bench.json
