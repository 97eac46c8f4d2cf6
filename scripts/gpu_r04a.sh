set -u
export TMPDIR=/tmp
R=gpurun_out/r04a
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_train_parity.py ) > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
( time timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/pytest_train_parity.txt 2>&1
( time timeout 900 python bench.py ) > $R/bench.json 2> $R/bench.err
timeout 300 python scripts/conv_bench.py --net C --layers conv3_1,conv4_1 > $R/conv_bench_wino.txt 2>&1
timeout 300 python scripts/conv_bench.py --net C --layers conv4,conv5,conv5_1,conv6,conv6_1 --only-plane > $R/conv_plane_bench.txt 2>&1
timeout 200 python scripts/tconv_bench.py --variants > $R/tconv_bench.txt 2>&1
tail -4 $R/pytest.txt; tail -5 $R/pytest_train_parity.txt; tail -c 2500 $R/bench.json; tail -3 $R/bench.err
