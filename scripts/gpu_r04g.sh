set -u
export TMPDIR=/tmp
R=gpurun_out/r04g
mkdir -p $R
( FN2_CONV_WINO=0 FN2_WINO_BWD=none timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/parity_nowino.txt 2>&1
( FN2_WINO_BWD=none timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/parity_nowinobwd.txt 2>&1
for f in parity_nowino parity_nowinobwd; do echo "== $f"; grep -A50 "^parameter" $R/$f.txt | grep "conv1.w\|conv4_1.w\|deconv5.w\|conv6_1.w\|conv5.w\|conv3_1.w\|all param"; done
