export TMPDIR=/tmp
mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_conv_plane.py -q -m gpu -k "k5s2" 2>&1 | grep -E "AssertionError|variant|passed|failed|Error" | head -40 > gpurun_out/r04q/pytest.txt 2>&1
cat gpurun_out/r04q/pytest.txt
