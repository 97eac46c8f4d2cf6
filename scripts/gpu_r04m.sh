export TMPDIR=/tmp
mkdir -p gpurun_out/r04m
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_train_parity.py -m gpu -q -s -k "fused_path_matches_stock or config4" ) > gpurun_out/r04m/pytest.txt 2>&1
grep "gradient agreement\|passed\|failed\|Error\|all parameters" gpurun_out/r04m/pytest.txt | head
