export TMPDIR=/tmp
mkdir -p gpurun_out/r04u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_ref_pin.py -q -m gpu -x -k "warp or l1loss or golden or reference" 2>&1 | tail -3
FN2_MB_ONLY="FlowWarp,L1Loss" timeout 600 python scripts/layer_microbench.py 2>&1 | grep -i "FlowWarp\|L1Loss" | tee gpurun_out/r04u/mb.txt
