export TMPDIR=/tmp
mkdir -p gpurun_out/r04s
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resample" 2>&1 | tail -3
timeout 300 python scripts/probes/resample_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04s/resample_bench.txt
