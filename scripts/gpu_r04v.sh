export TMPDIR=/tmp
mkdir -p gpurun_out/r04v
timeout 300 python scripts/probes/backward_repeat_probe.py 8x320x448 2>&1 | grep -v "amdgpu.ids"
timeout 300 python scripts/probes/backward_repeat_trace.py 2x128x192 2>&1 | grep "DIFFERENT\|differing"
timeout 900 python -m pytest tests/test_train_parity.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/r04v/train.err | tail -1 | cut -c1-330
