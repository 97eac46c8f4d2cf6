export TMPDIR=/tmp
mkdir -p gpurun_out/r04w
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/r04w/pytest.txt 2>&1
tail -6 gpurun_out/r04w/pytest.txt
timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/r04w/train.err | tail -1 | cut -c1-330
