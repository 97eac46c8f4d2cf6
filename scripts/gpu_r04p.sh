export TMPDIR=/tmp
mkdir -p gpurun_out/r04p
(time timeout 900 python -m pytest tests/test_conv_plane.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r04p/pytest.txt 2>&1
cat gpurun_out/r04p/pytest.txt
timeout 600 python scripts/probes/small_layer_routes.py > gpurun_out/r04p/routes_b1.txt 2>&1
grep -E "conv2 |conv3 |sum" gpurun_out/r04p/routes_b1.txt
timeout 600 python bench.py --net 2 --no-extras --no-cpu-baseline --batch 1 --height 448 --width 1024 --steps 30 --warmup 5 2>gpurun_out/r04p/b1.err | tail -1 | cut -c1-400
