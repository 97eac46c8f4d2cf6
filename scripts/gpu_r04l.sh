export TMPDIR=/tmp
mkdir -p gpurun_out/r04l
timeout 600 python scripts/probes/small_layer_routes.py > gpurun_out/r04l/routes_b1.txt 2>&1
cat gpurun_out/r04l/routes_b1.txt
