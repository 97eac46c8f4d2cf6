export TMPDIR=/tmp
mkdir -p gpurun_out/r04i
timeout 300 python scripts/probes/small_map_grad_accuracy.py > gpurun_out/r04i/grad_accuracy.txt 2>&1
cat gpurun_out/r04i/grad_accuracy.txt
