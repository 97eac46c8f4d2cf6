export TMPDIR=/tmp
mkdir -p gpurun_out/r04y
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
