export TMPDIR=/tmp
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-220
timeout 600 python bench.py --net 2 --batch 1 --height 448 --width 1024 --steps 60 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-220
