export TMPDIR=/tmp
R=gpurun_out/r04o; mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
run() { local name=$1; shift; local envs=$1; shift
  ( env $envs python bench.py "$@" --no-cpu-baseline --no-extras ) 2>$R/err.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-34s %8.2f pairs/s %8.4f ms/step p50 %.4f launch=%s' % ('$name', d['value'], d['ms_per_step'], d['ms_per_step_p10_p50_p90'][1], d['config']['launch']))" || tail -3 $R/err.txt
}
B1="--net 2 --batch 1 --height 448 --width 1024 --steps 40 --warmup 8"
run "b1 own" FN2_X=0 $B1
run "b1 own graph" FN2_X=0 $B1 --graph
run "b1 lib" FN2_CONV_SMALL=lib $B1
run "b1 lib graph" FN2_CONV_SMALL=lib $B1 --graph
run "b4 own graph" FN2_X=0 --net 2 --batch 4 --height 384 --width 768 --steps 30 --warmup 6 --graph
run "C b8 graph" FN2_X=0 --steps 100 --warmup 10 --graph
run "C b1" FN2_X=0 --batch 1 --steps 100 --warmup 10
run "C b1 graph" FN2_X=0 --batch 1 --steps 100 --warmup 10 --graph
