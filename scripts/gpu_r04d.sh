set -u
export TMPDIR=/tmp
R=gpurun_out/r04d
mkdir -p $R
run() {  # name, env..., then bench args
  local name=$1; shift
  ( env "$@" python bench.py --net 2 $ARGS --no-cpu-baseline --no-extras ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %8.2f pairs/s  %8.4f ms/step  cold %8.4f' % ('$name', d['value'], d['ms_per_step'], d['ms_per_step_cold']))" >> $R/flownet2_routes.txt 2>&1
}
for cfg in "b1:--batch 1 --height 448 --width 1024 --steps 40 --warmup 8" "b4:--batch 4 --height 384 --width 768 --steps 30 --warmup 6"; do
  tag=${cfg%%:*}; ARGS=${cfg#*:}
  run "$tag lib(default)" FN2_X=0
  run "$tag own" FN2_CONV_SMALL=own
  run "$tag own maxpix2000" FN2_CONV_SMALL=own FN2_CONV_PLANE_MAXPIX=2000
  run "$tag own maxpix4000" FN2_CONV_SMALL=own FN2_CONV_PLANE_MAXPIX=4000
  run "$tag own maxpix8000" FN2_CONV_SMALL=own FN2_CONV_PLANE_MAXPIX=8000
  run "$tag lib maxpix4000" FN2_CONV_PLANE_MAXPIX=4000
done
cat $R/flownet2_routes.txt
