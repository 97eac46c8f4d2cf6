export TMPDIR=/tmp
R=gpurun_out/r04n; mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
( timeout 900 python -m pytest tests/test_prototxt.py tests/test_authors_prototxt.py tests/test_gpu_parity.py -m gpu -q -x -k "bits or flownet2 or full_size or nets_py or epe or slice_path" ) > $R/pytest.txt 2>&1
tail -3 $R/pytest.txt
for hs in 1 0 1 0; do
  FN2_HEAD_STREAM=$hs python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('FlowNetC head_stream=$hs', d['value'], d['ms_per_step'], d['ms_per_step_p10_p50_p90'])"
done
for hs in 1 0; do
  FN2_HEAD_STREAM=$hs python bench.py --net 2 --batch 4 --height 384 --width 768 --steps 30 --warmup 6 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('FlowNet2 b4 head_stream=$hs', d['value'], d['ms_per_step'])"
  FN2_HEAD_STREAM=$hs python bench.py --net 2 --batch 1 --height 448 --width 1024 --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('FlowNet2 b1 head_stream=$hs', d['value'], d['ms_per_step'])"
done
