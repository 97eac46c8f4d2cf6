set -u
export TMPDIR=/tmp
R=gpurun_out/r04c
mkdir -p $R
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
( time timeout 600 python -m pytest tests/test_stem_wgrad.py tests/test_conv_plane.py tests/test_authors_prototxt.py tests/test_augmentation_random.py -m gpu -q -x ) > $R/pytest_new.txt 2>&1
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "data_gradient or predict_flow_conv_backward" ) > $R/pytest_dgrad.txt 2>&1
( time timeout 600 python -m pytest tests/test_train_parity.py -m gpu -q -s ) > $R/pytest_train_parity.txt 2>&1
python - > $R/stem_wgrad_time.txt 2>&1 <<'P'
import torch, sys
sys.path.insert(0, '.')
from flownet2_amd import ops
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((16, 3, 320, 448), device="cuda", generator=g); d = torch.randn((16, 64, 160, 224), device="cuda", generator=g)
w = torch.zeros((64, 3, 7, 7), device="cuda")
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
print("own stem wgrad  %.1f us" % t(lambda: ops.conv_k7s2_wgrad(d, x)))
print("library wgrad   %.1f us" % t(lambda: torch.ops.aten.convolution_backward(d, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])))
P
( time timeout 600 python bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-extras ) > $R/bench_train.json 2> $R/bench_train.err
tail -3 $R/pytest_new.txt; tail -3 $R/pytest_dgrad.txt; grep "config-4\|passed\|failed" $R/pytest_train_parity.txt | tail -3; cat $R/stem_wgrad_time.txt; python -c "
import json;d=json.loads(open('$R/bench_train.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['ms_per_step_cold'])"
