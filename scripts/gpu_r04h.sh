set -u
export TMPDIR=/tmp
R=gpurun_out/r04h
mkdir -p $R
timeout 300 python scripts/probes/train_grad_trace.py > $R/train_grad_trace.txt 2>&1
( time timeout 600 python -m pytest tests/test_stem_wgrad.py -m gpu -q -x ) > $R/pytest_stem.txt 2>&1
python - > $R/stem_wgrad_time.txt 2>&1 <<'P'
import torch, sys
sys.path.insert(0, '.')
from flownet2_amd import ops
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((16, 3, 320, 448), device="cuda", generator=g); d = torch.randn((16, 64, 160, 224), device="cuda", generator=g)
w = torch.zeros((64, 3, 7, 7), device="cuda")
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
print("own stem wgrad  %.1f us" % t(lambda: ops.conv_k7s2_wgrad(d, x)))
print("library wgrad   %.1f us" % t(lambda: torch.ops.aten.convolution_backward(d, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])))
P
cat $R/train_grad_trace.txt | tail -40; tail -3 $R/pytest_stem.txt; cat $R/stem_wgrad_time.txt
