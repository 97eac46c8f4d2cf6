"""RCCL on the hardware a 1-GPU box has: a process group with backend "nccl" (= RCCL on ROCm) and world_size 1 on cuda:0, and
parallel.GradientExchange forced through its FULL bucket path -- weight gradients produced in their bucket slots by the own kernels (on the
second HIP stream), post-accumulate hooks counting the buckets down, `all_reduce(async_op=True)` on RCCL's own stream ordered behind the
producing stream, `wait`, scale -- against the local step of the same process.  The reference tests its data-parallel solver on real
devices (src/caffe/test/test_gradient_based_solver.cpp:192-207, 458-483); P2PSync::on_gradients_ready is parallel.cpp:325-380.
Runs in a child process: the process group must not leak into the other tests of the session."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, socket, sys, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from flownet2_amd import functional as Fn, nets, parallel
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1, device_id=dev)
B, H, W = 2, 128, 192
P = {k: v.to(dev).requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
g = torch.Generator().manual_seed(3)
img0 = (torch.rand(B, 3, H, W, generator=g) - 0.43).to(dev)
img1 = (torch.rand(B, 3, H, W, generator=g) - 0.43).to(dev)
gt = (torch.randn(B, 2, H, W, generator=g) * 3).to(dev)
gt[:, :, :5, :7] = float("nan")
plist = list(P.values())

def run(ex, passes=1):
    ex.zero_grad()
    for _ in range(passes):
        loss = nets.multiscale_loss(nets.flownet_c_core(P, img0, img1, Fn), gt, Fn)
        loss.backward()
    n = ex.finish()
    torch.cuda.synchronize()
    return float(loss.detach()), [p.grad.clone() for p in plist], n

local = parallel.GradientExchange(plist, bucket_bytes=16 << 20)
assert not local.collective
l0, g0, _ = run(local)
l0b, g0b, _ = run(local)
local.remove()
assert l0 == l0b and all(torch.equal(a, b) for a, b in zip(g0, g0b)), "the local step is not reproducible"

ex = parallel.GradientExchange(plist, bucket_bytes=16 << 20, force_collective=True)
assert ex.collective and ex.world == 1 and len(ex.buckets) >= 4
out = {}
for rep in range(3):
    l1, g1, launched = run(ex)
    assert l1 == l0
    bad = [k for k, a, b in zip(P, g0, g1) if not torch.equal(a, b)]
    assert not bad, "gradients through the RCCL bucket path differ from the local step: %%s" %% bad[:4]
    total = sum(p.numel() * 4 for p in plist)
    # every gradient lives in its bucket slot, and the big ones were PRODUCED there: the hooks copied only biases / flow heads
    for p in plist:
        b = ex.buckets[ex._index[id(p)]]
        off, n = b["span"][id(p)]
        assert p.grad.data_ptr() == b["flat"].data_ptr() + 4 * off
    assert ex.copied_bytes <= 0.01 * total, (ex.copied_bytes, total)
    assert launched >= len(ex.buckets) - 1, (launched, len(ex.buckets))      # all but (at most) the last bucket left from inside backward
    out = {"buckets": len(ex.buckets), "launched_inside_backward": launched, "copied_bytes": ex.copied_bytes, "gradient_bytes": total}
# gradient accumulation (iter_size 2): two passes inside no_sync-style deferral add up in the slots, one exchange at the end
with ex.no_sync():
    ex.zero_grad()
    for _ in range(2):
        nets.multiscale_loss(nets.flownet_c_core(P, img0, img1, Fn), gt, Fn).backward()
ex.finish()
torch.cuda.synchronize()
for k, a, p in zip(P, g0, plist):
    assert torch.allclose(p.grad, 2 * a, rtol=1e-5, atol=1e-7), k
assert parallel.ranks_seen(dev) == 1
maps = open("/proc/self/maps").read()
out["librccl_mapped"] = "librccl" in maps
assert out["librccl_mapped"], "the nccl backend of this torch build is not RCCL?"
ex.remove()
dist.destroy_process_group()
print("RCCL_WORLD1_OK " + json.dumps(out))
"""


@pytest.mark.gpu
def test_gradient_exchange_runs_its_bucket_path_through_rccl_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
