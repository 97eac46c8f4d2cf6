"""Which calls of one FlowNetC training step make a strided COPY (Tensor.contiguous on a non-contiguous tensor, copy_): caller + shape."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flownet2_amd import functional as Fn, nets, parallel

N, H, W = 8, 320, 448
P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
ex = parallel.GradientExchange(list(P.values()))
g = torch.Generator().manual_seed(1)
img0, img1 = (torch.rand((N, 3, H, W), generator=g) - 0.43).cuda(), (torch.rand((N, 3, H, W), generator=g) - 0.43).cuda()
gt = (torch.randn((N, 2, H, W), generator=g) * 5).cuda()


def step():
    ex.zero_grad()
    loss = nets.multiscale_loss(nets.flownet_c_core(P, img0, img1, Fn), gt, Fn)
    loss.backward()
    ex.finish()


step(); step()
seen = collections.Counter()
orig_contig, orig_copy = torch.Tensor.contiguous, torch.Tensor.copy_


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "flownet2_amd" in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"


def contig(self, *a, **k):
    if self.is_cuda and not self.is_contiguous():
        seen[("contiguous", where(), tuple(self.shape))] += 1
    return orig_contig(self, *a, **k)


def copy_(self, src, *a, **k):
    if self.is_cuda:
        seen[("copy_", where(), tuple(self.shape))] += 1
    return orig_copy(self, src, *a, **k)


torch.Tensor.contiguous, torch.Tensor.copy_ = contig, copy_
step()
torch.Tensor.contiguous, torch.Tensor.copy_ = orig_contig, orig_copy
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
