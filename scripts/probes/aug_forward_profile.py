import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import numpy as np, torch
from flownet2_amd import augment
from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
import train_pipeline as tp
B, H, W, cw, ch = 8, 384, 512, 448, 320
aug_p = dict(crop_width=cw, crop_height=ch, mean=[0.411, 0.433, 0.45], mean_per_pixel=False)
aug0 = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", augmentation_param=aug_p))
img = torch.rand(B, 3, H, W, device="cuda")
p0 = augment.draw_batch(augment.make_rng(1, 0), tp.AUG0, B, W, H, cw, ch, discount=1.0)
b0 = [Blob.from_tensor(img), Blob.from_tensor(torch.from_numpy(p0).view(B, 42, 1, 1))]
b0[1].data = torch.from_numpy(p0).view(B, 42, 1, 1)
t = [Blob()]
aug0.SetUp(b0, t)
for _ in range(3): aug0.Forward(b0, t)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): aug0.Forward(b0, t)
torch.cuda.synchronize()
print("Forward: %.3f ms per call (host + device)" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): aug0.Forward(b0, t)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
