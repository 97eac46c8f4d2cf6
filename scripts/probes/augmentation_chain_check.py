"""DataAugmentation (TRAIN phase, own draws) -> GenerateAugmentationParameters (add) -> DataAugmentation (given coefficients) ->
FlowAugmentation through the Layer mirrors on the GPU: shapes and finiteness (the kernels themselves are pinned in tests/)."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_augmentation as T
from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry
img0 = Blob.from_tensor(torch.rand(4, 3, 384, 512, device="cuda")); img1 = Blob.from_tensor(torch.rand(4, 3, 384, 512, device="cuda"))
flow = Blob.from_tensor(torch.randn(4, 2, 384, 512, device="cuda"))
ap = dict(T.TRAIN_AUG, crop_width=448, crop_height=320, seed=1, mean=[0.4, 0.4, 0.4], mean_per_pixel=False, chromatic_eigvec=list(T.MG.EIGVEC))
a0 = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", phase="TRAIN", augmentation_param=ap))
t0 = [Blob(), Blob()]; a0.SetUp([img0], t0); a0.Forward([img0], t0)
gen = LayerRegistry.CreateLayer(LayerParameter(type="GenerateAugmentationParameters", phase="TRAIN", augmentation_param=dict(T.REL_AUG, mode="add", seed=2)))
tp = [Blob()]; gen.SetUp([t0[1], img0, t0[0]], tp); gen.Forward([t0[1], img0, t0[0]], tp)
a1 = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", phase="TRAIN", augmentation_param=dict(crop_width=448, crop_height=320, chromatic_eigvec=list(T.MG.EIGVEC))))
t1 = [Blob()]; a1.SetUp([img1, tp[0]], t1); a1.Forward([img1, tp[0]], t1)
fa = LayerRegistry.CreateLayer(LayerParameter(type="FlowAugmentation", augmentation_param=dict(crop_width=448, crop_height=320)))
tf = [Blob()]; fa.SetUp([flow, t0[1], tp[0]], tf); fa.Forward([flow, t0[1], tp[0]], tf)
print("ok", t0[0].shape(), t0[1].data.shape, t1[0].shape(), tf[0].shape(), bool(torch.isfinite(t0[0].data).all()), bool(torch.isfinite(tf[0].data).all()), float(t0[1].data.abs().sum()) > 0)
