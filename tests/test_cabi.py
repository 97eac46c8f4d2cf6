"""CPU tests: the C-ABI library loads and exports every symbol include/flownet2_hip.h declares; the
host-side Layer mirror enforces the reference's CHECKs; .flo I/O is byte-exact.  No compute calls."""
import os
import re
import struct

import numpy as np
import pytest
import torch

import flownet2_amd
from flownet2_amd import _lib, flo, layers, ops
from flownet2_amd.layers import Blob, CheckError, LayerParameter, LayerRegistry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "flownet2_hip.h")).read()
    declared = set(re.findall(r"\b(fn2_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fn2_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/flownet2_hip.h but not exported"
    assert set(_lib.EXPORTS) == declared
    assert "gfx950" in flownet2_amd.version()


def test_oracle_exports_cpu_twins():
    import oracle
    L = oracle.lib()
    for name in _lib.EXPORTS:
        if name in ("fn2_version", "fn2_last_error_string") or name.endswith("workspace_bytes") or name.endswith("_supported") \
                or name.endswith("_num_variants") or name.startswith("fn2_debug_set_") or name.endswith("_ksplit") \
                or name.endswith("_batch_invariant") or name.endswith("_sync_bytes"):
            # (tile-variant hooks; ksplit is a launch-geometry query whose value the twins take as an argument)
            continue
        if name in ("fn2_conv_route", "fn2_deconv_route", "fn2_conv_forward", "fn2_deconv_forward") or name.startswith(("fn2_conv_pack", "fn2_deconv_pack", "fn2_conv_backward_data", "fn2_conv_backward_weights")):
            # descriptor-level dispatchers (csrc/conv_route.cpp): no arithmetic of their own -- every kernel they route to has its twin
            continue
        if name.startswith("fn2_hdf5_"):
            # the HDF5 file format is libhdf5's (a third-party dependency of the reference, absent from its tree): the reader is pinned on
            # files libhdf5 1.10.6 wrote and on the reference's own .h5 fixtures (tests/test_hdf5.py), not on a second restatement
            continue
        assert hasattr(L, name + "_cpu"), name + "_cpu"


def test_predict_flow_backward_band_respects_lds():
    """The weight-gradient band of predict_flow's backward is chosen among the heights that fit 60 KB of LDS (host logic, no GPU): FlowNetC at
    1024x512 batch 8 (Convolution5 head: C = 194, 128x256) used to pick 32 rows = 70 KB and fail mid-training."""
    L = _lib.lib()
    assert L.fn2_predict_flow_conv_backward_supported(8, 194, 128, 256) == 1
    assert L.fn2_predict_flow_conv_backward_supported(8, 194, 80, 112) == 1 and L.fn2_predict_flow_conv_backward_supported(1, 18, 12, 1000) == 1
    assert L.fn2_predict_flow_conv_backward_supported(1, 16, 8, 3000) == 0 and L.fn2_predict_flow_conv_backward_supported(0, 16, 8, 8) == 0
    # workspace = (18 C + 2) floats per (sample, band): 128 rows of 256 pixels take 16-row bands (8 * 18 * 258 * 4 = 148 KB for 32 rows)
    assert L.fn2_predict_flow_conv_backward_workspace_bytes(8, 194, 128, 256) == 4 * 8 * 8 * (18 * 194 + 2)


def test_shape_function_matches_reference_reshape_and_rejects_bad_params():
    # FlowNetC: correlation_layer.cpp:52-73 with K=1, md=20, pad=20, s1=1, s2=2
    assert ops.correlation_out_shape(ops.corr_params(20, 1, 20, 1, 2), 256, 40, 56) == (441, 40, 56)
    assert ops.correlation_out_shape(ops.corr_params(3, 3, 2, 2, 1), 7, 8, 10) == (25, 4, 5)
    for bad in [ops.corr_params(4, 2, 4, 1, 1), ops.corr_params(1, 1, 4, 1, 1), ops.corr_params(0, 1, 0, 0, 1),
                ops.corr_params(0, 9, 0, 1, 1)]:
        with pytest.raises(flownet2_amd.Fn2Error):
            ops.correlation_out_shape(bad, 3, 8, 8)
    import oracle
    for args in [(20, 1, 20, 1, 2, 256, 40, 56), (3, 3, 2, 2, 1, 7, 8, 10), (5, 3, 4, 1, 2, 4, 10, 9), (6, 1, 4, 1, 2, 2, 6, 6)]:
        pad, K, md, s1, s2, C, H, W = args
        assert ops.correlation_out_shape(ops.corr_params(pad, K, md, s1, s2), C, H, W) == \
            oracle.correlation_out_shape(oracle.corr_params(pad, K, md, s1, s2), C, H, W)


def test_correlation1d_shape_function_matches_reference_reshape():
    import oracle
    # DispNetCorr1D parameters: correlation_layer1d.cpp:55-78 with K=1, md=40, pad=40, s1=1, s2=1, left only
    assert ops.correlation1d_out_shape(ops.corr_params(40, 1, 40, 1, 1, single_direction=-1), 256, 48, 96) == (41, 48, 96)
    assert ops.correlation1d_out_shape(ops.corr_params(40, 1, 40, 1, 1), 256, 48, 96) == (81, 48, 96)
    assert ops.correlation1d_out_shape(ops.corr_params(3, 3, 2, 2, 1, single_direction=1), 7, 9, 14) == (3, 4, 7)   # no padding in y
    for bad in [ops.corr_params(4, 2, 4, 1, 1), ops.corr_params(4, 1, 4, 1, 1, single_direction=2), ops.corr_params(0, 1, 0, 0, 1),
                ops.corr_params(0, 1, 4, 1, 1)]:
        with pytest.raises(flownet2_amd.Fn2Error):
            ops.correlation1d_out_shape(bad, 3, 8, 8)
    for args in [(40, 1, 40, 1, 1, -1, 8, 6, 30), (3, 3, 2, 2, 1, 1, 7, 9, 14), (5, 3, 4, 1, 2, 0, 4, 8, 15), (0, 1, 2, 1, 1, 0, 3, 6, 9)]:
        pad, K, md, s1, s2, sd, C, H, W = args
        assert ops.correlation1d_out_shape(ops.corr_params(pad, K, md, s1, s2, single_direction=sd), C, H, W) == \
            oracle.correlation1d_out_shape(oracle.corr_params(pad, K, md, s1, s2, 0, 0, sd), C, H, W)
    x = torch.zeros(1, 3, 4, 4)
    with pytest.raises(ValueError, match="no CPU path"):
        ops.correlation1d_forward(ops.corr_params(1, 1, 1, 1, 1), x, x)


def test_ops_refuse_cpu_tensors_no_fallback():
    x = torch.zeros(1, 3, 4, 4)
    with pytest.raises(ValueError, match="no CPU path"):
        ops.channel_norm_forward(x)
    with pytest.raises(ValueError, match="no CPU path"):
        ops.correlation_forward(ops.corr_params(1, 1, 1, 1, 1), x, x)


def test_layer_registry_and_blob_count_checks():
    assert LayerRegistry.LayerTypeList() == ["ChannelNorm", "Concat", "Convolution", "Correlation", "Correlation1D", "CustomData", "DataAugmentation",
                                             "Deconvolution", "Downsample", "Eltwise", "FlowAugmentation", "FlowWarp", "GenerateAugmentationParameters",
                                             "Input", "L1Loss", "ReLU", "Resample", "Silence", "Slice", "Split"]
    with pytest.raises(CheckError, match="Unknown layer type"):
        LayerRegistry.CreateLayer(LayerParameter(type="Nope"))
    with pytest.raises(CheckError, match="already registered"):
        layers.REGISTER_LAYER_CLASS("Correlation", layers.CorrelationLayer)
    corr = LayerRegistry.CreateLayer(LayerParameter(type="Correlation", correlation_param=dict(kernel_size=1, max_displacement=2, pad=2)))
    b = [Blob(1, 3, 6, 6, device="cpu")]
    with pytest.raises(CheckError, match="takes 2 bottom"):
        corr.SetUp(b, [Blob(device="cpu")])
    with pytest.raises(CheckError, match="kernel_size is not set"):
        LayerRegistry.CreateLayer(LayerParameter(type="Correlation", correlation_param=dict(max_displacement=2))).SetUp(b * 2, [Blob(device="cpu")])
    with pytest.raises(CheckError, match="Odd kernel size"):
        LayerRegistry.CreateLayer(LayerParameter(type="Correlation", correlation_param=dict(kernel_size=2, max_displacement=2))).SetUp(b * 2, [Blob(device="cpu")])
    with pytest.raises(CheckError, match="same width"):
        corr.SetUp([Blob(1, 3, 6, 6, device="cpu"), Blob(1, 3, 6, 7, device="cpu")], [Blob(device="cpu")])
    top = Blob(device="cpu")
    corr.SetUp([Blob(2, 3, 6, 6, device="cpu"), Blob(2, 3, 6, 6, device="cpu")], [top])
    assert top.shape() == [2, 25, 6, 6]
    warp = LayerRegistry.CreateLayer(LayerParameter(type="FlowWarp"))
    with pytest.raises(CheckError, match="2 channels"):
        warp.SetUp([Blob(1, 3, 4, 4, device="cpu"), Blob(1, 3, 4, 4, device="cpu")], [Blob(device="cpu")])
    rs = LayerRegistry.CreateLayer(LayerParameter(type="Resample", resample_param=dict(type="AREA", width=4, height=4)))
    with pytest.raises(CheckError, match="only CUBIC, LINEAR and NEAREST"):
        rs.SetUp([Blob(1, 3, 4, 4, device="cpu")], [Blob(device="cpu")])
    rs = LayerRegistry.CreateLayer(LayerParameter(type="Resample", resample_param=dict(width=8, height=6)))
    t = Blob(device="cpu")
    rs.SetUp([Blob(1, 3, 4, 4, device="cpu")], [t])
    assert t.shape() == [1, 3, 6, 8] and rs.layer_param_.reshape_every_iter is False and not rs.AllowBackward()
    with pytest.raises(CheckError, match="cannot do backward"):
        rs.Backward([t], [True], [Blob(1, 3, 4, 4, device="cpu")])
    l1 = LayerRegistry.CreateLayer(LayerParameter(type="L1Loss"))
    with pytest.raises(CheckError, match="at most 2"):
        l1.SetUp([Blob(1, 2, 4, 4, device="cpu")] * 3, [Blob(device="cpu")])


def test_flo_bytes_roundtrip_on_synthetic_blobs(tmp_path):      # the reference-held fixtures: tests/test_flo_fixtures.py
    rng = np.random.default_rng(0)
    blob = rng.standard_normal((2, 5, 7)).astype(np.float32)          # [2,H,W] like predict_flow_final
    p = str(tmp_path / "a.flo")
    flo.write_flo(p, blob)
    raw = open(p, "rb").read()
    assert raw[:4] == b"PIEH" and struct.unpack("<f", raw[:4])[0] == 202021.25
    assert struct.unpack("<ii", raw[4:12]) == (7, 5)
    body = np.frombuffer(raw[12:], "<f4").reshape(5, 7, 2)
    # writeFloFile (output.cpp:47-65): for y, for x: u = data[y*W+x], v = data[y*W+x+H*W]
    assert np.array_equal(body[..., 0], blob[0]) and np.array_equal(body[..., 1], blob[1])
    back = flo.read_flo(p)
    assert np.array_equal(back, blob.transpose(1, 2, 0))
    flo.write_flo(p, back)                                             # (H,W,2) form, as run-flownet.py writeFlow
    assert open(p, "rb").read() == raw
    with open(p, "wb") as f:
        f.write(b"XXXX" + raw[4:])
    with pytest.raises(ValueError, match="PIEH"):
        flo.read_flo(p)
    with open(p, "wb") as f:
        f.write(raw[:-4])
    with pytest.raises(ValueError, match="corrupted"):
        flo.read_flo(p)
