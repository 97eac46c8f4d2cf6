""".caffemodel reader (csrc/caffemodel.cpp, flownet2_amd/caffemodel.py; reference: Net::CopyTrainedLayersFrom net.cpp:752-800,
Blob::FromProto blob.cpp:459-508).  CPU tests: the committed fixture (written by the protobuf runtime with the REFERENCE's own
descriptor, tests/golden/make_caffemodel_fixture.py) read back bit for bit by the product reader and by the oracle twin; where
/root/reference exists, fresh messages are serialised on the fly as well (unpacked encodings, chunked data, truncation)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle
from flownet2_amd import _lib, caffemodel, nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
RAW = open(os.path.join(GOLD, "tiny.caffemodel"), "rb").read()
WANT = dict(np.load(os.path.join(GOLD, "tiny_caffemodel.npz")))


def oracle_read(data):
    L = oracle.lib()
    buf = np.frombuffer(data, np.uint8)
    n = C.c_int()
    L.fn2_caffemodel_index_cpu.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_lib.CaffemodelEntry), C.c_int, C.POINTER(C.c_int)]
    L.fn2_caffemodel_read_blob_cpu.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_lib.CaffemodelEntry), C.c_void_p, C.c_size_t]
    assert L.fn2_caffemodel_index_cpu(buf.ctypes.data, buf.size, None, 0, C.byref(n)) == 0
    ent = (_lib.CaffemodelEntry * n.value)()
    assert L.fn2_caffemodel_index_cpu(buf.ctypes.data, buf.size, ent, n.value, C.byref(n)) == 0
    out = []
    for e in ent:
        a = np.empty(e.count, np.float32)
        assert L.fn2_caffemodel_read_blob_cpu(buf.ctypes.data, buf.size, C.byref(e), a.ctypes.data, a.size) == 0
        out.append((bytes(buf[e.name_off:e.name_off + e.name_len]).decode(), e.blob_index, tuple(e.dim[:e.num_axes]), a, e.v1, e.v1_type))
    return out


def test_fixture_reads_back_bit_for_bit():
    layers = caffemodel.read_caffemodel(RAW)
    # wire order: protobuf writes fields by number, so the V1 `layers` (2) precede `layer` (100); ReLU1 has no blobs
    assert list(layers) == ["fuse_conv0", "img0s_aug", "conv1", "deconv5", "net2_conv6"]
    assert layers["conv1"]["type"] == "Convolution" and layers["img0s_aug"]["type"] == "DataAugmentation"
    assert layers["fuse_conv0"]["type"] == 4                                                   # V1LayerParameter.CONVOLUTION
    for name, keys in [("conv1", (".w", ".b")), ("deconv5", (".w", ".b")), ("net2_conv6", (".w", ".b")), ("fuse_conv0", (".w",))]:
        for blob, k in zip(layers[name]["blobs"], keys):
            want = WANT[name + k]
            assert blob.shape == want.shape and np.array_equal(blob.view(np.uint32), want.view(np.uint32)), name + k
    for blob, k in zip(layers["img0s_aug"]["blobs"], ("count", "pixel_mean", "mean")):
        assert np.array_equal(blob, WANT["img0s_aug." + k])
    # the oracle twin sees the same entries
    flat = [(n, i, b) for n, l in layers.items() for i, b in enumerate(l["blobs"])]
    twin = oracle_read(RAW)
    assert len(twin) == len(flat)
    for (n, i, b), (tn, ti, tshape, ta, _, _) in zip(flat, twin):
        assert (n, i, b.shape) == (tn, ti, tshape) and np.array_equal(b.ravel().view(np.uint32), ta.view(np.uint32))


def test_mapping_onto_net_parameters_follows_copy_trained_layers():
    layers = caffemodel.read_caffemodel(RAW)
    template = {"conv1.w": np.zeros((4, 3, 3, 3)), "conv1.b": np.zeros(4), "deconv5.w": np.zeros((4, 2, 4, 4)), "deconv5.b": np.zeros(2),
                "net2_conv6.w": np.zeros((2, 3, 1, 1)), "net2_conv6.b": np.zeros(2), "conv9.w": np.zeros((1, 1, 1, 1))}
    params, means, ignored = caffemodel.to_params(layers, template)
    assert ignored == ["fuse_conv0"]                                       # "Ignoring source layer" (net.cpp:763)
    assert sorted(params) == ["conv1.b", "conv1.w", "deconv5.b", "deconv5.w", "net2_conv6.b", "net2_conv6.w"]
    assert params["net2_conv6.b"].shape == (2,)                            # legacy [1,1,1,2] blob matches a [2] parameter (ShapeEquals)
    assert np.array_equal(params["net2_conv6.b"], WANT["net2_conv6.b"].reshape(2))
    assert np.array_equal(means["img0s_aug"], WANT["img0s_aug.mean"].reshape(3))
    bad = dict(template, **{"conv1.w": np.zeros((4, 3, 5, 5))})
    with pytest.raises(ValueError, match="shape mismatch"):
        caffemodel.to_params(layers, bad)


def test_flownet_parameter_names_round_trip_through_a_caffemodel(tmp_path):
    """Every FlowNet2 parameter name of nets.py (prefixes net2_/net3_/netsd_/fuse_) written as layers and read back."""
    raw = bytearray()

    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7f
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def ld(field, payload):
        return varint((field << 3) | 2) + varint(len(payload)) + payload

    P = nets.init_params("S", seed=1)
    small = {k: v.numpy()[..., :1, :1].copy() if v.dim() == 4 else v.numpy() for k, v in P.items()}     # keep the file small
    names = sorted({k[:-2] for k in small})
    for n in names:
        body = ld(1, n.encode()) + ld(2, b"Convolution")
        for suf in (".w", ".b"):
            a = np.ascontiguousarray(small[n + suf], np.float32)
            shape = ld(1, b"".join(varint(d) for d in a.shape))
            body += ld(7, ld(7, shape) + ld(5, a.tobytes()))
        raw += ld(100, body)
    layers = caffemodel.read_caffemodel(bytes(raw))
    params, means, ignored = caffemodel.to_params(layers, small)
    assert not ignored and not means and sorted(params) == sorted(small)
    for k in small:
        assert np.array_equal(params[k], small[k])


def test_malformed_files_are_refused():
    from flownet2_amd import Fn2Error
    for cut in (5, 40, len(RAW) - 3):
        with pytest.raises(Fn2Error):
            caffemodel.read_caffemodel(RAW[:cut])
    assert caffemodel.read_caffemodel(b"") == {}


@pytest.mark.skipif(not os.path.exists("/root/reference/python/caffe/proto/caffe_pb2.py"), reason="reference tree not present")
def test_against_the_protobuf_runtime_with_the_reference_descriptor():
    sys.path.insert(0, GOLD)
    import make_caffemodel_fixture as mk
    M = mk.reference_messages()
    net, arrays = mk.build(M["NetParameter"])
    assert net.SerializeToString() == RAW                                  # the committed fixture is what the script writes
    # a message protobuf would merge: data in two packed chunks + one unpacked element, shape.dim unpacked
    n2 = M["NetParameter"]()
    l = n2.layer.add(); l.name = "conv1"; l.type = "Convolution"
    raw = bytearray(n2.SerializeToString())
    a = np.arange(7, dtype=np.float32) * 0.5 - 1
    blob = (b"\x3a\x04\x08\x07" + b"\x2a\x0c" + a[:3].tobytes() + b"\x2d" + a[3:4].tobytes() + b"\x2a\x0c" + a[4:].tobytes())
    blob = blob.replace(b"\x3a\x04\x08\x07", b"\x3a\x02\x08\x07")       # BlobShape{dim: 7} with dim as a plain varint field
    layer = b"\x0a\x05conv1" + b"\x3a" + bytes([len(blob)]) + blob
    raw = b"\xa2\x06" + bytes([len(layer)]) + layer
    ref = M["NetParameter"].FromString(raw)
    assert list(ref.layer[0].blobs[0].data) == a.tolist() and list(ref.layer[0].blobs[0].shape.dim) == [7]
    got = caffemodel.read_caffemodel(raw)
    assert np.array_equal(got["conv1"]["blobs"][0], a)
    assert np.array_equal(oracle_read(raw)[0][3], a)
