""".caffemodel.h5 reader (csrc/hdf5_reader.cpp, flownet2_amd/caffemodel.py; reference: Net::CopyTrainedLayersFromHDF5 net.cpp:823-882,
hdf5_load_nd_dataset util/hdf5.cpp:9-79).  Pins, all CPU:
  * the three .h5 files the reference's own tests hold (src/caffe/test/test_data/, values by generate_sample_data.py; copied to
    tests/golden/ as data fixtures): contiguous float32, and chunked + gzip float32 / uint8;
  * files written by the REAL libhdf5 1.10.6 through the calls Net::ToHDF5 makes (tests/golden/make_h5_fixtures.py, tests/golden/h5/):
    the HDF5 twin of tests/golden/tiny.caffemodel (default and libver=latest layouts), 40 layers (several symbol-table nodes), and the
    other dataset forms hdf5_load_nd_dataset accepts (big-endian float64 + shuffle + gzip with edge chunks, int16, compact, unwritten);
  * the net-level semantics of the HDF5 route (no CustomCopyBlobs, missing blob of a shared parameter, the ".h5" suffix dispatch);
  * this package's writer (Net::ToHDF5 layout) read back by the reader, and -- where a libhdf5 is installed -- by libhdf5 itself."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from flownet2_amd import _lib, caffemodel, net as fnet, nets
from flownet2_amd._lib import Fn2Error
from flownet2_amd.layers import CheckError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
H5 = os.path.join(GOLD, "h5")


def datasets(path):
    """{absolute path: float32 array} of every dataset, through the C ABI; also returns the entries."""
    L = _lib.lib()
    buf = np.fromfile(path, dtype=np.uint8)
    n = C.c_int()
    assert L.fn2_hdf5_index(buf.ctypes.data, buf.size, None, 0, C.byref(n)) == 0, L.fn2_last_error_string()
    ent = (_lib.Hdf5Entry * max(1, n.value))()
    assert L.fn2_hdf5_index(buf.ctypes.data, buf.size, ent, n.value, C.byref(n)) == 0
    out, meta = {}, {}
    for e in list(ent)[:n.value]:
        a = np.empty(e.count, np.float32)
        assert L.fn2_hdf5_read_float(buf.ctypes.data, buf.size, C.byref(e), a.ctypes.data, a.size) == 0, L.fn2_last_error_string()
        out[e.path.decode()] = a.reshape(tuple(e.dim[k] for k in range(e.num_axes)))
        meta[e.path.decode()] = (e.type_class, e.type_size, e.big_endian, e.layout, e.num_filters)
    return out, meta


def test_reference_held_fixtures_read_as_generate_sample_data_wrote_them():
    # generate_sample_data.py:13-31: data = arange(10*8*6*5) as [10,8,6,5] float32, label = 1 + arange(10)[:, None], label2 = label + 1
    total = 10 * 8 * 6 * 5
    data = np.arange(total, dtype=np.float32).reshape(10, 8, 6, 5)
    label = (1 + np.arange(10, dtype=np.float32))[:, None]
    d, meta = datasets(os.path.join(GOLD, "sample_data.h5"))
    assert list(d) == ["/data", "/label", "/label2"]
    assert np.array_equal(d["/data"], data) and np.array_equal(d["/label"], label) and np.array_equal(d["/label2"], label + 1)
    assert all(m[3] == 1 and m[4] == 0 for m in meta.values())                             # contiguous, no filters
    # :33-47: data + total_size, gzip level 1 (chunked); labels stored as uint8 -- H5LTread_dataset_float converts, so does the reader
    d, meta = datasets(os.path.join(GOLD, "sample_data_2_gzip.h5"))
    assert np.array_equal(d["/data"], data + total) and np.array_equal(d["/label"], label) and np.array_equal(d["/label2"], label + 1)
    assert meta["/data"] == (1, 4, 0, 2, 1) and meta["/label"][:2] == (0, 1) and meta["/label"][3:] == (2, 1)
    # :51-68: random normal data [8,3,10,10] and targets [8,1] (values unknown; statistics of a standard normal, finite)
    d, _ = datasets(os.path.join(GOLD, "solver_data.h5"))
    assert d["/data"].shape == (8, 3, 10, 10) and d["/targets"].shape == (8, 1)
    assert np.isfinite(d["/data"]).all() and abs(float(d["/data"].mean())) < 0.1 and 0.9 < float(d["/data"].std()) < 1.1


@pytest.mark.parametrize("fn", ["tiny.caffemodel.h5", "tiny_latest.caffemodel.h5"])
def test_libhdf5_written_twin_of_the_binaryproto_fixture(fn):
    want = dict(np.load(os.path.join(GOLD, "tiny_caffemodel.npz")))
    layers = caffemodel.load_file(os.path.join(H5, fn))                                   # ".h5" suffix -> the HDF5 reader (net.cpp:804-811)
    assert isinstance(layers, caffemodel.Hdf5Layers) and layers.route == "hdf5"
    assert list(layers) == ["conv1", "deconv5", "fuse_conv0", "img0s_aug", "net2_conv6"]  # H5_INDEX_NAME order, not file order
    for name, keys in [("conv1", (".w", ".b")), ("deconv5", (".w", ".b")), ("fuse_conv0", (".w",)), ("img0s_aug", (".count", ".pixel_mean", ".mean"))]:
        assert len(layers[name]["blobs"]) == len(keys)
        for blob, k in zip(layers[name]["blobs"], keys):
            w = want[name + k]
            assert blob.shape == w.shape and np.array_equal(blob.view(np.uint32), w.view(np.uint32)), name + k
    assert np.array_equal(layers["net2_conv6"]["blobs"][1], want["net2_conv6.b"].reshape(2))
    # the same parameter dict as the binaryproto twin gives
    template = {"conv1.w": np.zeros((4, 3, 3, 3)), "conv1.b": np.zeros(4), "deconv5.w": np.zeros((4, 2, 4, 4)), "deconv5.b": np.zeros(2)}
    p_h5, means_h5, _ = caffemodel.to_params(layers, template)
    p_pb, means_pb, _ = caffemodel.to_params(caffemodel.load_file(os.path.join(GOLD, "tiny.caffemodel")), template)
    assert sorted(p_h5) == sorted(p_pb) and all(np.array_equal(p_h5[k], p_pb[k]) for k in p_pb)
    assert np.array_equal(means_h5["img0s_aug"], means_pb["img0s_aug"])


def test_many_layers_span_several_symbol_table_nodes_and_sort_by_name():
    want = dict(np.load(os.path.join(H5, "many_layers.npz")))
    layers = caffemodel.load_file(os.path.join(H5, "many_layers.caffemodel.h5"))
    assert len(layers) == 40 and list(layers) == sorted(layers)                            # "Conv0" < "Conv12" < ... < "layer_01" (byte order)
    assert list(layers)[:3] == ["Conv0", "Conv12", "Conv15"]
    for name, l in layers.items():
        for j, b in enumerate(l["blobs"]):
            assert np.array_equal(b, want["%s/%d" % (name, j)])


def test_other_dataset_forms_hdf5_load_nd_dataset_accepts():
    want = {k.replace("|", "/"): v for k, v in np.load(os.path.join(H5, "odd_types.npz")).items()}
    d, meta = datasets(os.path.join(H5, "odd_types.h5"))
    assert sorted(d) == sorted(want)
    for k in want:
        assert d[k].shape == want[k].shape and np.array_equal(d[k], want[k]), k
    assert meta["/f64be_gzip_shuffle"] == (1, 8, 1, 2, 2)       # float64, big-endian, chunked (2x4x8 over 5x7x11: partial chunks), shuffle + deflate
    assert meta["/i16_chunked"][:2] == (0, 2) and meta["/f32_compact"][3] == 0 and meta["/u32be"][:3] == (0, 4, 1)
    # a file without a `data` group is not a weight file (net.cpp:829-830)
    with pytest.raises(Fn2Error, match="no group 'data'"):
        caffemodel.load_file(os.path.join(H5, "odd_types.h5"))


def test_malformed_files_are_refused():
    L = _lib.lib()
    raw = np.fromfile(os.path.join(H5, "tiny.caffemodel.h5"), dtype=np.uint8)
    n = C.c_int()

    def index(buf):
        buf = np.ascontiguousarray(buf)
        return L.fn2_hdf5_index(buf.ctypes.data, buf.size, None, 0, C.byref(n))

    assert index(np.zeros(4096, np.uint8)) != 0 and b"not an HDF5 file" in L.fn2_last_error_string()
    assert index(raw[:200]) != 0                                                           # superblock only: the root group is gone
    assert index(raw[:len(raw) // 2]) != 0                                                 # truncated behind the first objects
    bad = raw.copy()
    bad[8] = 9                                                                             # unknown superblock version
    assert index(bad) != 0 and b"superblock version" in L.fn2_last_error_string()
    with pytest.raises(Fn2Error):
        caffemodel.read_caffemodel_h5(bytes(raw[:len(raw) // 2]))
    # a destination of the wrong size is an error, not an overrun
    ent = (_lib.Hdf5Entry * 16)()
    assert L.fn2_hdf5_index(raw.ctypes.data, raw.size, ent, 16, C.byref(n)) == 0
    a = np.empty(ent[0].count + 1, np.float32)
    assert L.fn2_hdf5_read_float(raw.ctypes.data, raw.size, C.byref(ent[0]), a.ctypes.data, a.size) != 0


def _libhdf5():
    for p in ("/opt/conda/lib/libhdf5.so.103", "libhdf5.so", "libhdf5_serial.so"):
        try:
            h5 = C.CDLL(p, mode=C.RTLD_GLOBAL)
            hl = C.CDLL(os.path.join(os.path.dirname(p), "libhdf5_hl.so") if "/" in p else p.replace("libhdf5", "libhdf5_hl"))
            return h5, hl
        except OSError:
            continue
    return None


def test_writer_round_trip_and_libhdf5_reads_what_it_writes(tmp_path):
    rng = np.random.default_rng(5)
    P = {k: v.numpy() for k, v in nets.init_params("C", seed=1).items() if k.startswith(("conv1.", "conv_redir.", "Convolution1.", "upsample_flow6to5."))}
    layers = {}
    for k, v in P.items():
        layers.setdefault(k[:-2], []).append(v)
    layers["img0s_aug"] = [np.array([2000.0], np.float32), rng.random((1, 3, 4, 4)).astype(np.float32), rng.random((1, 3, 1, 1)).astype(np.float32)]
    for i in range(30):                                                                    # > 8 links: several symbol-table nodes
        layers["extra_%02d" % i] = [rng.standard_normal((3, 2, 1, 1)).astype(np.float32)]
    path = str(tmp_path / "w.caffemodel.h5")
    caffemodel.write_caffemodel_h5(path, layers)
    back = caffemodel.load_file(path)
    assert list(back) == sorted(layers)
    for name, blobs in layers.items():
        assert len(back[name]["blobs"]) == len(blobs)
        for a, b in zip(blobs, back[name]["blobs"]):
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    libs = _libhdf5()
    if libs is None:
        pytest.skip("no libhdf5 on this box (the writer is checked against it in the build container)")
    h5, hl = libs
    hid = C.c_int64
    h5.H5Fopen.restype, h5.H5Fopen.argtypes = hid, [C.c_char_p, C.c_uint, hid]
    h5.H5Fclose.argtypes = [hid]
    hl.H5LTread_dataset_float.argtypes = [hid, C.c_char_p, C.c_void_p]
    hl.H5LTget_dataset_ndims.argtypes = [hid, C.c_char_p, C.POINTER(C.c_int)]
    f = h5.H5Fopen(path.encode(), 0, 0)
    assert f >= 0
    for name, blobs in layers.items():
        for j, a in enumerate(blobs):
            nd, out = C.c_int(), np.empty(a.size, np.float32)
            assert hl.H5LTget_dataset_ndims(f, ("data/%s/%d" % (name, j)).encode(), C.byref(nd)) >= 0 and nd.value == a.ndim
            assert hl.H5LTread_dataset_float(f, ("data/%s/%d" % (name, j)).encode(), out.ctypes.data) >= 0
            assert np.array_equal(out, a.ravel())
    h5.H5Fclose(f)


NET = """
name: "t"
input: "x" input_shape { dim: 1 dim: 3 dim: 8 dim: 8 }
layer { name: "aug" type: "DataAugmentation" bottom: "x" top: "xa" augmentation_param { recompute_mean: %d mean_per_pixel: false } }
layer { name: "ca" type: "Convolution" bottom: "xa" top: "a" param { name: "w" } param { name: "b" } convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
layer { name: "cb" type: "Convolution" bottom: "xa" top: "b" param { name: "w" } param { name: "b" } convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
layer { name: "cc" type: "Convolution" bottom: "a" top: "c" convolution_param { num_output: 2 kernel_size: 1 } }
"""


def test_net_level_semantics_of_the_hdf5_route(tmp_path):
    rng = np.random.default_rng(9)
    w, b = rng.standard_normal((4, 3, 3, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32)
    wc, bc = rng.standard_normal((2, 4, 1, 1)).astype(np.float32), rng.standard_normal(2).astype(np.float32)
    mean_px, mean_ch = rng.random((1, 3, 8, 8)).astype(np.float32), rng.random((1, 3, 1, 1)).astype(np.float32)
    src = {"aug": [np.array([77.0], np.float32), mean_px, mean_ch], "ca": [w, b.reshape(1, 1, 1, 4)], "cb": [], "cc": [wc, bc], "unknown_layer": [w]}
    path = str(tmp_path / "m.caffemodel.h5")
    caffemodel.write_caffemodel_h5(path, src)
    n = fnet.Net(NET % 0, phase="TEST", device="cpu")
    assert n.CopyTrainedLayersFrom(path) == ["unknown_layer"]                              # "Ignoring source layer" (net.cpp:835-838)
    ca, cb, cc, aug = (n.layer_by_name(k) for k in ("ca", "cb", "cc", "aug"))
    assert np.array_equal(ca.blobs_[0].data.numpy(), w) and np.array_equal(ca.blobs_[1].data.numpy(), b)       # [1,1,1,4] dataset -> [4] blob
    assert cb.blobs_[0] is ca.blobs_[0]                                                    # `cb` shares both parameters: its missing blobs are fine (:859-862)
    assert np.array_equal(cc.blobs_[0].data.numpy(), wc)
    # no CustomCopyBlobs on this route: the three blobs arrive although recompute_mean is 0 (the binaryproto route leaves them alone)
    assert aug.num_iter_ == 77 and np.array_equal(aug.mean_channel_.numpy(), mean_ch.reshape(3)) and np.array_equal(aug.mean_pixel_.numpy(), mean_px[0])
    n2 = fnet.Net(NET % 0, phase="TEST", device="cpu")
    n2.CopyTrainedLayersFrom({k: {"blobs": v} for k, v in src.items() if k not in ("cb",)})
    assert n2.layer_by_name("aug").num_iter_ == 0 and n2.layer_by_name("aug").mean_channel_ is None
    # a layer that owns its parameters must find every blob (:863-866); more links than blobs is refused too (:849-850)
    caffemodel.write_caffemodel_h5(path, {"cc": [wc]})
    with pytest.raises(CheckError, match="Incompatible number of blobs for layer cc"):
        fnet.Net(NET % 0, phase="TEST", device="cpu").CopyTrainedLayersFrom(path)
    caffemodel.write_caffemodel_h5(path, {"cc": [wc, bc, bc]})
    with pytest.raises(CheckError, match="Incompatible number of blobs for layer cc"):
        fnet.Net(NET % 0, phase="TEST", device="cpu").CopyTrainedLayersFrom(path)
    caffemodel.write_caffemodel_h5(path, {"cc": [w, bc]})
    with pytest.raises(CheckError, match="Cannot copy param 0"):
        fnet.Net(NET % 0, phase="TEST", device="cpu").CopyTrainedLayersFrom(path)
    # the dispatch is on the NAME (net.cpp:804-811): the same bytes under a name that does not end in ".h5" go to the protobuf reader
    assert caffemodel.is_hdf5_name("x.caffemodel.h5") and caffemodel.is_hdf5_name("a.h5") and not caffemodel.is_hdf5_name("x.caffemodel") and not caffemodel.is_hdf5_name("h5")


@pytest.mark.gpu
def test_run_flownet_with_h5_weights_writes_the_flo_of_the_binaryproto_twin(tmp_path):
    """scripts/run_flownet.py on a FlowNetC `.caffemodel.h5` against the same weights as a binaryproto `.caffemodel`: the two .flo files are
    equal byte for byte (both the built-in graph and the prototxt runner)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_authors_prototxt import TEMPLATE, v1_caffemodel
    from PIL import Image
    P = nets.init_params("C", seed=4)
    pb, h5p = str(tmp_path / "c.caffemodel"), str(tmp_path / "c.caffemodel.h5")
    open(pb, "wb").write(v1_caffemodel(P))
    caffemodel.write_caffemodel_h5(h5p, caffemodel.load_file(pb))
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (128, 192, 3), dtype=np.uint8)
    Image.fromarray(a).save(str(tmp_path / "a.png"))
    Image.fromarray(np.roll(a, (2, -3), (0, 1))).save(str(tmp_path / "b.png"))
    outs = {}
    for tag, wts in (("pb", pb), ("h5", h5p)):
        for form in ("builtin", "prototxt"):
            out = str(tmp_path / f"{tag}_{form}.flo")
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_flownet.py")]
            cmd += ["--net", "C", "--weights", wts] if form == "builtin" else [wts, TEMPLATE]
            subprocess.check_call(cmd + [str(tmp_path / "a.png"), str(tmp_path / "b.png"), out], cwd=ROOT)
            outs[tag, form] = open(out, "rb").read()
    assert outs["pb", "builtin"] == outs["h5", "builtin"] and outs["pb", "prototxt"] == outs["h5", "prototxt"]
    assert len(outs["h5", "builtin"]) == 12 + 128 * 192 * 8
