""".caffemodel -> parameter dict of flownet2_amd.nets.

What the reference does with `caffe.Net(deploy.prototxt, model.caffemodel, TEST)` (scripts/run-flownet.py:64-68 ->
Net::CopyTrainedLayersFromBinaryProto, src/caffe/net.cpp:752-819): layers are matched BY NAME, blobs by index, shapes CHECKed;
source layers the net does not have are ignored.  The wire walk is the C host function fn2_caffemodel_index /
fn2_caffemodel_read_blob (csrc/caffemodel.cpp); this module maps the result onto the names nets.py uses:
    Convolution / Deconvolution layer `<name>`  ->  `<name>.w` (blob 0), `<name>.b` (blob 1)
    DataAugmentation layers (DoesUseCustomCopyBlobs, data_augmentation_layer.cpp:162-205: blobs = iteration count, per-pixel mean,
    per-channel mean)  ->  means[`<name>`] = blob 2, the per-channel mean the deploy nets subtract (mean_per_pixel: false).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from . import _lib
from ._lib import CaffemodelEntry, Fn2Error


def _check(rc):
    if rc != 0:
        raise Fn2Error(rc, _lib.lib().fn2_last_error_string().decode())


def read_caffemodel(data: bytes) -> "OrderedDict[str, dict]":
    """-> {layer name: {"type": str or V1 enum, "blobs": [float32 arrays in blob shape]}} in file order (layers without blobs are
    not listed: they carry nothing CopyTrainedLayersFrom would copy)."""
    L = _lib.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    ptr = C.c_void_p(buf.ctypes.data)
    n = C.c_int()
    _check(L.fn2_caffemodel_index(ptr, buf.size, None, 0, C.byref(n)))
    entries = (CaffemodelEntry * max(1, n.value))()
    _check(L.fn2_caffemodel_index(ptr, buf.size, entries, n.value, C.byref(n)))
    out: "OrderedDict[str, dict]" = OrderedDict()
    for i in range(n.value):
        e = entries[i]
        name = bytes(buf[e.name_off:e.name_off + e.name_len]).decode("utf-8")
        typ = bytes(buf[e.type_off:e.type_off + e.type_len]).decode("utf-8") if not e.v1 else int(e.v1_type)
        arr = np.empty(e.count, np.float32)
        _check(L.fn2_caffemodel_read_blob(ptr, buf.size, C.byref(e), arr.ctypes.data_as(C.c_void_p), arr.size))
        shape = tuple(int(e.dim[k]) for k in range(e.num_axes))
        layer = out.setdefault(name, {"type": typ, "blobs": []})
        assert e.blob_index == len(layer["blobs"]) or name in out, "blobs arrive in index order"
        layer["blobs"].append(arr.reshape(shape))
    return out


class Hdf5Layers(OrderedDict):
    """{layer name: {"type": None, "blobs": [...]}} read from a `.caffemodel.h5`: a distinct type because Net::CopyTrainedLayersFromHDF5
    (net.cpp:823-882) treats the blobs differently from the binaryproto route -- no CustomCopyBlobs for DataAugmentation layers, a missing
    trailing blob is fine for a shared parameter, the source may hold FEWER blobs than the layer."""
    route = "hdf5"


def read_caffemodel_h5(data: bytes) -> Hdf5Layers:
    """The `data/<layer name>/<blob index>` datasets of an HDF5 weight file (Net::ToHDF5's layout, net.cpp:896-950) in the order
    CopyTrainedLayersFromHDF5 visits them (layer groups by name, H5_INDEX_NAME; blobs by index).  Datasets outside `data/` (the `diff`
    group a solver snapshot carries) and non-numeric dataset names are not listed.  Integer and float64 datasets come back as float32
    like H5LTread_dataset_float converts them (util/hdf5.cpp:56-63)."""
    L = _lib.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    ptr = C.c_void_p(buf.ctypes.data)
    n = C.c_int()
    _check(L.fn2_hdf5_index(ptr, buf.size, None, 0, C.byref(n)))
    entries = (_lib.Hdf5Entry * max(1, n.value))()
    _check(L.fn2_hdf5_index(ptr, buf.size, entries, n.value, C.byref(n)))
    has_data_group = False
    found = {}
    for i in range(n.value):
        e = entries[i]
        parts = e.path.decode("utf-8").split("/")               # "", "data", layer, index
        has_data_group |= len(parts) >= 2 and parts[1] == "data"
        if len(parts) != 4 or parts[1] != "data" or not parts[3].isdigit() or str(int(parts[3])) != parts[3]:
            continue
        arr = np.empty(e.count, np.float32)
        _check(L.fn2_hdf5_read_float(ptr, buf.size, C.byref(e), arr.ctypes.data_as(C.c_void_p), arr.size))
        found.setdefault(parts[2], {})[int(parts[3])] = arr.reshape(tuple(int(e.dim[k]) for k in range(e.num_axes)))
    if not has_data_group and n.value:
        raise Fn2Error(-1, "Error reading weights: the HDF5 file has no group 'data'")                  # net.cpp:830
    out = Hdf5Layers()
    for name in sorted(found):                              # byte order = H5_INDEX_NAME order
        idx = found[name]
        # the reference probes "0", "1", ... per TARGET blob (net.cpp:853-872); a hole ends what an index-ordered list can express
        blobs = []
        while len(blobs) in idx:
            blobs.append(idx[len(blobs)])
        out[name] = {"type": None, "blobs": blobs, "num_links": len(idx)}
    return out


def is_hdf5_name(path: str) -> bool:
    """Net::CopyTrainedLayersFrom's dispatch (net.cpp:804-811): a file name that ends in ".h5" goes to the HDF5 reader."""
    return len(path) >= 3 and path.endswith(".h5")


def load_file(path: str):
    with open(path, "rb") as f:
        data = f.read()
    return read_caffemodel_h5(data) if is_hdf5_name(path) else read_caffemodel(data)


def write_caffemodel_h5(path: str, layers) -> None:
    """Net::ToHDF5 with write_diff = false (net.cpp:896-950): group `data`, a group per layer, its blobs as contiguous float32 datasets
    "0", "1", ... (hdf5_save_nd_dataset -> H5LTmake_dataset_float, util/hdf5.cpp:81-101).  `layers`: {layer name: [arrays]} or the dict
    read_caffemodel / read_caffemodel_h5 return.  A minimal writer of the classic file layout (superblock v0, v1 object headers,
    symbol-table groups: what libhdf5 writes with default libver bounds); tests/test_hdf5.py reads its files back with this package's
    reader AND checks them against files libhdf5 itself wrote."""
    from . import _h5write
    norm = OrderedDict()
    for name, v in layers.items():
        blobs = v["blobs"] if isinstance(v, dict) else v
        norm[name] = {str(j): np.ascontiguousarray(b, np.float32) for j, b in enumerate(blobs)}
    _h5write.write(path, {"data": norm})


def _same_shape(src: np.ndarray, want: Tuple[int, ...]) -> bool:
    """Blob::ShapeEquals (blob.cpp:419-448): equal shapes, or a legacy <= 4-D blob whose leading axes are 1 (bias [1,1,1,C] vs [C])."""
    if tuple(src.shape) == tuple(want):
        return True
    if src.ndim == 4 and len(want) <= 4:
        return tuple(src.shape) == (1,) * (4 - len(want)) + tuple(want)
    return False


def to_params(layers, template: Dict[str, "object"]) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray], List[str]]:
    """Maps layer blobs onto the `<layer>.w` / `<layer>.b` names of `template` (a nets.init_params*-style dict giving the target
    shapes).  Returns (params, means, ignored): params holds only the names found in the file; a shape mismatch raises like the
    reference's LOG(FATAL) "Cannot copy param ... shape mismatch" (net.cpp:783-793); source layers the template does not know are
    listed in `ignored` ("Ignoring source layer", :763)."""
    params, means, ignored = {}, {}, []
    for name, layer in layers.items():
        blobs = layer["blobs"]
        if layer["type"] == "DataAugmentation" or (len(blobs) == 3 and blobs[0].size == 1 and blobs[2].ndim == 4 and blobs[2].shape[2:] == (1, 1)):
            means[name] = blobs[2].reshape(-1).astype(np.float32)
            continue
        if name + ".w" not in template:
            ignored.append(name)
            continue
        for suffix, blob in zip((".w", ".b"), blobs):
            want = tuple(template[name + suffix].shape)
            if not _same_shape(blob, want):
                raise ValueError(f"Cannot copy param {suffix} weights from layer '{name}'; shape mismatch.  Source param shape is "
                                 f"{tuple(blob.shape)}; target param shape is {want}.")
            params[name + suffix] = blob.reshape(want)
        if len(blobs) not in (1, 2):
            raise ValueError(f"Incompatible number of blobs for layer {name}")          # net.cpp:779-780
    return params, means, ignored
