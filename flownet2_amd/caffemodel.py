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


def load_file(path: str):
    with open(path, "rb") as f:
        return read_caffemodel(f.read())


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
