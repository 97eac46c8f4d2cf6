"""The on-disk sample format of the training sets (LMDB values of the reference's CustomData layer).

Host side of SURVEY.md section 8f row 4, over the C ABI (include/flownet2_hip.h, "CustomData sample format"):
  * Datum wire format           <- src/caffe/proto/caffe.proto:30-41 (libprotobuf in the reference)
  * encode_sample / make_record <- ImagePair::read_data + the Datum the writer fills, tools/convert_imageset_and_flow.cpp:142-206, :231-236
  * decode_batch                <- DecodeData + CustomDataLayerPrefetch, src/caffe/layers/custom_data_layer.cpp:44-136, :209-300
    (raw bytes go to the GPU and a kernel decodes them; the reference decodes on one host thread and uploads fp32 blobs)
The storage engine itself (LMDB) is out of scope: records are (key, bytes) pairs from whatever reads them.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import DatumView, check

UINT8, UINT16FLOW, BOOL1 = 1, 2, 3          # DataParameter.CHANNELENCODING, caffe.proto:923-927
# what tools/convert_imageset_and_flow.cpp writes: Datum.channels = 9 (:231), two images, flow, occlusions
FLOW_SAMPLE_CHANNELS = 9
FLOW_SAMPLE_SLICE_POINTS = (3, 6, 8)
FLOW_SAMPLE_ENCODINGS = (UINT8, UINT8, UINT16FLOW, BOOL1)


@dataclass
class Datum:
    channels: int
    height: int
    width: int
    label: int
    encoded: bool
    data: Optional[bytes]             # field 4
    float_data: Optional[np.ndarray]  # field 6


def _ints(v: Sequence[int]):
    arr = (C.c_int * max(1, len(v)))(*[int(x) for x in v])
    return arr, len(v)


def _buf(b):
    """(address, length, keep-alive) of a bytes-like object."""
    a = np.frombuffer(b, dtype=np.uint8)
    return C.c_void_p(a.ctypes.data), a.size, a


def parse_datum(record) -> Datum:
    addr, n, keep = _buf(record)
    v = DatumView()
    check(_lib.lib().fn2_datum_parse(addr, n, C.byref(v)))
    data = None
    if v.data:
        off = v.data - addr.value
        data = bytes(keep[off: off + v.data_bytes])
    fl = None
    if v.float_data_count:
        fl = np.empty(v.float_data_count, np.float32)
        check(_lib.lib().fn2_datum_float_data(addr, n, C.c_void_p(fl.ctypes.data), fl.size))
    return Datum(v.channels, v.height, v.width, v.label, bool(v.encoded), data, fl)


def serialize_datum(channels: int, height: int, width: int, data: bytes, label: int = 0) -> bytes:
    addr, n, keep = _buf(data)
    L = _lib.lib()
    need = L.fn2_datum_serialize(channels, height, width, addr, n, label, None, 0)
    if need < 0:
        check(int(need))
    out = np.empty(need, np.uint8)
    got = L.fn2_datum_serialize(channels, height, width, addr, n, label, C.c_void_p(out.ctypes.data), out.size)
    if got < 0:
        check(int(got))
    return out.tobytes()


def sample_bytes(channels: int, H: int, W: int, slice_points: Sequence[int], encodings: Sequence[int]) -> int:
    sp, nsp = _ints(slice_points)
    en, nen = _ints(encodings)
    n = _lib.lib().fn2_custom_data_sample_bytes(channels, H, W, sp, nsp, en, nen)
    if n == 0:
        raise ValueError(f"invalid slicing: channels={channels} slice_point={list(slice_points)} encoding={list(encodings)}")
    return n


def encode_sample(img0_hwc: np.ndarray, img1_hwc: np.ndarray, flow_chw: Optional[np.ndarray] = None,
                  occlusion: Optional[np.ndarray] = None) -> bytes:
    """Images [H,W,3] uint8 as cv::imread returns them, flow [2,H,W] float32 (NaN = unknown), occlusion [H,W] (non-zero = occluded)."""
    a = np.ascontiguousarray(img0_hwc, np.uint8)
    b = np.ascontiguousarray(img1_hwc, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3 or a.shape != b.shape:
        raise ValueError("images must be [H,W,3] uint8 of the same size")
    H, W = a.shape[:2]
    f = o = None
    if flow_chw is not None:
        f = np.ascontiguousarray(flow_chw, np.float32)
        if f.shape != (2, H, W):
            raise ValueError("flow must be [2,H,W]")
    if occlusion is not None:
        o = np.ascontiguousarray(occlusion).astype(np.uint8)
        if o.shape != (H, W):
            raise ValueError("occlusion must be [H,W]")
    out = np.empty(10 * H * W + (H * W - 1) // 8 + 1, np.uint8)
    p = lambda x: C.c_void_p(x.ctypes.data) if x is not None else None
    check(_lib.lib().fn2_custom_data_encode_sample(p(a), p(b), p(f), p(o), H, W, p(out), out.size))
    return out.tobytes()


def make_record(img0_hwc, img1_hwc, flow_chw=None, occlusion=None, label: int = 0) -> bytes:
    """One LMDB value as the writer tool produces it (Datum{channels 9, height, width, label, data}, :231-236)."""
    H, W = np.asarray(img0_hwc).shape[:2]
    return serialize_datum(FLOW_SAMPLE_CHANNELS, H, W, encode_sample(img0_hwc, img1_hwc, flow_chw, occlusion), label)


def decode_batch(samples: torch.Tensor, channels: int, H: int, W: int, slice_points: Sequence[int] = (),
                 encodings: Sequence[int] = (), mean: Optional[torch.Tensor] = None, scale: float = 1.0,
                 float_data: bool = False) -> List[torch.Tensor]:
    """samples: CUDA uint8 tensor [N, stride] holding one Datum.data payload per row (or float32 [N, channels*H*W] with
    float_data=True).  Returns one float32 tensor [N, slice channels, H, W] per slice: (decoded - mean) * scale."""
    if not isinstance(samples, torch.Tensor) or not samples.is_cuda:
        raise ValueError("samples: expected a CUDA (HIP) tensor; flownet2_amd has no CPU path")
    want = torch.float32 if float_data else torch.uint8
    if samples.dtype != want or samples.dim() != 2:
        raise ValueError(f"samples: expected a 2-D {want} tensor, got {samples.dtype} {tuple(samples.shape)}")
    samples = samples.contiguous()
    N = samples.shape[0]
    stride = samples.shape[1] * samples.element_size()
    bounds = [0] + [int(s) for s in slice_points] + [channels]
    if any(b <= a for a, b in zip(bounds, bounds[1:])):
        raise ValueError(f"invalid slicing: channels={channels} slice_point={list(slice_points)}")
    tops = [torch.empty((N, b - a, H, W), device=samples.device, dtype=torch.float32) for a, b in zip(bounds, bounds[1:])]
    if mean is not None:
        if not mean.is_cuda or mean.dtype != torch.float32 or mean.numel() != channels * H * W:
            raise ValueError("mean: expected a CUDA float32 tensor of channels*H*W elements")
        mean = mean.contiguous()
    sp, nsp = _ints(slice_points)
    en, nen = _ints(encodings)
    ptrs = (C.c_void_p * len(tops))(*[t.data_ptr() for t in tops])
    check(_lib.lib().fn2_custom_data_decode_forward(
        C.c_void_p(samples.data_ptr()), stride, N, channels, H, W, sp, nsp, en, nen, int(float_data),
        C.c_void_p(mean.data_ptr()) if mean is not None else None, float(scale), ptrs,
        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return tops


def stage_records(records: Sequence[bytes], device="cuda", expect_bytes: Optional[int] = None):
    """Parses a batch of LMDB values on the host (header walk only) and uploads their raw `data` payloads: returns (samples uint8
    [N, stride] on the device, the first Datum's (channels, H, W), labels).  All records must have the same shape (the layer CHECKs
    that, :545).  One host copy per record, from the record into the page-locked staging buffer (fn2_custom_data_stage_records)."""
    L = _lib.lib()
    n = len(records)
    bufs = [np.frombuffer(r, dtype=np.uint8) for r in records]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    ch, h, w, nb = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    labels = (C.c_int * n)()
    check(L.fn2_custom_data_stage_records(ptrs, lens, n, None, 0, C.byref(ch), C.byref(h), C.byref(w), C.byref(nb), labels))
    if expect_bytes is not None and nb.value < expect_bytes:
        # a short payload would be decoded from whatever the reused staging buffer still holds behind it
        raise _lib.Fn2Error(_lib.FN2_ERR_INVALID_ARG if hasattr(_lib, "FN2_ERR_INVALID_ARG") else 1,
                            f"record holds {nb.value} data bytes, the slicing needs {expect_bytes} (custom_data_layer.cpp:86-136 reads them all)")
    stride = (nb.value + 15) // 16 * 16                         # keeps every sample 16-byte aligned
    host = _staging(n, stride)
    check(L.fn2_custom_data_stage_records(ptrs, lens, n, C.c_void_p(host.data_ptr()), stride, None, None, None, None, None))
    out = host.to(device, non_blocking=True)
    if out.is_cuda:
        torch.cuda.current_stream().synchronize()                 # the staging buffer is reused by the next batch
    return out, (ch.value, h.value, w.value), list(labels)


_STAGING = {}


def _staging(n: int, stride: int) -> torch.Tensor:
    """Page-locked host buffer for one batch of packed samples, kept between batches (pinning 16 MB costs ~20 ms, copying it ~0.3 ms)."""
    key = (n, stride)
    if key not in _STAGING:
        _STAGING.clear()
        buf = torch.zeros((n, stride), dtype=torch.uint8)
        _STAGING[key] = buf.pin_memory() if torch.cuda.is_available() else buf
    return _STAGING[key]
