"""Middlebury .flo I/O, byte-compatible with the reference runner and C++ writer.

Format (scripts/run-flownet.py:100-126; src/caffe/util/output.cpp:16-65): ASCII "PIEH",
int32 width, int32 height, then height*width (u, v) float32 pairs, row-major, little endian --
i.e. the [2,H,W] blob transposed to [H,W,2].
"""
from __future__ import annotations

import numpy as np

MAGIC = b"PIEH"


def write_flo(path: str, flow) -> None:
    """flow: (H, W, 2) array (as run-flownet.py's writeFlow) or a [2,H,W] / [1,2,H,W] blob."""
    a = np.asarray(flow)
    if a.ndim == 4:
        a = a[0]
    if a.ndim == 3 and a.shape[0] == 2 and a.shape[2] != 2:
        a = a.transpose(1, 2, 0)                      # run-flownet.py:98
    if a.ndim != 3 or a.shape[2] != 2:
        raise ValueError(f"flow must be (H,W,2) or (2,H,W), got {a.shape}")
    with open(path, "wb") as f:
        f.write(MAGIC)
        np.array([a.shape[1], a.shape[0]], dtype="<i4").tofile(f)
        np.ascontiguousarray(a, dtype="<f4").tofile(f)


def read_flo(path: str) -> np.ndarray:
    """Returns (H, W, 2) float32 (run-flownet.py:100-115)."""
    with open(path, "rb") as f:
        if f.read(4) != MAGIC:
            raise ValueError("Flow file header does not contain PIEH")
        w, h = np.fromfile(f, "<i4", 2)
        data = np.fromfile(f, "<f4", int(w) * int(h) * 2)
    if data.size != int(w) * int(h) * 2:
        raise ValueError(f"File corrupted: {path}")      # output.cpp:39-41
    return data.reshape(int(h), int(w), 2).astype(np.float32)
