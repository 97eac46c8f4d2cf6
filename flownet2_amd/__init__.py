"""flownet2_amd: MI355X (gfx950) implementation of the FlowNet2 hot path of lmb-freiburg/flownet2.

csrc/      hand-written HIP kernels + the C ABI (include/flownet2_hip.h) -> libflownet2_hip.so
_lib/ops   ctypes binding (no fallback: a missing library raises)
layers     host-side mirror of the reference's Caffe Layer<Dtype> / LayerRegistry interface
functional torch.autograd glue,  nets: FlowNetC/S graphs,  flo: .flo I/O
"""
from . import _lib  # noqa: F401
from ._lib import Fn2Error, version  # noqa: F401

__all__ = ["Fn2Error", "version"]
