"""Tuned library-GEMM selections for the GEMM route (nets._use_gemm_conv, deconvolutions).

rocBLAS / hipBLASLt pick a kernel per GEMM shape from a heuristic; PyTorch's TunableOp can time the candidates once and
record the winner.  `gemm_gfx950.csv` holds the winners for the GEMM shapes of FlowNetC (batch 8 @448x320) and FlowNet2
(batch 4 @768x384) on an MI355X -- 10-35 % faster than the heuristic choice for the deconvolution GEMMs -- and
`enable()` switches TunableOp to look-up-only mode on that file (no tuning at run time; shapes that are not in the file keep
the library default; a file recorded with other library versions is rejected by TunableOp's own validators and ignored).
Regenerate with scripts/tune_gemms.py on the GPU box.
"""
from __future__ import annotations

import os

import torch

CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_gfx950.csv")
_state = None          # None = not tried yet, True = table active, False = unavailable (cached: enable() runs once per layer call)
_private = None


def _cleanup():
    if _private and os.path.exists(_private):
        try:
            os.unlink(_private)
        except OSError:
            pass


def active() -> bool:
    return bool(_state)


def enable() -> bool:
    """Idempotent.  Returns True when the tuned selections are active."""
    global _state, _private
    if _state is not None:
        return _state
    _state = False
    if os.environ.get("FN2_NO_TUNED_GEMM") == "1" or not os.path.exists(CSV) or not torch.cuda.is_available():
        return False
    try:
        import torch.cuda.tunable as tn
        if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":      # somebody is recording: do not interfere
            return False
        # TunableOp rewrites "its" file when the process exits: give every process a private copy, so that the table in the
        # tree is never touched and the ranks of a multi-GPU job do not write one file
        import atexit
        import shutil
        import tempfile
        fd, private = tempfile.mkstemp(prefix="fn2_gemm_gfx950_", suffix=".csv")
        os.close(fd)
        shutil.copyfile(CSV, private)
        _private = private
        tn.enable(True)
        tn.tuning_enable(False)
        tn.set_filename(private, insert_device_ordinal=False)
        tn.read_file(private)
        # look-up only: nothing to record.  TunableOp rewrites "its" file when the process exits; point it at the null device (or
        # switch the write off where the API exists) and drop the private copy right away
        if hasattr(tn, "write_file_on_exit"):
            tn.write_file_on_exit(False)
        else:
            tn.set_filename(os.devnull, insert_device_ordinal=False)
        _cleanup()
        atexit.register(_cleanup)
        _state = True
    except Exception:                                               # TunableOp unavailable in this torch build: library defaults
        return False
    return True
