"""Builds flownet2_amd/libflownet2_hip.so (gfx950 only) with hipcc, in-tree.

`python -m flownet2_amd.build` or `flownet2_amd.build.build()`; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libflownet2_hip.so")
ARCH = "gfx950"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def is_stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "flownet2_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
             "-Wall", "-Wno-unused-function"]
    if os.environ.get("FN2_ABLATION"):
        flags.append("-DFN2_ABLATION=1")
    # per-file flags: conv_wino.hip writes its packed-fp32 pairs out by hand; clang's SLP vectoriser adds pairs of its own there and
    # pays for each with v_mov_b32 to assemble the operands (26 moves per 32 MFMAs in the round-3 build of the k loop)
    per_file = {"conv_wino.hip": ["-fno-slp-vectorize"]}
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "flownet2_hip.h")]
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs):
            continue
        cmd = [hipcc()] + flags + per_file.get(os.path.basename(src), []) + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs + ["-lz"]      # zlib: csrc/hdf5_reader.cpp (gzip chunks)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
