""".flo I/O against the reference-held fixtures (data/FlyingChairs_examples/*-gt.flo, util/output.cpp:16-65,
scripts/run-flownet.py:100-126) and BASELINE.json configs[0]: FlowNetS deploy on one FlyingChairs pair -> .flo.

The fixture digests and the first pair are committed under tests/golden/chairs (tests/golden/make_flo_fixtures.py made them
from the reference tree); where /root/reference exists every reference file itself is read, re-written and byte-compared."""
import glob
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from flownet2_amd import flo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHAIRS = os.path.join(ROOT, "tests", "golden", "chairs")
REF_DIR = "/root/reference/data/FlyingChairs_examples"
FIX = json.load(open(os.path.join(CHAIRS, "flo_fixtures.json")))


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_fixture_digest_file_is_consistent():
    assert len(FIX["files"]) == 9
    h = hashlib.sha256()
    for name in sorted(FIX["files"]):
        e = FIX["files"][name]
        assert e["magic"] == "PIEH" and e["width"] == 512 and e["height"] == 384 and e["bytes"] == 12 + 512 * 384 * 8
        h.update(bytes.fromhex(e["sha256"]))
    assert h.hexdigest() == FIX["sha256_of_sha256s"]


def test_writer_reproduces_reference_fixture_bytes(tmp_path):
    """The committed ground truth of pair 0000000 (float32 [384,512,2]) written by flo.write_flo has the sha256 of the
    reference's 0000000-gt.flo: header, element order and endianness are the reference's.  Both accepted layouts."""
    gt = np.load(os.path.join(CHAIRS, "0000000-gt.npz"))["flow"]
    e = FIX["files"]["0000000-gt.flo"]
    for arr in (gt, gt.transpose(2, 0, 1), gt.transpose(2, 0, 1)[None]):          # (H,W,2), [2,H,W] blob, [1,2,H,W] blob
        p = str(tmp_path / "w.flo")
        flo.write_flo(p, arr)
        assert os.path.getsize(p) == e["bytes"] and sha(p) == e["sha256"]
    back = flo.read_flo(p)
    assert back.shape == (384, 512, 2) and np.array_equal(back.view(np.uint32), gt.view(np.uint32))
    assert [float(back[0, 0, 0]), float(back[0, 0, 1])] == e["first_uv"]
    assert [float(back[-1, -1, 0]), float(back[-1, -1, 1])] == e["last_uv"]


@pytest.mark.skipif(not os.path.isdir(REF_DIR), reason="reference tree not present (GPU box)")
def test_every_reference_flo_roundtrips_byte_exact(tmp_path):
    files = sorted(glob.glob(os.path.join(REF_DIR, "*-gt.flo")))
    assert len(files) == 9
    for f in files:
        e = FIX["files"][os.path.basename(f)]
        assert sha(f) == e["sha256"]
        raw = open(f, "rb").read()
        assert raw[:4] == b"PIEH" and np.frombuffer(raw, "<i4", 2, 4).tolist() == [512, 384]
        a = flo.read_flo(f)
        assert a.shape == (384, 512, 2) and a.dtype == np.float32
        assert abs(float(np.abs(a.astype(np.float64)).mean()) - e["mean_abs"]) < 1e-12
        out = str(tmp_path / os.path.basename(f))
        flo.write_flo(out, a)
        assert open(out, "rb").read() == raw
        flo.write_flo(out, a.transpose(2, 0, 1))                                    # blob layout (run-flownet.py:98 transposes it back)
        assert open(out, "rb").read() == raw


@pytest.mark.skipif(not os.path.isdir(REF_DIR), reason="reference tree not present (GPU box)")
def test_committed_pair_equals_reference_ppm():
    from PIL import Image
    for k in ("img0", "img1"):
        a = np.asarray(Image.open(os.path.join(REF_DIR, "0000000-%s.ppm" % k)))
        b = np.asarray(Image.open(os.path.join(CHAIRS, "0000000-%s.png" % k)))
        assert a.shape == (384, 512, 3) and np.array_equal(a, b)


@pytest.mark.gpu
def test_config1_flownet_s_on_a_flyingchairs_pair_writes_flo(tmp_path):
    """BASELINE.json configs[0]: FlowNetS deploy on one FlyingChairs pair through the runner (scripts/run_flownet.py, the
    re-authored scripts/run-flownet.py) -> .flo.  Weights are seeded random (no .caffemodel can be fetched), so the flow is
    checked for format, determinism and agreement with the CPU-oracle graph -- not for accuracy against the ground truth."""
    import torch
    from flownet2_amd import nets
    from oracle import backend as cpu_backend
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import run_flownet
    i0, i1 = (os.path.join(CHAIRS, "0000000-%s.png" % k) for k in ("img0", "img1"))
    outs = []
    for r in range(2):
        out = str(tmp_path / ("o%d.flo" % r))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "run_flownet.py"), "--net", "S", i0, i1, out])
        outs.append(out)
    raw = open(outs[0], "rb").read()
    assert raw[:4] == b"PIEH" and np.frombuffer(raw, "<i4", 2, 4).tolist() == [512, 384] and len(raw) == 12 + 512 * 384 * 8
    flow = flo.read_flo(outs[0])
    assert np.isfinite(flow).all()
    # two processes: our kernels are deterministic, but the library convolutions (MIOpen's find step times its candidates per
    # process) may pick different algorithms -- rounding-level differences only; no NaN retry loop needed (run-flownet.py:72-96)
    again = flo.read_flo(outs[1])
    assert float(np.sqrt(((flow - again) ** 2).sum(-1)).mean()) <= 1e-4
    P = nets.init_params("S", 0)
    a, b = torch.from_numpy(run_flownet.read_image(i0)), torch.from_numpy(run_flownet.read_image(i1))
    with torch.no_grad():
        want = nets.deploy_forward("S", P, a, b, cpu_backend)[0].numpy().transpose(1, 2, 0)
    epe = float(np.sqrt(((flow - want) ** 2).sum(-1)).mean())
    assert epe <= 1e-4, epe
