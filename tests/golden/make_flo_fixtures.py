"""Generates tests/golden/chairs/* from the reference-held FlyingChairs examples (data/FlyingChairs_examples, the only
golden data the reference ships for the FlowNet path).  Run HERE (needs /root/reference); the outputs are committed so the
GPU box -- which has no /root/reference -- can run config 1 (one FlyingChairs pair -> .flo) and check .flo I/O.

  python tests/golden/make_flo_fixtures.py

Writes: flo_fixtures.json  (per *-gt.flo: sha256, size, magic, width, height, first/last (u,v), mean |flow|, and a
                            checksum of checksums),
        0000000-img0.png / -img1.png  (the first pair, PPM -> PNG, lossless),
        0000000-gt.npz                (its ground-truth flow, float32 [384,512,2], deflate-compressed)."""
import glob
import hashlib
import json
import os
import struct

import numpy as np
from PIL import Image

SRC = "/root/reference/data/FlyingChairs_examples"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chairs")


def main():
    os.makedirs(DST, exist_ok=True)
    entries, allsum = {}, hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(SRC, "*-gt.flo"))):
        raw = open(p, "rb").read()
        w, h = struct.unpack("<ii", raw[4:12])
        f = np.frombuffer(raw, "<f4", offset=12).reshape(h, w, 2)
        sha = hashlib.sha256(raw).hexdigest()
        allsum.update(bytes.fromhex(sha))
        entries[os.path.basename(p)] = {
            "sha256": sha, "bytes": len(raw), "magic": raw[:4].decode("ascii"), "width": w, "height": h,
            "first_uv": [float(f[0, 0, 0]), float(f[0, 0, 1])], "last_uv": [float(f[-1, -1, 0]), float(f[-1, -1, 1])],
            "mean_abs": float(np.abs(f.astype(np.float64)).mean()),
        }
    json.dump({"source": "data/FlyingChairs_examples/*-gt.flo", "files": entries, "sha256_of_sha256s": allsum.hexdigest()},
              open(os.path.join(DST, "flo_fixtures.json"), "w"), indent=1, sort_keys=True)
    for k in ("img0", "img1"):
        Image.open(os.path.join(SRC, "0000000-%s.ppm" % k)).save(os.path.join(DST, "0000000-%s.png" % k), optimize=True)
    raw = open(os.path.join(SRC, "0000000-gt.flo"), "rb").read()
    np.savez_compressed(os.path.join(DST, "0000000-gt.npz"), flow=np.frombuffer(raw, "<f4", offset=12).reshape(384, 512, 2))
    print({k: os.path.getsize(os.path.join(DST, k)) for k in sorted(os.listdir(DST))})


if __name__ == "__main__":
    main()
