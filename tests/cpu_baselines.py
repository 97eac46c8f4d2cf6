#!/usr/bin/env python
"""CPU baselines of the sample decode on this box's host cores (TEST INFRASTRUCTURE: uses the oracle and oracle/_ref).
Prints the lines scripts/summarize_layer_microbench.py appends to profiles/<tag>_layer_microbench.md:
  * the oracle's restatement of DecodeData + the slice copy (one thread, like the reference's prefetch thread),
  * the reference's own CustomDataLayer (oracle/_ref: custom_data_layer.cpp compiled in place over an in-memory LMDB stand-in).
Usage: python tests/cpu_baselines.py >> gpurun_out/layer_microbench.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle            # noqa: E402
from oracle import ref   # noqa: E402

SP, ENC = (3, 6, 8), (1, 1, 2, 3)
Hs, Ws, Ns = 384, 512, 8
nb = oracle.custom_data_sample_bytes(9, Hs, Ws, SP, ENC)
host = np.random.default_rng(0).integers(0, 256, (Ns, (nb + 15) // 16 * 16), dtype=np.uint8)
t0 = time.time()
oracle.custom_data_decode(host, 9, Hs, Ws, SP, ENC)
t_or = time.time() - t0
print("CPU decode of a batch of %d samples %dx%d: oracle restatement %.1f ms (%.0f samples/s, 1 thread)" % (Ns, Ws, Hs, t_or * 1e3, Ns / t_or))
if ref.available():
    recs = [("%08d" % i, oracle.datum_serialize(9, Hs, Ws, host[i, :nb].tobytes(), i)) for i in range(Ns)]
    t0 = time.time()
    ref.custom_data(recs, Ns, SP, ENC, n_forward=4)
    t_ref = (time.time() - t0) / 5            # SetUp prefetches one batch too
    print("reference CustomDataLayer (oracle/_ref, host prefetch thread): %.1f ms per batch of %d (%.0f samples/s)" % (t_ref * 1e3, Ns, Ns / t_ref))
