#!/usr/bin/env python
"""py3 re-authoring of scripts/run-flownet-many.py on the MI355X path, sharded over the GPUs of a node.

    python scripts/run_flownet_many.py [--net C|S|2] list.txt                                  # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_flownet_many.py list.txt

list.txt lines: `img0 img1 out.flo` (run-flownet-many.py:27-36).  The reference re-creates the caffe.Net for every
entry (:77-81 inside the loop at :38); here each rank builds the net once, takes every world-th entry
(flownet2_amd.parallel.shard -- no collective on the data path) and batches pairs of equal size.

* Bit-exact outputs whatever the batching / sharding (default; `--no-batch-invariant` trades it for speed): the net runs
  in flownet2_amd.functional's batch-invariant mode, in which no summation order depends on the batch size, so a pair's
  .flo has the same bytes from `run_flownet.py` (batch 1), from this script on 1 GPU and from any N-GPU sharding of the list.
* Host side: every image is decoded ONCE (the size of the next group's first image decides the grouping), by a prefetch
  thread that stays `--prefetch` groups ahead of the GPU and hands over pinned staging buffers (asynchronous H2D copies);
  the .flo files are written by a second thread, so decode, compute and encode of consecutive groups overlap.
* The process group (only a final barrier -- the data path has no exchange) uses gloo: ranks may even share one GPU.
"""
import argparse
import os
import queue
import sys
import threading

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from flownet2_amd import flo, parallel  # noqa: E402
from flownet2_amd import functional as Fn  # noqa: E402
import run_flownet as RF  # noqa: E402


def groups_of(entries, batch, read):
    """Yield (entries, img0 [n,3,H,W], img1) groups of at most `batch` consecutive pairs of equal size; every file is read once."""
    pending = None                         # (entry, img0, img1) that closed the previous group
    it = iter(entries)
    while True:
        group = []
        if pending is not None:
            group.append(pending)
            pending = None
        for e in it:
            item = (e, read(e[0]), read(e[1]))
            if group and item[1].shape != group[0][1].shape:
                pending = item
                break
            group.append(item)
            if len(group) == batch:
                break
        if not group:
            return
        yield [g[0] for g in group], np.concatenate([g[1] for g in group]), np.concatenate([g[2] for g in group])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listfile")
    ap.add_argument("--net", choices=["C", "S", "2"], default="C")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--prefetch", type=int, default=2, help="groups the decode thread may run ahead of the GPU")
    ap.add_argument("--no-batch-invariant", action="store_true", help="let kernel selection depend on the batch size (faster library "
                    "calls; a pair's bits then depend on the batch it was computed in)")
    ap.add_argument("--gpu", type=int, default=None, help="device index (default: LOCAL_RANK)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")              # control plane only; the data path has no collective
    dev = torch.device("cuda", a.gpu if a.gpu is not None else int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    Fn.set_batch_invariant(not a.no_batch_invariant)
    entries = [l.split() for l in open(a.listfile) if l.strip()]
    mine = parallel.shard(entries)
    P, mean = RF.load_params(a.net, a.weights, dev)

    staged = queue.Queue(maxsize=max(1, a.prefetch))
    done = queue.Queue()
    errors = []

    def producer():
        try:
            for ents, i0, i1 in groups_of(mine, a.batch, RF.read_image):
                staged.put((ents, torch.from_numpy(i0).pin_memory(), torch.from_numpy(i1).pin_memory()))
        except BaseException as e:      # noqa: BLE001 -- surfaced on the main thread
            errors.append(e)
        finally:
            staged.put(None)

    def writer():
        try:
            while True:
                item = done.get()
                if item is None:
                    return
                ents, flow, ev = item
                ev.synchronize()
                arr = flow.numpy()
                for k, e in enumerate(ents):
                    flo.write_flo(e[2], arr[k])
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    tp, tw = threading.Thread(target=producer, daemon=True), threading.Thread(target=writer, daemon=True)
    tp.start(); tw.start()
    while True:
        item = staged.get()
        if item is None:
            break
        ents, h0, h1 = item
        i0, i1 = h0.to(dev, non_blocking=True), h1.to(dev, non_blocking=True)
        flow = RF.infer(a.net, P, i0, i1, mean)
        host = torch.empty(flow.shape, dtype=flow.dtype, pin_memory=True)
        host.copy_(flow, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        done.put((ents, host, ev))
    done.put(None)
    tw.join(); tp.join()
    if errors:
        raise errors[0]
    my_rank = parallel.rank()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if my_rank == 0:
        print(f"{len(entries)} pairs, {world} rank(s), batch-invariant {'off' if a.no_batch_invariant else 'on'}")


if __name__ == "__main__":
    main()
