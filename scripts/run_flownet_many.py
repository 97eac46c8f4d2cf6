#!/usr/bin/env python
"""py3 re-authoring of scripts/run-flownet-many.py on the MI355X path, sharded over the GPUs of a node.

    python scripts/run_flownet_many.py [--net C|S|2] list.txt                                  # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_flownet_many.py list.txt

list.txt lines: `img0 img1 out.flo` (run-flownet-many.py:27-36).  The reference re-creates the caffe.Net for every
entry (:77-81 inside the loop at :38); here each rank builds the net once, takes every world-th entry
(flownet2_amd.parallel.shard -- no collective on the data path) and batches pairs of equal size."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from flownet2_amd import flo, parallel  # noqa: E402
import run_flownet as RF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listfile")
    ap.add_argument("--net", choices=["C", "S", "2"], default="C")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    entries = [l.split() for l in open(a.listfile) if l.strip()]
    mine = parallel.shard(entries)
    P, mean = RF.load_params(a.net, a.weights, dev)
    i = 0
    while i < len(mine):
        first = RF.read_image(mine[i][0])
        group = [mine[i]]
        while len(group) < a.batch and i + len(group) < len(mine) and RF.read_image(mine[i + len(group)][0]).shape == first.shape:
            group.append(mine[i + len(group)])
        i0 = torch.cat([torch.from_numpy(RF.read_image(e[0])) for e in group]).to(dev)
        i1 = torch.cat([torch.from_numpy(RF.read_image(e[1])) for e in group]).to(dev)
        flow = RF.infer(a.net, P, i0, i1, mean).cpu().numpy()
        for k, e in enumerate(group):
            flo.write_flo(e[2], flow[k])
        i += len(group)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parallel.rank() == 0:
        print(f"{len(entries)} pairs, {world} rank(s)")


if __name__ == "__main__":
    main()
