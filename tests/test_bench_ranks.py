"""bench.py's multi-rank logic on CPU (world_size 2, gloo): the SAME run_workload the driver's `torchrun ... bench.py --gpus N --mode train`
executes -- barrier + synchronize on both sides of exactly K timed steps, max-over-ranks timing, per-rank synthetic batches, the initial
weight broadcast, the bucketed gradient exchange from the hooks, an identical optimizer step on every rank -- with a CPU stand-in for
the device runtime (the C oracle's forward / backward restatements behind autograd instead of the HIP kernels; test infrastructure
only).  What RCCL does on hardware gloo does here; the first 8-GPU run then exercises xGMI, not the bench's bookkeeping."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostEvent:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class StubRuntime:
    """CPU stand-in for bench.CudaRuntime."""

    def __init__(self):
        from test_parallel import _cpu_train_backend
        self.backend = _cpu_train_backend()
        self.syncs = 0

    def synchronize(self):
        self.syncs += 1

    def event(self):
        return _HostEvent()

    def optimizer(self, plist):
        return torch.optim.Adam(plist, lr=1e-5)


def _worker(rank, world, port, out, gpus_claimed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        res = {"rank": rank}
        try:
            res["ranks_seen"] = bench.check_ranks(world, gpus_claimed, "cpu")
        except SystemExit as e:
            res["refused"] = str(e)
            torch.save(res, os.path.join(out, f"b_rank{rank}.pt"))
            return
        rt = StubRuntime()
        # a rank that is slower than the other: the reported time must be the MAX over ranks on every rank
        if rank == 1:
            orig = rt.optimizer

            def slow_opt(plist):
                o = orig(plist)
                step0 = o.step

                def step(*a, **k):
                    time.sleep(0.25)
                    return step0(*a, **k)
                o.step = step
                return o
            rt.optimizer = slow_opt
        m = bench.run_workload("C", "train", 1, 128, 128, steps=2, warmup=1, device=torch.device("cpu"), world=world, rank=rank,
                               bucket_mb=24, settle_s=0.0, rt=rt)
        res.update(elapsed=m["elapsed"], elapsed_cold=m["elapsed_cold"], settle_steps=m["settle_steps"], loss=float(m["out"].detach()),
                   img_sum=float(m["img0"].sum()), syncs=rt.syncs, marks=len(m["marks"]),
                   step_ms=[m["marks"][i].elapsed_time(m["marks"][i + 1]) for i in range(2)],
                   params={k: v.detach().clone() for k, v in m["params"].items()})
        torch.save(res, os.path.join(out, f"b_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _leg_worker(rank, world, port, out, gpus_claimed):
    """bench.train_leg -- the leg a multi-rank bench line carries under extra.train_448x320 -- on every rank, at a CPU-sized shape."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        leg = bench.train_leg(torch.device("cpu"), world, rank, bucket_mb=24, B=1, H=128, W=128, steps=2, warmup=1, settle_s=0.0, rt=StubRuntime())
        torch.save({"rank": rank, "leg": leg}, os.path.join(out, f"b_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run(tmp_path, world, gpus_claimed, worker=None):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker or _worker, args=(r, world, port, str(tmp_path), gpus_claimed)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return [torch.load(os.path.join(str(tmp_path), f"b_rank{r}.pt")) for r in range(world)]


def test_bench_train_rank_logic_world2(tmp_path):
    import bench
    r0, r1 = _run(tmp_path, 2, 2)
    assert r0["ranks_seen"] == 2 and r1["ranks_seen"] == 2
    # max over ranks: both ranks report the same wall time, and it contains the slow rank's two 0.25 s sleeps
    assert r0["elapsed"] == r1["elapsed"] and r0["elapsed"] >= 0.5 and r0["elapsed_cold"] == r1["elapsed_cold"]
    assert r0["marks"] == 3 and r0["settle_steps"] == 0 and r0["syncs"] >= 4
    assert sum(r1["step_ms"]) >= 500.0 and sum(r1["step_ms"]) <= r1["elapsed"] * 1e3 + 1.0
    # per-rank synthetic batches (weak scaling): different inputs, different losses
    assert bench.rank_seed(0) != bench.rank_seed(1) and r0["img_sum"] != r1["img_sum"] and r0["loss"] != r1["loss"]
    # one exchange per step + identical update: the ranks end with bit-identical weights, and they moved away from the initial ones
    from flownet2_amd import nets
    P0 = nets.init_params("C", seed=0)
    moved = 0
    for k in P0:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
        moved += int(not torch.equal(r0["params"][k], P0[k]))
    assert moved >= len(P0) - 2


def test_multi_rank_bench_line_carries_the_train_leg_world2(tmp_path):
    """world > 1: extra.train_448x320 comes from train_leg run on ALL ranks -- the gradient all-reduce is inside the timed steps, the same
    step is timed again with world-local gradients, and the line says how many buckets left from inside backward."""
    r0, r1 = _run(tmp_path, 2, 2, worker=_leg_worker)
    for r in (r0, r1):
        leg = r["leg"]
        assert leg["n_gpus"] == 2 and leg["global_batch"] == 2 and leg["ranks_seen_by_rccl"] == 2
        assert leg["grad_buckets"] >= 2 and 1 <= leg["buckets_launched_inside_backward"] <= leg["grad_buckets"]
        assert leg["ms_per_step"] > 0 and leg["ms_per_step_local_gradients"] > 0
        assert abs(leg["allreduce_ms_exposed"] - (leg["ms_per_step"] - leg["ms_per_step_local_gradients"])) < 1e-3
        assert "dp2" in leg["parallelism"] and leg["library_conv_fallbacks"] == 0
    # max over ranks: both ranks report the same times
    assert r0["leg"]["ms_per_step"] == r1["leg"]["ms_per_step"] and r0["leg"]["ms_per_step_local_gradients"] == r1["leg"]["ms_per_step_local_gradients"]


def test_bench_refuses_a_rank_count_that_differs_from_gpus(tmp_path):
    r0, r1 = _run(tmp_path, 2, 4)
    assert "launcher started WORLD_SIZE=2" in r0["refused"] and "--gpus 4" in r1["refused"]


def test_bench_gpus_flag_fails_loudly_without_the_gpus():
    """`python bench.py --gpus 8` on a box with fewer GPUs must not report an 8-GPU number (spawn_ranks)."""
    import bench
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 GPUs")
    with pytest.raises(SystemExit, match="refusing to report"):
        bench.spawn_ranks(8)
