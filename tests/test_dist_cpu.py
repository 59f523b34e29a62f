"""world_size-2 gloo test of the multi-GPU (replica) harness logic used by bench.py."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from bitdance_amd.dist_util import job_throughput, max_over_ranks, rank_seed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dt = 1.0 + rank                       # rank 1 is the slow replica
    dist.barrier()
    mx = max_over_ranks(dt, dist, "cpu")
    seeds = [rank_seed(1234, rank, i) for i in range(3)]
    q.put((rank, mx, job_throughput(2, 3, mx, world), seeds))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_harness_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, t0, s0), (r1, m1, t1, s1) = res
    assert m0 == m1 == 2.0                      # max over ranks, identical on every rank
    assert t0 == t1 == 2 * 2 * 3 / 2.0          # whole-job images/s
    assert not set(s0) & set(s1)                # disjoint image seeds per replica


# ---------------------------------------------------------------------------------------------------------
# FID-50k data-parallel sampler (imagenet_gen/sample_ddp*.py): the image-index / class-label sharding
def _reference_rank_loop(num_fid, num_classes, n, world, rank):
    """The reference loop's index arithmetic, line for line (sample_ddp_parallel.py:126-183), without the model."""
    import numpy as np
    images_per_class = num_fid // num_classes
    class_label_gen_world = np.arange(0, num_classes).repeat(images_per_class)
    class_label_gen_world = np.hstack([class_label_gen_world, np.zeros(50000)])
    iterations = num_fid // (n * world) + 1
    calls, saved = [], []
    for i in range(iterations):
        idx_start = world * n * i + rank * n
        idx_end = idx_start + n
        if idx_start >= num_fid:
            break
        labels_np = class_label_gen_world[idx_start:idx_end]
        if len(labels_np) == 0:
            break
        calls.append((idx_start, labels_np.astype("int64").tolist()))
        for b_id in range(len(labels_np)):
            img_id = world * n * i + rank * n + b_id
            if img_id >= num_fid:
                break
            saved.append(img_id)
    return calls, saved


def _sampler_worker(rank, world, port, q, num_fid, num_classes, n):
    sys.path.insert(0, ROOT)
    from bitdance_amd.imagenet_sampler import rank_plan, rank_seed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = rank_plan(num_fid, num_classes, n, world, rank)
    mine = torch.zeros(num_fid, dtype=torch.int64)
    for start, labels, keep in plan:
        mine[start:start + keep] += 1
    dist.all_reduce(mine)                                   # every image index saved by exactly one rank
    q.put((rank, [(s, l.tolist(), k) for s, l, k in plan], mine.tolist(), rank_seed(99, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_fid_sampler_sharding_world2():
    """world-2 gloo run of the DP sampler's plan: per rank identical to the reference loop's calls and saved indices, the
    ranks together cover every image exactly once (also with a ragged tail and a batch that does not divide), labels follow
    the fixed class list, per-rank seeds differ."""
    num_fid, num_classes, n = 1000, 10, 96                  # 1000 / (96 * 2) = 5.2 iterations: ragged tail on both ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_sampler_worker, args=(r, 2, port, q, num_fid, num_classes, n)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from bitdance_amd.imagenet_sampler import class_list
    labels_all = class_list(num_fid, num_classes)
    for rank, plan, cover, seed in res:
        ref_calls, ref_saved = _reference_rank_loop(num_fid, num_classes, n, 2, rank)
        assert [(s, l) for s, l, _ in plan] == ref_calls
        assert [s + b for s, _, k in plan for b in range(k)] == ref_saved
        assert cover == [1] * num_fid
        for s, l, k in plan:
            assert l == labels_all[s:s + n].astype("int64").tolist() and len(l) == n
    assert res[0][3] != res[1][3] and res[0][3] == 99 * 2


def test_fid_sampler_uint8_and_args():
    sys.path.insert(0, ROOT)
    from bitdance_amd.imagenet_sampler import folder_name, get_args, to_uint8
    x = torch.tensor([-1.2, -1.0, 0.0, 0.999, 1.5]).view(1, 1, 1, 5).repeat(1, 3, 1, 1)
    assert to_uint8(x)[0, 0, :, 0].tolist() == [0, 0, 128, 255, 255]        # clamp(127.5 x + 128), truncation to uint8
    a = get_args(["--model", "BitDance-B", "--ckpt", "models/BitDance_B_16x.pt", "--cfg-scale", "6.1", "--parallel-num", "16"])
    assert folder_name(a) == "BitDance-B-BitDance_B_16x-size-256-steps-100-cfg-6.1-seed-99"


def test_bench_launch_line_for_gpus_n(monkeypatch):
    """`python bench.py --gpus N` from a plain interpreter re-executes itself as one process per GPU: the launch line is the
    driver's own (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <the same arguments>`), the rendezvous is on 127.0.0.1 and the dmabuf IPC switch travels in the environment."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setenv("BD_BENCH_BACKEND", "gloo")             # (no devices here: the RCCL path refuses before launching)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    args = bench.parse()
    assert args.gpus == 4
    assert bench.spawn_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4"]
    assert cmd[5:7] == ["--master-addr", "127.0.0.1"] and cmd[7] == "--master-port" and 1024 < int(cmd[8]) < 65536
    assert cmd[9] == os.path.join(root, "bench.py") and cmd[10:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("BD_BENCH_BACKEND")
    assert bench.spawn_ranks(args) == 2                        # RCCL backend, fewer devices than ranks: refused with a message, nothing launched
