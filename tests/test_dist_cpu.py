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
