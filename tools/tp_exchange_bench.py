"""Latency of the tensor-parallel exchange kernel (csrc/bd_comm.hip) with the ranks as contexts of ONE process on ONE GPU
(one stream per rank, peers linked by plain pointers): protocol cost without xGMI -- two flag round trips, the staging and
result pushes through the local memory system.  A lower bound for the multi-GPU exchange, not a substitute for measuring it.
GPU_MAX_HW_QUEUES must give every rank stream its own hardware queue.   python tools/tp_exchange_bench.py"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd.tp import TPComm          # noqa: E402

DEV = "cuda"
import itertools
STREAMS = [torch.cuda.Stream() for _ in range(4)]          # once: torch hands out pooled streams, a fresh batch per case wraps onto the same ones
for tp, fences, pre in itertools.product((2, 4), (1, 0), (0, 1)):
    for rows, N in ((128, 5120), (32, 5120), (256, 5120)):
        comms = TPComm.in_process(tp, rows * N, DEV)
        for c in comms:
            c.set_timeout(5.0)
            c.l.bd_comm_set_fences(c.h, fences)
        streams = STREAMS[:tp]
        parts = [torch.randn(rows, N, device=DEV) for _ in range(tp)]
        bias = torch.zeros(N, dtype=torch.bfloat16, device=DEV)
        torch.cuda.synchronize()
        reps = 200

        def burst(n):
            import ctypes as C
            for r in range(tp):
                with torch.cuda.stream(streams[r]):
                    out, is32 = C.c_void_p(), C.c_int()
                    for _ in range(n):
                        if pre:                       # phase 2 alone: what is left of the exchange when the GEMM's epilogue has pushed the slices
                            comms[r].l.bd_comm_mark_prepushed(comms[r].h)
                        comms[r].l.bd_comm_allreduce(comms[r].h, parts[r].data_ptr(), bias.data_ptr(), rows, N, C.byref(out), C.byref(is32),
                                                     streams[r].cuda_stream)
        burst(10)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(tp)]
        for r in range(tp):
            ev[r][0].record(streams[r])
        burst(reps)
        for r in range(tp):
            ev[r][1].record(streams[r])
        torch.cuda.synchronize()
        for c in comms:
            c.check()
        us = max(e0.elapsed_time(e1) for e0, e1 in ev) * 1e3 / reps
        print(f"tp={tp} fences={fences} push {'in the GEMM epilogue' if pre else 'in the exchange kernel'} rows={rows:4d} N={N}: {us:6.1f} us per exchange (back to back, {reps} launches per rank; payload "
              f"{rows * N * 6 * (tp - 1) / tp / 1e6:.2f} MB pushed per rank)", flush=True)
