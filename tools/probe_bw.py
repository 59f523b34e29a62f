"""Practical HBM read roofline on this box: pure 16 B/lane non-temporal read stream (bd_probe_read)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd._lib import check, lib   # noqa: E402

x = torch.empty(3 * 1024 ** 3, dtype=torch.uint8, device="cuda").random_(0, 255)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for nbytes in (734 << 20, 3 << 30):
    for blocks in (32, 64, 128, 192, 256, 512, 1024):
        for _ in range(2):
            check(lib().bd_probe_read(x.data_ptr(), nbytes, blocks, sink.data_ptr(), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps):
            check(lib().bd_probe_read(x.data_ptr(), nbytes, blocks, sink.data_ptr(), st))
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"read {nbytes >> 20:5d} MiB blocks={blocks:5d}: {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
y = torch.empty_like(x[: 1 << 30])
for _ in range(2):
    y.copy_(x[: 1 << 30])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y.copy_(x[: 1 << 30])
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"torch copy 1 GiB: {us:.1f} us  read+write {2 * (1 << 30) / us / 1e3:.0f} GB/s")
