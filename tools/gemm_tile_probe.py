"""Which side binds the LDS-tiled GEMM (csrc/bd_gemm_tile.hip): the same launch with the MFMA work switched off (DMA + barriers only)
and with the DMA switched off after the prologue (LDS reads + MFMA + barriers only).  python tools/gemm_tile_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16
l = lib()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=DEV).manual_seed(1)
for name, N, K, RB in (("ada x16", 71680, 5120, 64), ("ada x4", 71680, 5120, 16), ("imagenet w1-like", 4096, 768, 384)):
    M = RB * 32
    w = (torch.randn(N * K, device=DEV, generator=g) * 0.02).to(BF16)
    a = torch.randn(M * K, device=DEV, generator=g).to(BF16)
    out = torch.zeros(M * N, dtype=BF16, device=DEV)
    for dbg, what in ((0, "full"), (1, "no MFMA (DMA + barriers)"), (2, "no DMA (LDS reads + MFMA + barriers)"), (3, "barriers only")):
        check(l.bd_set_gemm_option(b"tile.debug", dbg))
        run = lambda: check(l.bd_gemm_bf16(a.data_ptr(), RB, w.data_ptr(), None, N, K, 1, 8, None, None, out.data_ptr(), st))
        run(); run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        print(f"{name:18s} rows={M:6d} {what:40s} {us:9.1f} us  ({2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s equiv, {(M + N) * K * 2 * (N // 256) * 0 + 0:.0f})", flush=True)
check(l.bd_set_gemm_option(b"tile.debug", 0))
