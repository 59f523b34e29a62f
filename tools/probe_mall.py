"""Measurement tool: how much faster does a 128-row weight-streaming GEMM run when its weights sit in the 256 MB Infinity Cache (MALL) instead of HBM?
(the price of any scheme that prefetches the NEXT launch's weights during a launch's tail / a row kernel -- DESIGN.md 3.3)
  copies = 1: the same packed matrix every launch (resident after the first pass, when it fits);  copies = 4: rotation through > 256 MB (always cold).
python tools/probe_mall.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd._lib import check, lib                    # noqa: E402

DEV = "cuda"
SHAPES = [("head.qkv / w1", 15360, 5120, 2, 4 + 16 * 3), ("head.wo", 5120, 5120, 3, 4 + 16 * 2 + 256), ("head.w2", 5120, 7680, 3, 4 + 16 * 2 + 256)]


def main():
    l = lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(128, 7680, device=DEV, generator=g)
    for name, N, K, S, code in SHAPES:
        xf = torch.zeros(128 * K, dtype=torch.bfloat16, device=DEV)
        check(l.bd_rows_to_frag(xf.data_ptr(), x[:, :K].contiguous().data_ptr(), 1, 128, K, 4, st))
        for copies in (1, 4):
            ws = [E.pack_linear([(torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)], DEV) for _ in range(copies)]
            scratch = torch.empty(S * 128 * N, device=DEV)
            cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
            out = torch.empty(128 * N, dtype=torch.bfloat16, device=DEV)
            part = torch.empty(S * 128 * N, device=DEV)

            def launch(i):
                w = ws[i % copies]
                if S == 2:
                    check(l.bd_gemm_bf16(xf.data_ptr(), 4, w.data_ptr(), None, N, K, S, code, scratch.data_ptr(), cnt.data_ptr(), out.data_ptr(), st))
                else:
                    check(l.bd_gemm_partial(xf.data_ptr(), 4, w.data_ptr(), N, K, S, code, part.data_ptr(), st))
            for i in range(8):
                launch(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 48
            e0.record()
            for i in range(n):
                launch(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f"{name:16s} N {N:6d} K {K:5d}  {N * K * 2 / 1e6:6.1f} MB  weights {'resident in the Infinity Cache (1 copy)' if copies == 1 else 'cold (4 copies in rotation)':42s} "
                  f"{us:7.2f} us per launch = {N * K * 2 / us * 1e-6:5.2f} TB/s", flush=True)
            del ws


if __name__ == "__main__":
    main()
