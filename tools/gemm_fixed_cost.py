"""Fixed (K-independent) cost of one weight-streaming GEMM launch: time vs K at a fixed grid, linear fit.
python tools/gemm_fixed_cost.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV = "cuda"


def time_gemm(N, K, S, nw, ring=2, M=128, reps=40, rotate=1):
    st = torch.cuda.current_stream().cuda_stream
    wps = []
    for _ in range(rotate):
        w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
        wps.append(E.pack_linear([w], DEV))
        del w
    x = torch.randn(M, K, device=DEV)
    xf = torch.zeros(M * K, dtype=torch.bfloat16, device=DEV)
    check(lib().bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, M // 32, st))
    out = torch.empty(S * M * N, dtype=torch.float32, device=DEV)
    code = nw + 16 * ring

    def launch(i):
        check(lib().bd_gemm_partial(xf.data_ptr(), M // 32, wps[i % rotate].data_ptr(), N, K, S, code, out.data_ptr(), st))
    for i in range(3):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
  for layout in (0, 1):
    check(lib().bd_set_weight_layout(layout))
    print(f"=== weight layout {layout} ({'stage-major' if layout else 'panel-major'})", flush=True)
    for label, N, S, nw in [("qkv-like 240 blocks", 15360, 2, 4), ("wo-like 240 blocks", 5120, 6, 4),
                              ("S=1 240 blocks", 30720, 1, 4), ("ada-like 224 blocks nw10", 71680, 1, 10)]:
          ks, ts = [], []
          for stages in (2, 8, 16, 32, 64, 128):
              K = stages * 64 * S
              nbytes = N * K * 2
              rot = max(1, min(8, int(600e6 // nbytes)))          # rotate weight buffers past the 256 MB Infinity Cache
              if nbytes * rot > 6e9:
                  continue
              us = time_gemm(N, K, S, nw, rotate=rot)
              ks.append(stages)
              ts.append(us)
              print(f"{label:26s} N={N:6d} K={K:6d} S={S} stages/slice={stages:4d} rot={rot}  {us:8.2f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
          a, b = np.polyfit(np.array(ks[1:], float), np.array(ts[1:], float), 1)
          print(f"  fit(>=8 stages): {b:.2f} us fixed + {a:.3f} us/stage  (stage = {N * 64 * 2 * S / 1e3:.0f} KB of W -> {N * 64 * 2 * S / a / 1e3:.0f} GB/s steady)", flush=True)


if __name__ == "__main__":
    main()
