"""Measurement tool: what the vendor library (hipBLASLt through torch.mm) reaches on the repo's MFMA-bound GEMM shapes on THIS box -- the practical
dense-bf16 ceiling the hand-written >= 512-row kernels are read against.  Prints TFLOP/s per shape, min / median of 20 runs."""
import torch, time, sys
dev = "cuda:0"
shapes = [(512, 15360, 5120), (512, 5120, 5120), (512, 5120, 7680), (1024, 15360, 5120), (2048, 15360, 5120), (2048, 5120, 7680),
          (2048, 71680, 5120), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 0.05
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05 for _ in range(3)]      # rotate: weights are cold in the product too
    for w in ws: torch.mm(a, w.t())
    torch.cuda.synchronize()
    ts = []
    for i in range(21):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.mm(a, ws[i % 3].t()); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    fl = 2.0 * M * N * K
    print(f"torch.mm bf16 {M:5d} x {N:5d} x {K:5d}: min {ts[0]*1e3:8.1f} us  median {ts[10]*1e3:8.1f} us  -> {fl/ts[10]*1e-9:7.1f} TFLOP/s median ({fl/ts[10]*1e-9/2500:.3f} of 2500)", flush=True)
    del a, ws
