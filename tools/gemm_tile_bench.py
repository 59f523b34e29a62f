"""The LDS-tiled MFMA-bound GEMM (csrc/bd_gemm_tile.hip) against the 256-row weight-streaming kernel it replaces at >= 512 rows:
values (bit-identical for one K slice: same MFMA, same K order) and time, at the shapes that launch it -- the adaLN projection
of 4 .. 52 evaluations (N = 71 680, K = 5120), the eval batch (512 rows: qkv / wo / w1 / w2 of the 14B head) and the ImageNet
batch (12 288 rows at width 768).  Random bf16 data (timing on zeros overstates throughput: clocks).
python tools/gemm_tile_bench.py            tile kernel vs 256-row kernel
python tools/gemm_tile_bench.py keep       256-row kernel only: non-temporal vs default-policy weight loads (bd_gemm.hip wide.keep)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


def timed(launch, reps):
    for i in range(2):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    l = lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=DEV).manual_seed(1)
    shapes = [  # name, N, K, row blocks, form (b = bf16+bias, p = fp32 slabs, s = swiglu), S
        ("head.ada x4", 71680, 5120, 16, "b", 1), ("head.ada x8", 71680, 5120, 32, "b", 1), ("head.ada x16", 71680, 5120, 64, "b", 1),
        ("head.ada x52", 71680, 5120, 208, "b", 1),
        ("head.qkv 512", 15360, 5120, 16, "p", 2), ("head.w1 512", 15360, 5120, 16, "s", 1), ("head.wo 512", 5120, 5120, 16, "p", 5),
        ("head.w2 512", 5120, 7680, 16, "p", 5), ("llm.gu 512", 34816, 5120, 16, "s", 1),
        ("imagenet qkv", 2304, 768, 384, "b", 1), ("imagenet w1", 4096, 768, 384, "s", 1), ("imagenet w2", 768, 2048, 384, "b", 1),
    ]
    keep_mode = len(sys.argv) > 1 and sys.argv[1] == "keep"
    if keep_mode:
        shapes = [sh for sh in shapes if sh[3] <= 64]
    for name, N, K, RB, form, S in shapes:
        M = RB * 32
        nrot = max(1, min(4, int(600e6 // (N * K * 2))))
        ws = [(torch.randn(N * K, device=DEV, generator=g) * 0.02).to(BF16) for _ in range(nrot)]     # packed order is irrelevant for timing / identity
        a = (torch.randn(M * K, device=DEV, generator=g)).to(BF16)
        bias = (torch.randn(N, device=DEV, generator=g) * 0.1).to(BF16)
        outs = {}
        res = {}
        for tile in ((0, 1) if keep_mode else (0, 3, 2)):
            if keep_mode:                                   # slot 0: nt loads, slot 1: default-policy loads, both on the 256-row kernel
                check(l.bd_set_gemm_option(b"tile", 0))
                check(l.bd_set_gemm_option(b"wide.keep", tile))
            else:
                check(l.bd_set_gemm_option(b"tile", tile))
            if form == "p":
                out = torch.zeros(S * M * N, dtype=torch.float32, device=DEV)
                launch = lambda i: check(l.bd_gemm_partial(a.data_ptr(), RB, ws[i % nrot].data_ptr(), N, K, S, 8, out.data_ptr(), st), "gemm_partial")
            elif form == "b":
                out = torch.zeros(M * N, dtype=BF16, device=DEV)
                launch = lambda i: check(l.bd_gemm_bf16(a.data_ptr(), RB, ws[i % nrot].data_ptr(), bias.data_ptr(), N, K, 1, 8, None, None,
                                                        out.data_ptr(), st), "gemm_bf16")
            else:
                out = torch.zeros(M * N // 2, dtype=BF16, device=DEV)
                launch = lambda i: check(l.bd_gemm_swiglu(a.data_ptr(), RB, ws[i % nrot].data_ptr(), bias.data_ptr(), N, K, 8, out.data_ptr(), st),
                                         "gemm_swiglu")
            launch(0)
            torch.cuda.synchronize()
            outs[tile] = out.clone()
            reps = 3 if RB >= 128 else 10
            res[tile] = timed(launch, reps)
        vw = torch.int16 if form != "p" else torch.int32
        same = torch.equal(outs[0].view(vw), outs[1 if keep_mode else 3].view(vw))
        fl = 2.0 * M * N * K
        la, lb = ("nt loads      ", "default loads") if keep_mode else ("256-row kernel", "tile kernel (LDS-DMA)")
        t1 = 1 if keep_mode else 3
        line = (f"{name:16s} N={N:6d} K={K:5d} rows={M:6d} S={S}  {la} {res[0]:9.1f} us {fl / res[0] / 1e6:7.0f} TFLOP/s | "
                f"{lb} {res[t1]:9.1f} us {fl / res[t1] / 1e6:7.0f} TFLOP/s ({fl / res[t1] / 1e6 / 2500:.3f} of peak) | bit-identical {same}")
        if 2 in res:                                            # the tile kernel with register-staged operand fetch (option "tile" = 2)
            same2 = torch.equal(outs[0].view(vw), outs[2].view(vw))
            line += f" | register-staged {res[2]:9.1f} us {fl / res[2] / 1e6:7.0f} TFLOP/s | bit-identical {same2}"
        print(line, flush=True)
        del ws, a, outs
        torch.cuda.empty_cache()
    check(l.bd_set_gemm_option(b"tile", 1))
    check(l.bd_set_gemm_option(b"wide.keep", -1))


if __name__ == "__main__":
    main()
