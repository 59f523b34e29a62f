"""Round-2 GEMM sweep at M = 128: workgroup shapes (waves, K-parts), grid split-K and epilogue form (fp32 slabs vs the
in-launch reduction to bf16) on the shapes of the AR step, weights rotated past the 256 MB Infinity Cache.
python tools/gemm_sweep2.py [name ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV = "cuda"
SHAPES = {"head.qkv": (15360, 5120), "head.wo": (5120, 5120), "head.w2": (5120, 7680), "llm.qkv": (7168, 5120),
          "llm.down": (5120, 17408), "llm.gu": (34816, 5120)}
# (waves, kparts, S, form): form p = fp32 slabs, r = reduced inside the launch (bf16 out)
CANDS = {
    "head.qkv": [(4, 1, 2, "r"), (4, 1, 2, "p"), (8, 2, 2, "r"), (8, 2, 2, "p"), (4, 2, 1, "r"), (8, 1, 3, "p"), (8, 1, 3, "r"), (10, 2, 1, "r"), (8, 2, 1, "r")],
    "head.wo": [(4, 1, 6, "p"), (8, 2, 6, "p"), (4, 2, 3, "r"), (4, 2, 3, "p"), (2, 1, 3, "r"), (2, 1, 3, "p"), (4, 1, 3, "r"), (8, 2, 3, "r"), (4, 1, 4, "p"), (8, 2, 4, "p")],
    "head.w2": [(4, 1, 6, "p"), (8, 2, 6, "p"), (4, 2, 3, "r"), (2, 1, 3, "r"), (8, 2, 3, "r"), (8, 2, 4, "p")],
    "llm.qkv": [(4, 1, 4, "p"), (8, 2, 4, "p"), (4, 2, 2, "r"), (8, 2, 2, "r"), (4, 1, 2, "r")],
    "llm.down": [(8, 1, 9, "p"), (8, 2, 6, "p"), (4, 1, 6, "p"), (4, 2, 3, "r"), (8, 2, 3, "r"), (8, 1, 12, "p")],
    "llm.gu": [(8, 1, 1, "p"), (8, 2, 1, "p"), (8, 2, 2, "p"), (4, 1, 1, "p")],
}


def main():
    names = [a for a in sys.argv[1:] if a in SHAPES] or list(SHAPES)
    M, RB = 128, 4
    st = torch.cuda.current_stream().cuda_stream
    for name in names:
        N, K = SHAPES[name]
        rot = max(2, min(8, int(700e6 // (N * K * 2))))
        wps = []
        for _ in range(rot):
            w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
            wps.append(E.pack_linear([w], DEV))
            del w
        x = torch.randn(M, K, device=DEV)
        xf = torch.zeros(M * K, dtype=torch.bfloat16, device=DEV)
        check(lib().bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, RB, st))
        scratch = torch.empty(12 * M * N, dtype=torch.float32, device=DEV)
        outb = torch.empty(M * N, dtype=torch.bfloat16, device=DEV)
        cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
        for nw, kw, S, form in CANDS[name]:
            if N % (32 * nw // kw) or K % (64 * kw):
                continue
            for pipe in (0, 1):
              if pipe and nw > 4:
                  continue
              code = nw + 32 + 256 * (kw - 1) + 2048 * pipe
              bench_one(name, N, K, nw, kw, S, form, code, pipe, xf, wps, rot, scratch, outb, cnt, RB, st)
        del wps


def bench_one(name, N, K, nw, kw, S, form, code, pipe, xf, wps, rot, scratch, outb, cnt, RB, st):
    def launch(i):
        if form == "p":
            return lib().bd_gemm_partial(xf.data_ptr(), RB, wps[i % rot].data_ptr(), N, K, S, code, scratch.data_ptr(), st)
        return lib().bd_gemm_bf16(xf.data_ptr(), RB, wps[i % rot].data_ptr(), None, N, K, S, code, scratch.data_ptr(),
                                  cnt.data_ptr(), outb.data_ptr(), st)
    if launch(0) != 0:
        print(f"{name:9s} nw={nw} kw={kw} S={S} {form}: rejected ({lib().bd_last_error().decode()})")
        return
    for i in range(3):
        launch(i)
    reps = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    blocks = N // (32 * nw // kw) * S
    print(f"{name:9s} N={N:6d} K={K:6d} waves={nw:2d} kparts={kw} pipe={pipe} S={S:2d} {'slabs' if form == 'p' else 'reduced'} blocks={blocks:4d} "
          f"{us:7.1f} us {N * K * 2 / us / 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
