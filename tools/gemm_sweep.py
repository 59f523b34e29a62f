"""Sweep (nwaves, split-K) of the weight-streaming GEMM on the 14B shapes; prints GB/s of weight bytes.
python tools/gemm_sweep.py [--quick]"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV = "cuda"
SHAPES = [("head.ada", 71680, 5120, False), ("head.qkv", 15360, 5120, False), ("head.wo", 5120, 5120, False),
          ("head.w1", 15360, 5120, True), ("head.w2", 5120, 7680, False), ("llm.qkv", 7168, 5120, False),
          ("llm.gu", 34816, 5120, True), ("llm.down", 5120, 17408, False)]


def main():
    quick = "--quick" in sys.argv
    M = int(sys.argv[sys.argv.index("--M") + 1]) if "--M" in sys.argv else 128
    RB = M // 32
    res = []
    st = torch.cuda.current_stream().cuda_stream
    for name, N, K, swiglu in SHAPES:
        w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
        wp = E.pack_swiglu(w[: N // 2], w[N // 2:], DEV) if swiglu else E.pack_linear([w], DEV)
        del w
        x = torch.randn(M, K, device=DEV)
        xf = torch.zeros(M * K, dtype=torch.bfloat16, device=DEV)
        check(lib().bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, RB, st))
        out = torch.empty(16 * M * N if not swiglu else M * N, dtype=torch.float32, device=DEV)
        Ss = [1] if swiglu else [1, 2, 3, 4, 6, 8, 12]
        kws = "--kw" in sys.argv
        cfgs = ([(4, 2, 1), (8, 2, 1), (4, 2, 2), (8, 2, 2), (10, 2, 2)] if kws else
                [(2, 2, 1), (4, 2, 1), (4, 3, 1), (8, 2, 1), (8, 3, 1), (8, 4, 1)]) if M == 128 else [(4, 2, 1), (8, 2, 1)]
        for nw, ring, kw in cfgs:
            if N % (32 * nw // kw) or K % (64 * kw):
                continue
            code = nw + 16 * ring + 256 * (kw - 1)
            for S in Ss:
                nst = K // (64 * kw)
                if S > nst or (S > 1 and (S - 1) * ((nst + S - 1) // S) >= nst):
                    continue
                blocks = N // (32 * nw // kw) * S
                blocks *= max(1, M // 256)
                if blocks < 100 or blocks > 1300:
                    continue

                def launch():
                    if swiglu:
                        check(lib().bd_gemm_swiglu(xf.data_ptr(), RB, wp.data_ptr(), None, N, K, code, out.data_ptr(), st))
                    else:
                        check(lib().bd_gemm_partial(xf.data_ptr(), RB, wp.data_ptr(), N, K, S, code, out.data_ptr(), st))
                for _ in range(3):
                    launch()
                reps = 10 if quick else 30
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    launch()
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                gbs = N * K * 2 / us / 1e3
                tfs = 2.0 * M * N * K / us / 1e6
                res.append(dict(name=name, N=N, K=K, nw=nw, ring=ring, kw=kw, S=S, blocks=blocks, us=round(us, 1), GBs=round(gbs)))
                print(f"{name:9s} N={N:6d} K={K:6d} nw={nw} kw={kw} R={ring} S={S:2d} blocks={blocks:5d}  {us:8.1f} us  {gbs:7.0f} GB/s  {tfs:6.0f} TFLOP/s", flush=True)
        del wp
    best = {}
    for r in res:
        if r["name"] not in best or r["GBs"] > best[r["name"]]["GBs"]:
            best[r["name"]] = r
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
