"""Ring depth must not change a GEMM's values: slabs of the same (shape, tile form) at ring 2 / 3 / 4 compared bit for bit.
python tools/ring_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd._lib import check, lib                   # noqa: E402


def main():
    l = lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(5)
    M = 128
    for name, N, K, S, nw, kw, rings in (("wo 2x2", 5120, 5120, 3, 4, 2, (2, 3)), ("w2 2x2", 5120, 7680, 3, 4, 2, (2, 3)),
                                         ("qkv 4x1", 15360, 5120, 2, 4, 1, (2, 3, 4)), ("tiny 2x2", 256, 256, 1, 4, 2, (2, 3)),
                                         ("wo 2x2 S1", 5120, 5120, 1, 4, 2, (2, 3)), ("wo 2x2 S2", 5120, 5120, 2, 4, 2, (2, 3))):
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        wp = E.pack_linear([w], "cuda")
        x = torch.randn(M, K, device="cuda", generator=g)
        xf = torch.zeros(M * K, dtype=torch.bfloat16, device="cuda")
        check(l.bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, M // 32, st))
        ref = None
        for ring in rings:
            outs = []
            for rep in range(3):
                out = torch.zeros(S * M * N, dtype=torch.float32, device="cuda")
                check(l.bd_gemm_partial(xf.data_ptr(), M // 32, wp.data_ptr(), N, K, S, nw + 16 * ring + 256 * (kw - 1), out.data_ptr(), st))
                torch.cuda.synchronize()
                outs.append(out)
            stable = all(torch.equal(outs[0], o) for o in outs[1:])
            if ref is None:
                ref = outs[0]
            d = (outs[0] - ref).abs()
            full = (x.to(torch.bfloat16).float() @ w.float().t())
            err = (outs[0].view(S, M, N).sum(0) - full).abs().max().item()
            print(f"{name:12s} ring {ring}: run-to-run identical {stable}; vs ring {rings[0]}: max |d| {d.max().item():.3g} ({int((d > 0).sum())} of {d.numel()} differ); "
                  f"sum of slabs vs fp32 matmul max err {err:.3g}", flush=True)


if __name__ == "__main__":
    main()
