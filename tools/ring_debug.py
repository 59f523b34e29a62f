import os, sys
import torch
sys.path.insert(0, os.getcwd())
from bitdance_amd import engine as E
from bitdance_amd._lib import check, lib
l = lib(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(5)
M = 128
NW, KWP = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 2)
def run(N, K, S, nw, kw, ring, xf, wp):
    out = torch.zeros(S * M * N, dtype=torch.float32, device="cuda")
    check(l.bd_gemm_partial(xf.data_ptr(), M // 32, wp.data_ptr(), N, K, S, nw + 16 * ring + 256 * (kw - 1), out.data_ptr(), st))
    torch.cuda.synchronize()
    return out.view(S, M, N)
for N, K, S in [(5120, 5120, 3), (256, 5120, 3), (5120, 5120, 4), (5120, 5120, 5), (5120, 1792, 1), (5120, 1536, 1), (5120, 1664, 1), (5120, 3584, 2)] + [(1024, 64 * KWP * n, 1) for n in range(1, 31)]:
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    wp = E.pack_linear([w], "cuda")
    x = torch.randn(M, K, device="cuda", generator=g)
    xf = torch.zeros(M * K, dtype=torch.bfloat16, device="cuda")
    check(l.bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, M // 32, st))
    a, b = run(N, K, S, NW, KWP, 2, xf, wp), run(N, K, S, NW, KWP, int(sys.argv[3]) if len(sys.argv) > 3 else 3, xf, wp)
    d = (a != b)
    nst = K // (64 * KWP)
    q = (nst + S - 1) // S
    print(f"N={N} K={K} S={S} stages/slice {[min(q, nst - s * q) for s in range(S)]}: differ {int(d.sum())}", end="")
    if d.any():
        idx = d.nonzero()
        print("  slabs", sorted(set(idx[:, 0].tolist())), " rows", sorted(set(idx[:, 1].tolist()))[:40], " ncols", len(set(idx[:, 2].tolist())), "col sample", sorted(set(idx[:, 2].tolist()))[:6])
    else:
        print()
