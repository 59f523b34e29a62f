"""Round-4 A/B of the 128-row weight-streaming GEMM loops on the head / LLM shapes: the plain loop against the interleaved loop
(+16384 in the launch code, bd_gemm_kernel.h MODE 4).  Random operands (weights rotated past the Infinity Cache); every
interleaved result is compared bit for bit with the plain loop's.
python tools/gemm_exp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


def timed(launch, reps=40):
    if launch(0) != 0:
        return None
    for i in range(4):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def weights(N, K):
    rot = max(3, min(10, int(900e6 // (N * K * 2))))
    g = torch.Generator(device=DEV).manual_seed(N + K)
    return [(torch.randn(N * K, device=DEV, generator=g) * 0.05).to(BF16) for _ in range(rot)]


def code(nw, ring, kw, extra=0):
    return nw + 16 * ring + 256 * (kw - 1) + extra


def main():
    M, RB = 128, 4
    st = torch.cuda.current_stream().cuda_stream
    NEW = 16384
    shapes = (("head.qkv", 15360, 5120, "b", [(code(4, 3, 1), 2), (code(4, 2, 2), 1), (code(4, 3, 2), 1), (code(4, 4, 1), 2), (code(4, 2, 1), 2)]),
              ("head.w1", 15360, 5120, "s", [(code(4, 3, 1), 2), (code(4, 2, 2), 1)]),
              ("head.wo", 5120, 5120, "p", [(code(4, 2, 2), 3), (code(4, 3, 2), 3), (code(4, 2, 2), 2), (code(4, 3, 1), 6), (code(4, 3, 1), 4)]),
              ("head.w2", 5120, 7680, "p", [(code(4, 2, 2), 3), (code(4, 3, 2), 3), (code(4, 2, 2), 2), (code(4, 3, 1), 6)]),
              ("llm.qkv", 7168, 5120, "p", [(code(8, 2, 2), 4), (code(4, 3, 1), 4), (code(4, 2, 2), 2)]),
              ("llm.down", 5120, 17408, "p", [(code(8, 2, 1), 9), (code(4, 2, 2), 3), (code(4, 3, 1), 6)]))
    for name, N, K, form, cands in shapes:
        wps = weights(N, K)
        xf = (torch.randn(M * K, device=DEV) * 0.5).to(BF16)
        outb = torch.zeros(M * N, dtype=BF16, device=DEV)
        outp = torch.zeros(9 * M * N, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
        for cd, S in cands:
            ref = None
            for extra in (0, NEW):
                def launch(i):
                    w = wps[i % len(wps)].data_ptr()
                    if form == "b":
                        return lib().bd_gemm_bf16(xf.data_ptr(), RB, w, None, N, K, S, cd + extra, outp.data_ptr(), cnt.data_ptr(), outb.data_ptr(), st)
                    if form == "s":
                        return lib().bd_gemm_swiglu_splitk(xf.data_ptr(), RB, w, None, N, K, S, cd + extra, outp.data_ptr(), cnt.data_ptr(), outb.data_ptr(), st)
                    return lib().bd_gemm_partial(xf.data_ptr(), RB, w, N, K, S, cd + extra, outp.data_ptr(), st)
                waves, ring, kw = cd & 15, (cd >> 4) & 15, ((cd >> 8) & 3) + 1
                tag = f"{name} waves={waves} kparts={kw} ring={ring} S={S} {'interleaved' if extra else 'plain'}"
                outb.zero_(); outp.zero_()
                rc = launch(0)
                torch.cuda.synchronize()
                if rc != 0:
                    print(f"{tag:58s} rejected ({lib().bd_last_error().decode()})", flush=True)
                    continue
                res = (outp[: S * M * N].clone() if form == "p" else outb.clone())
                same = ""
                if extra == 0:
                    ref = res
                elif ref is not None:
                    same = f"bit-identical to plain: {bool(torch.equal(res, ref))}"
                us = timed(launch)
                print(f"{tag:58s} {us:7.1f} us  W {N * K * 2 / us / 1e3:6.0f} GB/s  {same}", flush=True)
        del wps


if __name__ == "__main__":
    main()
