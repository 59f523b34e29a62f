"""Round-4 A/B of the in-launch split-K reductions of the 128-row weight-streaming GEMM on the head / LLM shapes: the last
arriver reduces alone (RED = 1) against all slices of a tile reducing it together (+32768 in the launch code, bd_gemm_kernel.h
RED = 2), and the same tiles parking fp32 slabs for the consumer.  Random operands (weights rotated past the Infinity Cache);
every distributed result is compared bit for bit with the last-arriver result of the same tiling.
python tools/gemm_exp.py [gemm names]"""
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
    RED2 = 32768
    # (launch code, K slices); every config runs with the last-arriver reduction and with the distributed one
    shapes = (("head.qkv", 15360, 5120, "b", [(code(4, 3, 1), 2), (code(8, 2, 1), 4), (code(8, 2, 1), 2), (code(4, 3, 1), 4), (code(8, 2, 2), 2)]),
              ("head.w1", 15360, 5120, "s", [(code(4, 3, 1), 2), (code(8, 2, 1), 4)]),
              ("head.wo", 5120, 5120, "b", [(code(4, 2, 2), 3), (code(4, 3, 1), 6), (code(8, 2, 2), 6), (code(8, 2, 1), 12), (code(4, 3, 1), 4)]),
              ("head.w2", 5120, 7680, "b", [(code(4, 2, 2), 3), (code(4, 3, 1), 6), (code(8, 2, 2), 6), (code(8, 2, 1), 12)]),
              ("llm.qkv", 7168, 5120, "b", [(code(8, 2, 2), 4), (code(4, 3, 1), 4), (code(8, 2, 1), 8)]),
              ("llm.gu", 34816, 5120, "s", [(code(8, 2, 1), 1), (code(8, 2, 1), 2)]),
              ("llm.down", 5120, 17408, "b", [(code(8, 2, 1), 9), (code(4, 3, 1), 6), (code(8, 2, 1), 12)]))
    only = sys.argv[1:]
    for name, N, K, form, cands in shapes:
        if only and name not in only:
            continue
        wps = weights(N, K)
        xf = (torch.randn(M * K, device=DEV) * 0.5).to(BF16)
        outb = torch.zeros(M * N, dtype=BF16, device=DEV)
        outp = torch.zeros(12 * M * N, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
        # slabs for the consumer (what the engine launches today for S > 2): timing reference
        for cd, S in cands:
            ref = None
            for extra in ((0, RED2) if S > 1 else (0,)):
                def launch(i):
                    w = wps[i % len(wps)].data_ptr()
                    if form == "b":
                        return lib().bd_gemm_bf16(xf.data_ptr(), RB, w, None, N, K, S, cd + extra, outp.data_ptr(), cnt.data_ptr(), outb.data_ptr(), st)
                    return lib().bd_gemm_swiglu_splitk(xf.data_ptr(), RB, w, None, N, K, S, cd + extra, outp.data_ptr(), cnt.data_ptr(), outb.data_ptr(), st)
                waves, ring, kw = cd & 15, (cd >> 4) & 15, ((cd >> 8) & 3) + 1
                tag = f"{name} waves={waves} kparts={kw} ring={ring} S={S} {'distributed' if extra else 'last-arriver'}"
                outb.zero_()
                rc = launch(0)
                torch.cuda.synchronize()
                if rc != 0:
                    print(f"{tag:62s} rejected ({lib().bd_last_error().decode()})", flush=True)
                    continue
                res = outb.clone()
                same = ""
                if extra == 0:
                    ref = res
                elif ref is not None:
                    same = f"bit-identical to last-arriver: {bool(torch.equal(res, ref))}"
                us = timed(launch)
                torch.cuda.synchronize()
                err = int(cnt[16383])
                print(f"{tag:62s} {us:7.1f} us  W {N * K * 2 / us / 1e3:6.0f} GB/s  {same}{'  GAVE UP WAITING' if err else ''}", flush=True)
                cnt.zero_()
            if S > 1:                                   # the same tiles parking slabs for the consumer
                def launch_p(i):
                    return lib().bd_gemm_partial(xf.data_ptr(), RB, wps[i % len(wps)].data_ptr(), N, K, S, cd, outp.data_ptr(), st)
                us = timed(launch_p)
                if us is not None:
                    print(f"{name + f' waves={cd & 15} S={S} slabs for the consumer':62s} {us:7.1f} us", flush=True)
        del wps


if __name__ == "__main__":
    main()
