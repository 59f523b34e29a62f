"""Round-2 (second part) GEMM sweep.  Timing only: the packed weights are random bytes of the right size, rotated past the
256 MB Infinity Cache.
  * 128 rows: ragged workgroup shapes (9 waves for the adaLN projection, 5 for gate/up) against the current ones;
  * 512 rows (num_images = 4): the 256-row kernel's weight-ring depth and XCD placement, and the 128-column / in-launch
    reduced alternatives for the N = 5120 shapes.
python tools/gemm_sweep3.py [128] [512]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd._lib import check, lib                   # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


def timed(launch, reps=30):
    if launch(0) != 0:
        return None
    for i in range(3):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def weights(N, K):
    rot = max(2, min(8, int(700e6 // (N * K * 2))))
    return [torch.randn(N * K // 2, device=DEV).view(BF16) for _ in range(rot)]     # random bits: timing only


def report(tag, N, K, M, us, extra=""):
    if us is None:
        print(f"{tag:34s} rejected ({lib().bd_last_error().decode()})", flush=True)
        return
    print(f"{tag:34s} N={N:6d} K={K:6d} M={M:4d} {us:8.1f} us {N * K * 2 / us / 1e3:6.0f} GB/s "
          f"{2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s {extra}", flush=True)


def sweep128():
    M, RB = 128, 4
    st = torch.cuda.current_stream().cuda_stream
    for name, N, K, form, cands in (
            ("head.ada", 71680, 5120, "b", [10, 9, 8]),
            ("llm.gu", 34816, 5120, "s", [8, 5, 10 + 256, 8 + 256]),
            ("llm.gu(slabs)", 34816, 5120, "p", [8, 10 + 256])):
        wps = weights(N, K)
        xf = torch.zeros(M * K, dtype=BF16, device=DEV)
        outb = torch.empty(M * N, dtype=BF16, device=DEV)
        outp = torch.empty(M * N, dtype=torch.float32, device=DEV) if form == "p" else None
        for nw in cands:
            def launch(i):
                w = wps[i % len(wps)].data_ptr()
                if form == "b":
                    return lib().bd_gemm_bf16(xf.data_ptr(), RB, w, None, N, K, 1, nw, None, None, outb.data_ptr(), st)
                if form == "s":
                    return lib().bd_gemm_swiglu(xf.data_ptr(), RB, w, None, N, K, nw, outb.data_ptr(), st)
                return lib().bd_gemm_partial(xf.data_ptr(), RB, w, N, K, 1, nw, outp.data_ptr(), st)
            waves, kw = nw & 15, ((nw >> 8) & 3) + 1
            blocks = (N // 32 + waves // kw - 1) // (waves // kw)
            report(f"{name} waves={waves} kparts={kw} blocks={blocks}", N, K, M, timed(launch))
        del wps


def sweep512():
    M, RB = 512, 16
    st = torch.cuda.current_stream().cuda_stream
    shapes = (("head.qkv", 15360, 5120, 2), ("head.wo", 5120, 5120, 5), ("head.w2", 5120, 7680, 5), ("head.ada", 71680, 5120, 1),
              ("llm.gu", 34816, 5120, 1))
    for name, N, K, S in shapes:
        wps = weights(N, K)
        xf = torch.zeros(M * K, dtype=BF16, device=DEV)
        outp = torch.empty(max(S, 6) * M * N, dtype=torch.float32, device=DEV)
        outb = torch.empty(M * N, dtype=BF16, device=DEV)
        cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
        for ring in (2, 3):
            for xcd in (0, 1):
                check(lib().bd_set_gemm_option(b"wide.ring", ring))
                check(lib().bd_set_gemm_option(b"wide.xcd", xcd))

                def launch(i):
                    return lib().bd_gemm_partial(xf.data_ptr(), RB, wps[i % len(wps)].data_ptr(), N, K, S, 8, outp.data_ptr(), st)
                report(f"{name} wide S={S} ring={ring} xcd={xcd}", N, K, M, timed(launch))
        check(lib().bd_set_gemm_option(b"wide.ring", 2))
        check(lib().bd_set_gemm_option(b"wide.xcd", -1))
        if N == 15360:
            # 256 rows x 128 columns per workgroup (4 waves x 1 panel): N / 128 tiles x 2 row tiles = 240 workgroups with NO K split,
            # so the bf16 result is written straight from the accumulators (no fp32 slabs for the consumer)
            for nw4 in (4, 8):                                     # 8: the 256-column kernel at S = 1 (120 workgroups) for comparison
                def launch(i):
                    return lib().bd_gemm_bf16(xf.data_ptr(), RB, wps[i % len(wps)].data_ptr(), None, N, K, 1, nw4, None, None,
                                              outb.data_ptr(), st)
                report(f"{name} {nw4} waves S=1 bf16 direct", N, K, M, timed(launch))
        if N == 5120:
            # 128-column tiles (4 waves x 256 rows), fewer slices, reduced inside the launch: no slabs for the consumer
            for S2 in (2, 3):
                def launch(i):
                    return lib().bd_gemm_bf16(xf.data_ptr(), RB, wps[i % len(wps)].data_ptr(), None, N, K, S2, 4, outp.data_ptr(),
                                              cnt.data_ptr(), outb.data_ptr(), st)
                report(f"{name} 4 waves x 256 rows S={S2} reduced", N, K, M, timed(launch))
            for S2 in (3, 6):
                def launch(i):
                    return lib().bd_gemm_partial(xf.data_ptr(), RB, wps[i % len(wps)].data_ptr(), N, K, S2, 4, outp.data_ptr(), st)
                report(f"{name} 4 waves x 256 rows S={S2} slabs", N, K, M, timed(launch))
        del wps, outp


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in ("128", "512")] or ["128", "512"]
    if "128" in which:
        sweep128()
    if "512" in which:
        sweep512()
