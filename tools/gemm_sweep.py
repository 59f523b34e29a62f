"""GEMM launch-configuration sweeps of rounds 1-3 in ONE script (the three former tools/gemm_sweep{,2,3}.py; the logs under
profiles/r01_gemm_sweep_*.log, r02_gemm_sweep2*.log, r02_gemm_sweep3*.log came from them).  Round 4 stopped sweeping launch
configurations (profiles/r04_probe_*.log: the 128-row GEMMs sit on the per-CU vector-memory cap); kept for reproducing those logs.

  python tools/gemm_sweep.py r1 [--quick]      (waves, split-K) of every 14B shape
  python tools/gemm_sweep.py r2 [name ...]     workgroup shapes x grid split-K x epilogue form at M = 128
  python tools/gemm_sweep.py r3 [128] [512]    ragged tiles; 512-row ring depth / XCD placement / reduced alternatives
"""
import os
import sys

SWEEPS = {}


def _register(tag, src):
    SWEEPS[tag] = src


_register('r1', r'''"""Sweep (nwaves, split-K) of the weight-streaming GEMM on the 14B shapes; prints GB/s of weight bytes.
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
''')

_register('r2', r'''"""Round-2 GEMM sweep at M = 128: workgroup shapes (waves, K-parts), grid split-K and epilogue form (fp32 slabs vs the
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
''')

_register('r3', r'''"""Round-2 (second part) GEMM sweep.  Timing only: the packed weights are random bytes of the right size, rotated past the
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
''')


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in SWEEPS:
        print(__doc__)
        sys.exit(2)
    tag = sys.argv.pop(1)
    g = {"__name__": "__main__", "__file__": os.path.abspath(__file__)}
    exec(compile(SWEEPS[tag], f"gemm_sweep[{tag}]", "exec"), g)
