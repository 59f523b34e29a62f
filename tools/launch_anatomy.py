"""Anatomy of the per-launch fixed cost: where the microseconds of a 128-row GEMM launch go that are not streaming.

Runs one AR step's head sampling (N + 1 evaluations of the 6-block head at true 14B dimensions, one captured graph, replayed) on the
MEASUREMENT build of the library (libbitdance_hip_stamp.so = the product sources with -DBD_GEMM_STAMP, `python -m bitdance_amd.build
--stamp`): thread 0 of every workgroup of every GEMM / row kernel of the chain writes the chip-wide 100 MHz clock (s_memrealtime, 10 ns
resolution, comparable across CUs and kernels) at

  GEMM   0 workgroup start | 1 first A stage in LDS | 2 first W stage landed | 3 K loop done | 4 K parts reduced through LDS
         | 5 slabs drained + ticket taken (in-launch reduction only) | 6 last store drained
  rows   0 start | 6 last store drained

and this tool reduces them, per kernel NAME, over the grid (min / median / max per phase) and over the launches of the chain (mean of
those), plus the boundaries: first workgroup of kernel k + 1 minus last store of kernel k.

  BD_HIP_LIB=bitdance_amd/libbitdance_hip_stamp.so python tools/launch_anatomy.py [bf16|fp8a] [--tp-shard r/N] [n_steps]

The stamps cost time themselves (two extra s_waitcnt in the GEMM prologue, a drained store at the end of every workgroup): the replay
is ~3-5 % slower than the product's; phase RATIOS are what to read."""
import ctypes as C
import os
import sys
from collections import OrderedDict, defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BD_HIP_LIB", os.path.join(ROOT, "bitdance_amd", "libbitdance_hip_stamp.so"))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd._lib import check, lib                    # noqa: E402
from oracle import tiny_models as tm                        # noqa: E402  (shape table + seeded random weights only)
from oracle.true_dims import device_seeded_state            # noqa: E402

TICK_US = 0.01                                             # s_memrealtime: 100 MHz


def stats(v):
    v = np.asarray(v, dtype=np.float64)
    return float(v.min()), float(np.median(v)), float(v.max())


def main():
    args = [a for a in sys.argv[1:]]
    shard = None
    if "--tp-shard" in args:
        i = args.index("--tp-shard")
        shard = tuple(int(v) for v in args[i + 1].split("/"))
        del args[i:i + 2]
    extra = {}
    for a in list(args):
        if a.startswith(("tp.", "tune.", "opt.")):
            k, v = a.split("=")
            extra[k] = int(v)
            args.remove(a)
    wmode = args[0] if args and args[0] in ("bf16", "fp8", "fp8a") else "bf16"
    n = int(args[1]) if len(args) > 1 else 12
    dev = "cuda"
    l = lib()
    for name, (res, at) in {"anatomy_begin": (C.c_int, [C.c_void_p, C.c_longlong]), "anatomy_count": (C.c_int, []),
                            "anatomy_get": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int)])}.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, at
    with torch.cuda.stream(torch.cuda.Stream()):
        cfgd = dict(ch_target=32, ch_cond=5120, ch_latent=5120, depth_latent=6, depth_adanln=2)
        sd = device_seeded_state(tm.head_shapes(cfgd), 101, dev)
        comm = None
        if shard:
            hw = E.HeadWeights.from_state_dict(sd, dev, weights=wmode, tp_rank=shard[0], tp_size=shard[1])
            from bitdance_amd.tp import TPComm, ada_gather_bytes, seq_hbuf_bytes
            comm = TPComm.loopback_rank(shard[0], shard[1], 128 * 5120, dev, gather_bytes=ada_gather_bytes(128, 14 * 5120),
                                        hbuf_bytes=seq_hbuf_bytes(128, 5120))
            comm.set_timeout(5.0)
        else:
            hw = E.HeadWeights.from_state_dict(sd, dev, weights=wmode)
        del sd
        for k, v in extra.items():                          # process-wide GEMM options ("opt.half.form=4": measurement forms of bd_gemm_half.hip)
            if k.startswith("opt."):
                check(l.bd_set_gemm_option(k[4:].encode(), v))
        tune = {k[5:]: v for k, v in extra.items() if k.startswith("tune.")}
        ei = {k: v for k, v in extra.items() if k.startswith("tp.")}
        B = int(os.environ.get("BD_ANATOMY_B", "1"))        # images per pass (4: the 256-row kernel's wait-cycle stamps, "wide:<gemm>")
        eng = E.Engine(hw, None, None, num_images=B, branches=2, device=dev, max_tokens=64, parallel_num=64, tune=tune or None, comm=comm,
                       extra_ints=ei or None)
        g = torch.Generator(device=dev).manual_seed(7)
        cond = torch.randn(2 * B, 64, 5120, device=dev, generator=g)
        noise = torch.randn(1, n + 1, B, 64, 32, device=dev, generator=g)
        eng.set_schedule(n, 7.5, 1)
        eng.load_noise(noise)
        eng.reset([0] * (2 * B))
        eng.set_cond(cond)
        eng.head_sample()                                   # eager warm-up, unstamped (no buffer yet)
        torch.cuda.synchronize()
        buf = torch.zeros(96 << 20, dtype=torch.uint8, device=dev)
        check(l.anatomy_begin(buf.data_ptr(), buf.numel()))
        eng.capture(0)                                      # the stamp regions are baked into the captured kernel arguments
        for _ in range(3):
            eng.reset([0] * (2 * B))
            eng.launch(0)
        torch.cuda.synchronize()
        if comm is not None:
            comm.check()
        words = buf.view(torch.int64).cpu().numpy().astype(np.uint64)
        recs = []
        nm, off, nwg = C.create_string_buffer(64), C.c_longlong(), C.c_int()
        for i in range(l.anatomy_count()):
            check(l.anatomy_get(i, nm, C.byref(off), C.byref(nwg)))
            w = words[off.value: off.value + nwg.value * 8].reshape(nwg.value, 8).astype(np.int64)
            recs.append((nm.value.decode(), w))
    if os.environ.get("BD_ANATOMY_DUMP"):                    # raw stamps of the last three evaluations' launches, for offline analysis
        keep = recs[-3 * 43:]
        np.savez_compressed(os.environ["BD_ANATOMY_DUMP"], names=np.array([n_ for n_, _ in keep]), **{f"w{i}": w for i, (_, w) in enumerate(keep)})
    print(f"# launch anatomy: {wmode} weights, {'rank %d of %d in loop-back' % shard if shard else 'one GPU (tp = 1)'}, {n + 1} evaluations, "
          f"{len(recs)} stamped launches; seq {int(getattr(eng, 'seq_parallel', False))}; times in us (10 ns clock)")
    cfgs = ", ".join(f"{k} S={eng.gemm_config('head.' + k)[0]} code={eng.gemm_config('head.' + k)[1]}" for k in ("qkv", "wo", "w1", "w2"))
    print(f"# launch configurations: {cfgs}")
    # ---- the 256-row kernel (num_images >= 2): shader cycles wave 0 of every workgroup spent waiting, against its whole K loop
    wide = defaultdict(list)
    for name, w in recs:
        if name.startswith("wide:") and w[:, 0].min() > 0:
            wide[name].append(w)
    for name, ws in wide.items():
        loop = np.concatenate([w[:, 5] for w in ws[1:] or ws]).astype(np.float64)
        ww, wa, wb = (np.concatenate([w[:, k] for w in ws[1:] or ws]).astype(np.float64) for k in (1, 2, 4))
        nst = ws[0][:, 7].max()
        span = np.mean([(w[:, 6].max() - w[:, 0].min()) * TICK_US for w in ws[1:] or ws])
        lp = np.mean([np.median(w[:, 3] - w[:, 0]) * TICK_US for w in ws[1:] or ws])
        tail = np.mean([np.median(w[:, 6] - w[:, 3]) * TICK_US for w in ws[1:] or ws])
        tailmax = np.mean([(w[:, 6] - w[:, 3]).max() * TICK_US for w in ws[1:] or ws])
        if name.startswith("wide:half:"):                   # bd_gemm_half.hip: words 1 / 2 / 4 = waiting for the loads it parks / LOAD segments / barriers
            print(f"\n== {name}: {len(ws)} launches x {ws[0].shape[0]} workgroups, {nst} sub-stages per wave group (16 MFMAs = 512 cycles of matrix pipe per group and sub-stage, two groups per SIMD)")
            print(f"   K loop: median {np.median(loop):9.0f} cycles = {np.median(loop) / nst:6.0f} per sub-stage pair (1024 of matrix pipe); of which waiting for its loads {np.median(ww / loop):.3f}, "
                  f"in LOAD segments {np.median(wa / loop):.3f}, at the barriers {np.median(wb / loop):.3f} (medians over workgroups; max loads {np.max(ww / loop):.3f} barrier {np.max(wb / loop):.3f})")
            print(f"   realtime: K loop {lp:7.2f} us, loop end -> last store drained median {tail:6.2f} / max {tailmax:6.2f} us, kernel span {span:7.2f} us")
            continue
        print(f"\n== {name}: {len(ws)} launches x {ws[0].shape[0]} workgroups, {nst} stages of 64 MFMAs per wave (2048 cycles of matrix pipe each)")
        print(f"   K loop: median {np.median(loop):9.0f} cycles = {np.median(loop) / nst:6.0f} per stage; of which waiting for W {np.median(ww / loop):.3f}, "
              f"for A {np.median(wa / loop):.3f}, at the barrier {np.median(wb / loop):.3f} (medians over workgroups; max W {np.max(ww / loop):.3f} A {np.max(wa / loop):.3f} barrier {np.max(wb / loop):.3f})")
        print(f"   realtime: K loop {lp:7.2f} us, loop end -> last store drained median {tail:6.2f} / max {tailmax:6.2f} us, kernel span {span:7.2f} us")
    recs = [(n_, w) for n_, w in recs if not n_.startswith("wide:")]
    sp = [w for n_, w in recs if n_ == "ln_mod_sp" and w[:, 0].min() > 0 and w[:, 5].min() > 0]
    if sp:
        lab = ["start -> local loads landed", "-> partial flags of every peer seen", "-> staged partials summed, residual written", "-> LayerNorm statistics",
               "-> operand rows pushed and drained", "-> flags raised, end"]
        print(f"\n== ln_mod_sp, phase by phase ({len(sp)} launches x {sp[0].shape[0]} workgroups; median over workgroups, mean over launches)")
        for k, name in enumerate(lab):
            a_, b_ = (0, 1) if k == 0 else ((k, k + 1) if k < 5 else (5, 6))
            print(f"   {name:48s} {np.mean([np.median(w[:, b_] - w[:, a_]) for w in sp[2:] or sp]) * TICK_US:6.2f}")
    # ---- per kernel name: phases reduced over the grid, then averaged over the launches
    phases = OrderedDict([("dispatch skew (start - first start)", (None, 0)), ("start -> first A in LDS", (0, 1)), ("start -> first W landed", (0, 2)),
                          ("first W -> K loop done", (2, 3)), ("K parts through LDS", (3, 4)), ("slabs drained + ticket", (4, 5)),
                          ("ticket -> last store drained", (5, 6)), ("whole workgroup (start -> drained)", (0, 6))])
    by = defaultdict(list)
    for name, w in recs:
        by[name].append(w)
    skip_first = 1                                         # the first evaluation's launches follow the cond / adaLN group GEMMs
    for name, ws in by.items():
        is_gemm = name.startswith("head.") or name == "gemm"
        ws = [w for w in ws if w[:, 0].min() > 0]
        if not ws:
            continue
        print(f"\n== {name}: {len(ws)} launches x {ws[0].shape[0]} workgroups")
        rows = []
        for label, (a, b) in phases.items():
            if not is_gemm and label not in ("dispatch skew (start - first start)", "whole workgroup (start -> drained)"):
                continue
            acc = []
            for w in ws[skip_first:] or ws:
                if a is None:
                    v = (w[:, 0] - w[:, 0].min()) * TICK_US
                else:
                    va, vb = w[:, a], w[:, b]
                    if label == "slabs drained + ticket" and (w[:, 5] == 0).all():
                        continue
                    if label == "ticket -> last store drained":
                        va = np.where(w[:, 5] > 0, w[:, 5], w[:, 4])
                    ok = (va > 0) & (vb > 0)
                    if not ok.any():
                        continue
                    v = (vb[ok] - va[ok]) * TICK_US
                acc.append(stats(v))
            if acc:
                m = np.mean(np.asarray(acc), axis=0)
                rows.append((label, m))
        span = np.mean([(w[:, 6].max() - w[:, 0].min()) * TICK_US for w in (ws[skip_first:] or ws)])
        tail = np.mean([(w[:, 6].max() - np.median(w[:, 6])) * TICK_US for w in (ws[skip_first:] or ws)])
        for label, m in rows:
            print(f"   {label:40s} min {m[0]:7.2f}  median {m[1]:7.2f}  max {m[2]:7.2f}")
        print(f"   {'kernel span (first start -> last drained)':40s} {span:7.2f}    last workgroup behind the median one: {tail:5.2f}")
        if is_gemm and (ws[0][:, 7] & 1).any():
            la = np.mean([((w[:, 6] - w[:, 5])[(w[:, 7] & 1) == 1] * TICK_US).mean() for w in (ws[skip_first:] or ws)])
            print(f"   {'last arriver: ticket -> stores drained':40s} mean {la:7.2f}")
    # ---- the chain: boundaries between consecutive stamped launches
    print("\n== boundaries (first workgroup start of the next launch - last drained store of this one), by (this -> next)")
    gaps = defaultdict(list)
    for (n0, w0), (n1, w1) in zip(recs[:-1], recs[1:]):
        if w0[:, 6].max() > 0 and w1[:, 0].min() > 0:
            gaps[(n0, n1)].append((w1[:, 0].min() - w0[:, 6].max()) * TICK_US)
    for (n0, n1), v in gaps.items():
        if len(v) >= 3:
            print(f"   {n0:14s} -> {n1:14s} n {len(v):4d}  min {min(v):6.2f}  median {float(np.median(v)):6.2f}  max {max(v):6.2f}")
    # ---- one evaluation's budget
    ev = [i for i, (nm_, _) in enumerate(recs) if nm_ == "head_final"]
    if len(ev) >= 3:
        a, b = ev[-3], ev[-2]
        t0 = recs[a][1][:, 6].max()
        t1 = recs[b][1][:, 6].max()
        inside = sum((w[:, 6].max() - w[:, 0].min()) for _, w in recs[a + 1: b + 1]) * TICK_US
        print(f"\n== one evaluation (head_final to head_final): {(t1 - t0) * TICK_US:8.2f} us wall; sum of kernel spans {inside:8.2f}; boundaries + unstamped kernels "
              f"{(t1 - t0) * TICK_US - inside:7.2f} over {b - a} stamped launches")


if __name__ == "__main__":
    main()
