"""HBM traffic (read AND write) and matrix-pipe occupancy of every weight-streaming GEMM shape of one AR step, from PMC counters.

  1. on the GPU box, each counter group in its OWN pass (MI355X_MICROARCH.md, HBM / PMC-slot sections; --pmc is never combined
     with the sys/hip/hsa trace domains):
       tools/run_pmc_passes.sh            # FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  2. python tools/pmc_gemm_traffic.py parse out.json fetch.db write.db sq.db

`run` launches each (shape, split-K, waves) exactly as the engine configures it at M = 128 -- fp32 slabs where the engine
leaves the slices to the consumer (split-K > 2), the in-launch reduction to bf16 otherwise -- REPS times in a row, preceded by
two calibration dispatches of known size (a 256 MiB fill = pure write, a 256 MiB read-reduce = pure read).  `parse` takes the
gemm dispatches in order, REPS per shape.  FETCH_SIZE on gfx950 tallies a wide (16 B/lane) streaming read's 128 B requests as
64 B: the guide's x2 correction is applied and cross-checked against the calibration read; WRITE_SIZE is scaled by the
calibration fill (the guide calls it uncalibrated).
"""
import json
import os
import sqlite3
import sys

REPS = 4
M = 128
CAL_BYTES = 256 << 20
# name, N, K, split-K, waves, kparts, ring: the in-situ configuration of bench.py's default workload (roofline.per_gemm).  head.ada is
# the per-evaluation launch (the default pipeline runs it grouped over 4 evaluations on the 256-row kernel: MFMA-bound, listed apart)
SHAPES = [("head.ada", 71680, 5120, 1, 9, 1, 2), ("head.qkv", 15360, 5120, 2, 4, 1, 3), ("head.wo", 5120, 5120, 3, 4, 2, 2),
          ("head.w1", 15360, 5120, 2, 4, 1, 3), ("head.w2", 5120, 7680, 3, 4, 2, 2), ("head.cond", 5120, 5120, 3, 4, 2, 2),
          ("proj.fc2", 5120, 5120, 3, 4, 2, 2), ("llm.qkv", 7168, 5120, 4, 8, 2, 2), ("llm.o", 5120, 5120, 3, 4, 2, 2),
          ("llm.gu", 34816, 5120, 1, 8, 1, 2), ("llm.down", 5120, 17408, 9, 8, 1, 2),
          # the grouped adaLN projection as the pipeline launches it: 4 evaluations x 128 rows, launch code 8 waves / ring 2 = the 256-row
          # kernel (gemm_wide_kernel: default-policy weight loads and XCD placement with two row tiles, bd_gemm.hip launch_gemm_wide).
          # (Round 3 listed it with launch code 4 waves, which selects gemm_kernel<4,1,8> -- a different kernel: its 2.16x read ratio and
          # 514 us were not the engine's launch.)
          ("head.ada[x4]", 71680, 5120, 1, 8, 1, 2, 512),
          # round 4 default for one image on one GPU: 16 evaluations x 128 rows = 2048 rows on the LDS-tiled kernel (bd_gemm_tile.hip,
          # register-staged operand fetch): the same 734 MB of weights, 293 MB of bf16 modulation tensor written
          ("head.ada[x16]", 71680, 5120, 1, 8, 1, 2, 2048)]


# num_images = 4 (the eval scripts' batch, eval/eval_dpg.py:44): 512 rows per pass, the engine's launch configurations at that row
# count (bench.py b4.roofline.per_gemm): the 256-row kernel, K slices left to the consumer; the grouped adaLN projection of two
# evaluations = 1024 rows on the LDS-tiled kernel.  Selected with BD_PMC_ROWS=512 -> profiles/r05_pmc_gemm_traffic_rows512.json
# (round 6: the 256 x 128-tile kernel, bd_gemm_half.hip -- launch code + 4096: one K slice with a rounded output for the N = 15360 shapes,
# 3 / 2 slabs for the N = 5120 / 7168 shapes -> profiles/r06_pmc_gemm_traffic_rows512.json)
H = 4096
SHAPES_512 = [("head.qkv", 15360, 5120, 1, 8 + H, 1, 2, 512), ("head.wo", 5120, 5120, 3, 8 + H, 1, 2, 512), ("head.w1", 15360, 5120, 1, 8 + H, 1, 2, 512),
              ("head.w2", 5120, 7680, 3, 8 + H, 1, 2, 512), ("llm.qkv", 7168, 5120, 2, 8 + H, 1, 2, 512), ("llm.o", 5120, 5120, 3, 8 + H, 1, 2, 512),
              ("llm.gu", 34816, 5120, 1, 8 + H, 1, 2, 512), ("llm.down", 5120, 17408, 3, 8 + H, 1, 2, 512),
              ("head.ada[x2]", 71680, 5120, 1, 8, 1, 2, 1024)]
if os.environ.get("BD_PMC_ROWS") == "512":
    SHAPES, M = SHAPES_512, 512
if os.environ.get("BD_PMC_SHAPES"):                      # "name:N:K:S:nw:kw:ring[:rows];..." -- ad-hoc sets (A/B of a launch configuration)
    SHAPES = [tuple([f.split(":")[0]] + [int(v) for v in f.split(":")[1:]]) for f in os.environ["BD_PMC_SHAPES"].split(";") if f]


def _shape(sh):
    """(name, N, K, S, nw, kw, ring, rows): rows defaults to M."""
    return tuple(sh) + ((M,) if len(sh) == 7 else ())


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bitdance_amd import engine as E
    from bitdance_amd._lib import check, lib
    st = torch.cuda.current_stream().cuda_stream
    cal = torch.empty(CAL_BYTES // 4, dtype=torch.float32, device="cuda")
    sink = torch.zeros(64, dtype=torch.int32, device="cuda")
    for _ in range(2):
        cal.fill_(1.0)                                                                  # calibration: 256 MiB written
        check(lib().bd_probe_read(cal.data_ptr(), CAL_BYTES, 1024, sink.data_ptr(), st))  # calibration: 256 MiB read (16 B/lane)
    torch.cuda.synchronize()
    for name, N, K, S, nw, kw, ring, M in map(_shape, SHAPES):
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        wp = E.pack_linear([w], "cuda")
        del w
        x = torch.randn(M, K, device="cuda")
        xf = torch.zeros(M * K, dtype=torch.bfloat16, device="cuda")
        check(lib().bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, M // 32, st))
        out = torch.empty(max(S, 1) * M * N, dtype=torch.float32, device="cuda")
        outb = torch.empty(M * N, dtype=torch.bfloat16, device="cuda")
        cnt = torch.zeros(16384, dtype=torch.int32, device="cuda")
        code = nw + 16 * ring + 256 * (kw - 1)
        for _ in range(REPS):
            if (S > 2 and name not in ("head.cond", "proj.fc2")) or name == "llm.gu" or (M >= 256 and S > 1):      # > 2 slices (256-row passes: > 1): slabs for the consumer
                check(lib().bd_gemm_partial(xf.data_ptr(), M // 32, wp.data_ptr(), N, K, S, code, out.data_ptr(), st))
            else:
                check(lib().bd_gemm_bf16(xf.data_ptr(), M // 32, wp.data_ptr(), None, N, K, S, code, out.data_ptr(), cnt.data_ptr(),
                                         outb.data_ptr(), st))
        torch.cuda.synchronize()
        del wp, out


def _rows(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    # SQ counters come as one row per shader engine: summed per dispatch.  GRBM_GUI_ACTIVE comes once per XCD and every
    # instance reports the whole kernel: averaged (= busy cycles of the dispatch)
    agg = "avg" if counter.startswith("GRBM") else "sum"
    return cur.execute(f"select dispatch_id, name, {agg}(counter_value), max(duration) from pmc_events where counter_name = ? "
                       "group by dispatch_id, name order by dispatch_id", (counter,)).fetchall()


def parse(out_path, fetch_db, write_db=None, sq_db=None):
    def gemms(rows):
        g = [r for r in rows if "gemm_kernel" in r[1] or "gemm_wide" in r[1] or "gemm_tile" in r[1] or "gemm_half" in r[1]]
        assert len(g) == REPS * len(SHAPES), (len(g), REPS * len(SHAPES))
        return g

    def cal(rows, key):
        c = [r for r in rows if key in r[1]]
        return sum(r[2] for r in c[1:]) / max(1, len(c) - 1) if len(c) > 1 else None

    fr = _rows(fetch_db, "FETCH_SIZE")
    fcal = cal(fr, "probe_read_kernel")
    fetch_scale = 2.0 * 1024.0
    note = {"fetch": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction)"}
    if fcal:
        note["fetch_calibration"] = f"256 MiB probe read reported {fcal:.0f} KiB raw -> x{CAL_BYTES / (fcal * 1024):.3f} would be exact"
    wr = _rows(write_db, "WRITE_SIZE") if write_db else None
    wscale = None
    if wr:
        fills = sorted(r[2] for r in wr if "FillFunctor" in r[1])
        wcal = fills[-1] if fills else None                     # the 256 MiB calibration fill is the largest fill of the run
        wscale = CAL_BYTES / wcal if wcal else 1024.0
        note["write"] = (f"WRITE_SIZE KiB x {wscale:.1f} (calibration: the 256 MiB fill reported {wcal:.0f} KiB)" if wcal
                         else "WRITE_SIZE KiB x 1024 (uncalibrated)")
    sq = {c: _rows(sq_db, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")} if sq_db else None
    gf = gemms(fr)
    gw = gemms(wr) if wr else None
    gs = {c: gemms(r) for c, r in sq.items() if r} if sq else {}
    res = {}
    for i, (name, N, K, S, nw, kw, ring, rows) in enumerate(map(_shape, SHAPES)):
        sl = slice(i * REPS + 1, (i + 1) * REPS)              # drop the first launch of each shape (cold TLB / code)
        avg = lambda g: sum(r[2] for r in g[sl]) / (REPS - 1)
        alg = N * K * 2
        e = dict(N=N, K=K, rows=rows, splitk=S, nwaves=nw & 15, kernel=('gemm_half_kernel' if nw & 4096 else 'by launch code'), kparts=kw, ring=ring, fetch_size_kib_raw=round(avg(gf), 1), hbm_read_bytes=round(avg(gf) * fetch_scale),
                 algorithmic_bytes=alg)
        e["read_ratio"] = round(e["hbm_read_bytes"] / alg, 4)
        e["avg_ns"] = round(sum(r[3] for r in gf[sl]) / (REPS - 1))
        if gw:
            e["hbm_write_bytes"] = round(avg(gw) * wscale)
            e["traffic_ratio"] = round((e["hbm_read_bytes"] + e["hbm_write_bytes"]) / alg, 4)
        if gs:
            mf, gui = avg(gs["SQ_VALU_MFMA_BUSY_CYCLES"]), avg(gs["GRBM_GUI_ACTIVE"])
            e["mfma_busy_cycles"], e["gui_active_cycles"] = round(mf), round(gui)
            e["mfma_util"] = round(mf / (gui * 1024.0), 4)    # matrix-pipe busy cycles / (kernel cycles x 256 CUs x 4 SIMDs)
            e["mfma_util_from_flops"] = round(2.0 * rows * N * K / (e["avg_ns"] * 1e-9) / 2.5e15, 4)   # same thing from 2*M*N*K / time / 2.5 PFLOP/s
            # effective shader clock of THIS dispatch = busy cycles / wall time: the nominal 2.5 PFLOP/s assumes 2.4 GHz; a dense GEMM under
            # profiling runs lower (DVFS), so a fraction of the nominal peak understates how busy the matrix pipe was
            e["eff_clock_ghz"] = round(gui / e["avg_ns"], 3)
            e["mfma_frac_at_eff_clock"] = round(e["mfma_util_from_flops"] * 2.4 / max(e["eff_clock_ghz"], 1e-6), 4)
            if "SQ_BUSY_CYCLES" in gs:
                e["sq_busy_cycles"] = round(avg(gs["SQ_BUSY_CYCLES"]))
        res[name] = e
    out = dict(counters="FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate rocprofv3 --pmc passes)",
               notes=note, rows_M=M, reps_averaged=REPS - 1, gemms=res)
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:10s} read {v['hbm_read_bytes'] / 1e6:8.1f} MB  write {v.get('hbm_write_bytes', 0) / 1e6:7.1f} MB  algorithmic "
              f"{v['algorithmic_bytes'] / 1e6:8.1f} MB  x{v.get('traffic_ratio', v['read_ratio'])}  mfma_util {v.get('mfma_util')}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(*sys.argv[2:])
