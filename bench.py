#!/usr/bin/env python
"""Benchmark of the BitDance generation hot path on N MI355X (BASELINE.json metric: images/sec @1024px BitDance-14B-64x).

    python bench.py --gpus N --steps K --warmup W          (N > 1 from a plain interpreter: bench.py starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one whole pass of the hot path: one ``gen_image`` call (prefill, 64 AR steps x [51 diffusion-head
evaluations + sign + projector + cond/uncond LLM forward], AE decode) for ``--num-images`` images on synthetic
(random-weight, true-shape) models with inputs resident in HBM.

N > 1, ``--parallel tp`` (default, the north star's mode): ONE job, the 14B model tensor-parallel over the N GPUs
(bitdance_amd/tp.py: column/row-split Linears, kv cache by head, one hand-written xGMI exchange per row-split Linear inside
the step graphs); value = images / max-over-ranks time, "scaling": "strong".  ``--parallel replicas``: one full replica per
GPU over disjoint images (what the reference's own multi-GPU evaluation does, eval/eval_dpg.py:25-29): "scaling": "weak".

``--workload`` selects the other BASELINE configs: ``14b-16x-512`` (config 3: BitDance-14B-16x, 512 px) and
``imagenet-b16x`` (config 2: class-conditional BitDance-B-16x, 256 px, batch 384, 100 sampling steps, linear CFG 6.1,
imagenet_gen/sample_ddp_parallel.py:199-214); each prints its own metric name -- only the default is the headline.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline"     : the dominant kernel family (the weight-streaming / MFMA GEMM), measured live with HIP events on the
                   pipeline's stream: achieved = weight bytes physically streamed by every GEMM launch / time vs the 8 TB/s peak at
                   M <= 256 rows ("algorithmic": N*K*2 per Linear and evaluation over the same time -- a grouped launch serves
                   several evaluations with one pass; "hbm_bound_launches": the one-evaluation launches alone), or TFLOP/s vs
                   2.5 PFLOP/s beyond; "traffic": HBM bytes per launch from the committed PMC passes
  "cpu_baseline" : the CPU oracle (a port of the reference algorithm) timed on this box's host cores on a bounded sample at
                   TRUE dimensions as SURVEY 8(d) specifies it (oracle/true_dims.py: one FULL 6-block head evaluation, one
                   Qwen3-14B layer step against ~2k cached tokens, one 256 x 256 ae_d16c32 decode), extrapolated to one image
                   by counts only -- and, from the same sample, "parity": the max / mean error of the HIP path against that
                   oracle output on identical inputs
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFS = 2500.0         # dense bf16 MFMA peak

WORKLOADS = {
    # name: (pipeline size, height, width, metric)
    "14b-64x-1024": ("14b-64x", 1024, 1024, "images/sec @1024px BitDance-14B-64x"),
    "14b-16x-512": ("14b-16x", 512, 512, "images/sec @512px BitDance-14B-16x"),
    "tiny": ("tiny", 256, 256, "images/sec (tiny smoke config)"),
    "imagenet-b16x": (None, 256, 256, "images/sec @256px ImageNet BitDance-B-16x"),
    # the other released ImageNet checkpoints (imagenet_gen/README.md:10-15): 4x parallel variant; 1x = causal transformer + MLP head
    "imagenet-b4x": (None, 256, 256, "images/sec @256px ImageNet BitDance-B-4x"),
    "imagenet-b1x": (None, 256, 256, "images/sec @256px ImageNet BitDance-B-1x"),
    "imagenet-l1x": (None, 256, 256, "images/sec @256px ImageNet BitDance-L-1x"),
    "imagenet-h1x": (None, 256, 256, "images/sec @256px ImageNet BitDance-H-1x"),
    # the tokenizer's conv decoder alone on a random +-1 latent (config 5's ae_d32c256 decoder cannot be fed by a 14B
    # checkpoint with a 32-channel head, so it is benchmarked standalone; ae-d16c32 = the decoder the headline pipeline ends in)
    "ae-d32c256-decode": (None, 1024, 1024, "images/sec @1024px ae_d32c256 decoder (standalone)"),
    "ae-d16c32-decode": (None, 1024, 1024, "images/sec @1024px ae_d16c32 decoder (standalone)"),
}
# README defaults of the reference's sampler per checkpoint: (classes per call, CFG scale)   imagenet_gen/README.md:27-80
IMAGENET_DEFAULTS = {"b16x": (384, 6.1), "b4x": (384, 3.9), "b1x": (384, 3.2), "l1x": (352, 4.0), "h1x": (224, 4.55)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS))
    ap.add_argument("--size", default=None, choices=["14b-64x", "tiny"], help="(old spelling of --workload)")
    ap.add_argument("--parallel", default="tp", choices=["tp", "replicas"], help="N > 1: tensor parallel (one job) or replicas")
    ap.add_argument("--tp-comm", default=None, choices=["ipc", "rccl"],
                    help="tensor parallel: the per-Linear exchange as the hand-written xGMI push kernel (ipc, default) or ncclAllReduce (rccl) -- "
                         "one lease can A/B them; default: BD_TP_COMM or ipc (falls back to rccl by itself when IPC / uncached memory / the "
                         "self-test fails)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--num-images", type=int, default=None, help="images per gen_image call (imagenet: classes per sample call, default 384)")
    ap.add_argument("--sampling-steps", type=int, default=None)
    ap.add_argument("--guidance", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-b4", action="store_true", help="headline workload on one GPU: skip the extra num_images=4 point (SURVEY 8(d): the eval scripts' batch)")
    ap.add_argument("--no-throughput", action="store_true", help="headline workload on one GPU: skip the extra saturated-batch point")
    ap.add_argument("--throughput-images", type=int, default=16, help="num_images of the `throughput` point (default 16: where one GPU saturates -- 0.55 / 0.575 / 0.576 images/s at 8 / 16 / 32)")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1, tensor parallel: skip the independent-replicas point timed after the tensor-parallel pass")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="imagenet: skip the conv decoder in the timed pass")
    ap.add_argument("--tp-ada-split", default="auto", choices=["auto", "0", "1"],
                    help="tensor parallel: column-split the adaLN projection and all-gather its output (auto: from 4 ranks up)")
    ap.add_argument("--tp-seq", default="auto", choices=["auto", "0", "1"],
                    help="tensor parallel: sequence-parallel row kernels (csrc/bd_sp.hip: a rank owns rows / N rows of the head's residual stream, "
                         "no stand-alone exchange kernel in an evaluation) instead of all-reduce + replicated row kernels (auto: up to 4 ranks)")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8", "fp8a"],
                    help="fp8: streamed Linear weights as e4m3 + per-channel scales (BASELINE config 5; a separate precision mode); "
                         "fp8a: also e4m3 activations (per-row scales) on the fp8 matrix pipe for the GEMMs fed by a row kernel")
    ap.add_argument("--attn-splits", type=int, default=None, help="KV splits of the LLM decode attention (default: 12 beyond 2k cached tokens, else 8)")
    ap.add_argument("--tune", default="", help="comma list name.S=4,name.nw=2,kw2=0 overriding GEMM launch configs")
    a = ap.parse_args()
    if a.workload is None:
        a.workload = "tiny" if a.size == "tiny" else "14b-64x-1024"
    return a


# ---------------------------------------------------------------------------------------------------------
def pmc_entries(rows: int) -> dict:
    """The committed PMC passes for GEMM launches at ``rows`` rows (tools/pmc_gemm_traffic.py; newest round first)."""
    tag = "" if rows == 128 else f"_rows{rows}"
    for r in ("r06", "r05", "archive/r04"):
        pj = os.path.join(ROOT, "profiles", f"{r}_pmc_gemm_traffic{tag}.json")
        if os.path.exists(pj):
            d = json.load(open(pj))
            if d.get("rows_M", 128) == rows:
                return dict(d["gemms"], _source=f"profiles/{r}_pmc_gemm_traffic{tag}.json")
    return {}


def annotate_pmc(per: list, rows: int) -> str | None:
    """mfma_busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)) and the effective shader clock (GRBM_GUI_ACTIVE / kernel
    time) of the PMC pass beside every per-GEMM row whose launch configuration that pass measured: an MFMA fraction against the
    NOMINAL 2.5 PFLOP/s peak means little without the clock the kernel actually ran at."""
    pm = pmc_entries(rows)
    if not pm:
        return None
    for q in per:
        e = pm.get(q["name"])
        if e is None or e.get("splitk") != q["splitk"] or e.get("nwaves") != q["nwaves"] or e.get("kparts", 1) != q["kparts"]:
            continue
        if "mfma_util" in e:
            q["mfma_busy"] = e["mfma_util"]
        if e.get("gui_active_cycles") and e.get("avg_ns"):
            q["eff_clock_ghz"] = round(e["gui_active_cycles"] / e["avg_ns"], 3)
    return pm.get("_source")


def gemm_roofline(eng, run, rows: int) -> dict:
    """In-situ roofline of the dominant kernel family (the GEMMs of bd_gemm.hip): ``run`` (one AR step's worth of eager
    launches) is executed with every GEMM launch bracketed by HIP events on the launch stream (bd_prof_*), so weights are
    NOT cache-resident between launches.  rows <= 256: HBM weight streaming bounds the launch (arithmetic intensity = rows
    FLOP/B, under the ~310 FLOP/B ridge): achieved = algorithmic weight bytes (N*K*2 per launch) / event time.  More rows
    (several images / the imagenet batch): the MFMA roofline bounds it: achieved = 2*rows*N*K / event time."""
    prof = eng.profile_gemms(run)
    # a launch named "<gemm>[xG]" is ONE pass over the weights for G evaluations' rows (the grouped adaLN projection, bd_api.hip
    # head_ada_group): G * rows rows per pass -- the matrix pipe, not HBM, bounds that launch; it is listed with its TFLOP/s, counts
    # its physically streamed bytes ONCE in "achieved" (G x in "algorithmic") and stays out of the "hbm_bound_launches" sums
    def split(name):
        if name.endswith("]") and "[x" in name:
            base, g = name[:-1].split("[x")
            return base, int(g)
        return name, 1
    per, grouped = [], []
    fam, allg = {}, {}
    for name, r in sorted(prof.items()):
        base, G = split(name)
        S, nw = eng.gemm_config(base)
        rec = {"name": name, "launches": r["count"], "avg_us": round(r["ms"] / r["count"] * 1e3, 2),
               "GBs": round(r["bytes"] / r["ms"] / 1e6, 1), "splitk": S if G == 1 else 1, "nwaves": (nw & 15) if G == 1 else 8,
               "ring": (nw >> 4) & 15 if G == 1 else 2, "kparts": (((nw >> 8) & 3) + 1) if G == 1 else 1}
        allg[name] = dict(r, G=G)
        if G * rows > 256 and rows <= 256:
            rec["rows_per_pass"] = G * rows
            rec["evaluations_per_launch"] = G
            rec["algorithmic_GBs"] = round(r["bytes"] * G / r["ms"] / 1e6, 1)
            bpw = 1 if eng.wdtype else 2                          # bytes per weight: 2 * rows * N * K flop = bytes * rows * 2 / bpw
            peak = MFMA_PEAK_TFS * (2 if eng.wdtype == 2 else 1)  # fp8 x fp8 MFMA: twice the bf16 rate
            rec["TFLOPs"] = round(r["bytes"] * (2 / bpw) * G * rows / r["ms"] / 1e9, 1)
            rec["frac_of_mfma_peak"] = round(rec["TFLOPs"] / peak, 4)
            rec["note"] = (f"ONE pass over the weights serves {G} evaluations ({'LDS-tiled 256 x 256 kernel' if G * rows >= 1024 else '256-row kernel'}, MFMA-bound at {G * rows} rows): GBs = bytes "
                           f"physically streamed / time, algorithmic_GBs = {G} x N*K*2 (the reference streams them once per evaluation) / time")
            grouped.append(rec)
        else:
            fam[name] = r
        per.append(rec)
    tot_b = sum(r["bytes"] for r in fam.values())
    tot_ms = sum(r["ms"] for r in fam.values())
    n_launch = sum(r["count"] for r in fam.values())
    common = {"kernel": "gemm_kernel<NP,KW,MB,EPI,R,RED> / gemm_wide_kernel (bd_gemm.hip) / gemm_tile_kernel (bd_gemm_tile.hip: grouped adaLN): every GEMM launch of one AR step, in situ",
              "launches": n_launch, "avg_launch_us": round(tot_ms / n_launch * 1e3, 2), "per_gemm": per}
    prof = fam
    pmc_src = annotate_pmc(per, rows)
    if rows > 256:
        flops = sum(r["bytes"] * rows * allg[n]["G"] for n, r in fam.items())   # 2*rows*N*K flop = bytes * rows (x G evaluations' rows per grouped pass)
        ach = flops / tot_ms / 1e9
        for q in per:
            r = allg[q["name"]]
            q["TFLOPs"] = round(r["bytes"] * rows * r["G"] / r["ms"] / 1e9, 1)
            q["frac_of_mfma_peak"] = round(q["TFLOPs"] / MFMA_PEAK_TFS, 4)
        # time-weighted busy fraction / clock of the launches the PMC pass covers (null without a pass at this row count)
        cov = [(q, allg[q["name"]]["ms"]) for q in per if "mfma_busy" in q]
        tw = sum(ms for _, ms in cov)
        busy = round(sum(q["mfma_busy"] * ms for q, ms in cov) / tw, 4) if tw > 0 else None
        clk = round(sum(q["eff_clock_ghz"] * ms for q, ms in cov) / tw, 3) if tw > 0 else None
        return {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFS, 4), "traffic": None, "rows": rows, "mfma_busy": busy, "eff_clock_ghz": clk,
                "pmc_source": pmc_src,
                "flop_per_launch": int(flops / n_launch), **common}
    # achieved / frac = bytes PHYSICALLY streamed by ALL GEMM launches of the step / their summed time, against the HBM peak (a launch that
    # serves G evaluations in one pass over its weights -- the grouped adaLN projection -- moved its weights ONCE and counts once: the
    # figure cannot exceed the peak and stays comparable with rounds 1-2).  The reference-equivalent figure (N*K*2 per Linear
    # and EVALUATION, what the reference's loop would stream) is reported beside it as "algorithmic", not as a fraction of the HBM peak.
    all_b = sum(r["bytes"] for r in allg.values())
    alg_b = sum(r["bytes"] * r["G"] for r in allg.values())
    all_ms = sum(r["ms"] for r in allg.values())
    all_n = sum(r["count"] for r in allg.values())
    ach = all_b / all_ms / 1e6
    streamed = {"achieved": round(tot_b / tot_ms / 1e6, 1), "frac": round(tot_b / tot_ms / 1e6 / HBM_PEAK_GBS, 4), "launches": n_launch,
                "bytes_per_launch": int(tot_b / n_launch),
                "note": "the same over the launches with one evaluation per pass only (HBM-bound); the grouped launches are MFMA-bound and carry their TFLOP/s in per_gemm"}
    algorithmic = {"GBs": round(alg_b / all_ms / 1e6, 1), "bytes_per_launch": int(alg_b / all_n),
                   "note": "N*K*2 per Linear and evaluation (what the reference streams) / the same time: a grouped launch counts G x its weights; not a fraction of the HBM peak"}
    common.update(launches=all_n, avg_launch_us=round(all_ms / all_n * 1e3, 2))
    # The path that actually binds the 128-row GEMMs (profiles/r04_probe_mix.log, r04_probe_gemmlike.log): a CU's
    # vector-memory path carries the weights AND every column tile's re-read of the activations AND the split-K slabs.  Bytes through
    # that path per launch = W + (N / tile columns) x rows x K x 2 + S x rows x N x 4 (S > 1); the bound is what a kernel that ONLY
    # issues the qkv GEMM's loads reaches on this chip (10.4 TB/s: 314 MB in 30.2 us, no LDS / MFMA / barrier / epilogue).  Reported
    # beside the HBM figure, never instead of it.
    vpath = None
    kdim = {}
    if getattr(eng, "head", None) is not None:
        h = eng.head
        kdim.update({"head.qkv": h.D, "head.wo": h.D // max(1, h.tp_size), "head.w1": h.D, "head.w2": h.H // max(1, h.tp_size), "head.cond": h.Dz, "head.ada": h.D})
    if getattr(eng, "llm", None) is not None:
        c = eng.llm.cfg
        tpl = max(1, getattr(eng.llm, "tp_size", 1))
        kdim.update({"llm.qkv": c["hidden_size"], "llm.gu": c["hidden_size"], "llm.o": c["num_attention_heads"] * c["head_dim"] // tpl,
                     "llm.down": c["intermediate_size"] // tpl})
    if getattr(eng, "proj", None) is not None:
        kdim["proj.fc2"] = eng.proj.D
    if rows <= 256 and not eng.wdtype:
        pb = pt = 0.0
        rows_pad = 32 if rows <= 32 else (64 if rows <= 64 else (rows + 127) // 128 * 128)
        for q in per:
            name, r = q["name"], allg[q["name"]]
            if name not in kdim or r["G"] != 1:
                continue
            K = kdim[name]
            N = int(round(r["bytes"] / r["count"] / 2 / K))
            cols = 32 * max(1, q["nwaves"] // q["kparts"])
            a_bytes = -(-N // cols) * rows_pad * K * 2
            slab = q["splitk"] * rows_pad * N * 4 if q["splitk"] > 1 else 0
            q["path_GBs"] = round((r["bytes"] / r["count"] + a_bytes + slab) * r["count"] / r["ms"] / 1e6, 1)
            pb += (r["bytes"] / r["count"] + a_bytes + slab) * r["count"]
            pt += r["ms"]
        if pt > 0:
            vpath = {"achieved": round(pb / pt / 1e6, 1), "bound": 10400.0, "unit": "GB/s", "frac": round(pb / pt / 1e6 / 10400.0, 4),
                     "note": "weights + per-tile activation re-reads + split-K slab writes through the CUs' vector-memory path / time, over the "
                             "one-evaluation launches; bound = the same traffic as bare loads (tools/probe_gemmlike.hip, profiles/r04_probe_gemmlike.log)"}
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs, gfx950 x2
    # wide-read correction: tools/pmc_gemm_traffic.py -> profiles/r0*_pmc_gemm_traffic.json), weighted by this step's
    # launch mix; only valid for the shapes / launch configs that pass measured, else null
    traffic, traffic_src = None, None
    for fn in ("r06_pmc_gemm_traffic.json", "r05_pmc_gemm_traffic.json"):
        pj = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(pj) or rows != 128:
            continue
        pm = json.load(open(pj))["gemms"]
        cfgs = {q["name"]: q for q in per}

        def entry(n):                                           # a short last group ("[x3]") streams the same weights as "[x4]"
            if n in pm:
                return pm[n]
            base, G = split(n)
            alts = [v for k, v in pm.items() if G > 1 and k.startswith(base + "[x")]
            return alts[0] if alts else None
        if all(entry(n) is not None and entry(n)["splitk"] == cfgs[n]["splitk"] and entry(n)["nwaves"] == cfgs[n]["nwaves"] and
               entry(n).get("kparts", 1) == cfgs[n]["kparts"] and
               entry(n)["N"] * entry(n)["K"] * 2 * r["count"] == int(r["bytes"]) for n, r in allg.items()):
            traffic = int(sum((entry(n)["hbm_read_bytes"] + entry(n).get("hbm_write_bytes", 0)) * r["count"] for n, r in allg.items()) / all_n)
            traffic_src = f"profiles/{fn} (FETCH_SIZE{' + WRITE_SIZE' if any('hbm_write_bytes' in v for v in pm.values()) else ''}, bytes per launch, launch-mix weighted)"
            break
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": int(all_b / all_n), "hbm_bound_launches": streamed, "algorithmic": algorithmic,
            "vector_memory_path": vpath, "pmc_source": pmc_src, **common}


# ---------------------------------------------------------------------------------------------------------
def cpu_baseline_t2i(args, P: int, ar_steps: int, n_eval: int, px: int = 1024) -> dict:
    """The CPU oracle (port of the reference algorithm, oracle/) on the host cores at TRUE 14B dimensions, as SURVEY 8(d) specifies
    the bounded sample: ONE full head evaluation (all 6 blocks, both adaLN projections, M = 2*P rows: t_head), ONE Qwen3-14B
    decoder-layer step of the cond + uncond sequences (2 x P tokens) against ~2k cached tokens (t_layer), ONE 256 x 256 decode of
    the ae_d16c32 decoder (t_ae256) -- all three through oracle/true_dims.py, the same inputs through the HIP path, the differences
    reported as "parity".  Extrapolation, no scaling of any sample:
        T = AR * (N + 1) * t_head + (AR + 1) * 40 * t_layer + (px / 256)^2 * t_ae256
    (AR * (N + 1) = 3264 evaluations; (AR + 1) * 2 = 130 single-branch forwards incl. the two prompt calls counted as decode-sized
    steps -- SURVEY's 132 counts the discarded last forward too; 16 decoder tiles at 1024 px)."""
    from oracle.true_dims import ae_case, head_case, llm_case
    # (torch's default intra-op thread count = the physical cores: forcing one thread per hardware thread made the oracle 10x slower
    # on the 256-thread hosts of this pool -- 61 s for the head evaluation the test suite runs in 6 s)
    tiny = args.workload == "tiny"
    REP = 3                                                  # each leg >= 3 repeats; value from the MINIMUM, medians listed
    if tiny:
        h = head_case(D=256, P=P, B=1, branches=2, depth=4, nada=2, repeats=REP)
        from oracle.tiny_models import TINY_LLM
        l = llm_case(layers=1, P=P, past=(100, 117), cfg=TINY_LLM, repeats=REP)
        a = None
        nblk, L = 4, 2
    else:
        h = head_case(D=5120, P=P, B=1, branches=2, depth=6, nada=2, repeats=REP)
        l = llm_case(layers=1, P=P, past=(2000, 2017), repeats=REP)
        a = ae_case(config="AE_D16C32", px=256, repeats=REP)
        nblk, L = 6, 40
    tiles = (px / 256.0) ** 2
    t_img = ar_steps * n_eval * h["t_cpu_s"] + (ar_steps + 1) * L * l["t_cpu_s"] + (tiles * a["t_cpu_s"] if a else 0.0)
    ae_txt = f" + one 256 x 256 ae_d16c32 decode ({a['t_cpu_s']:.2f} s)" if a else ""
    par = {"head_xhat_max_err": round(h["max_err"], 5), "head_xhat_mean_err": round(h["mean_err"], 6),
           "head_gemm_cfg": h["gemm_cfg"], "llm_hidden_max_err": round(l["max_err"], 5),
           "llm_hidden_mean_err": round(l["mean_err"], 6), "llm_gemm_cfg": l["gemm_cfg"],
           "bounds": "head x_hat in [-1,1]: max 5e-2 / mean 6e-3; LLM hidden: max 0.12 / mean 1e-2 (tests/test_gpu_true_dims.py); "
                     "decoder image: mean 0.03 x mean|x| + 2e-3 (tests/test_gpu_ae.py)",
           "within_bounds": bool(h["max_err"] <= 5e-2 and h["mean_err"] <= 6e-3 and l["max_err"] <= 0.12 and l["mean_err"] <= 1e-2 and
                                 (a is None or a["mean_err"] <= 0.03 * a["ref_abs_mean"] + 2e-3))}
    if a:
        par.update(ae_image_max_err=round(a["max_err"], 5), ae_image_mean_err=round(a["mean_err"], 6), ae_ref_abs_mean=round(a["ref_abs_mean"], 4))
    t_med = (ar_steps * n_eval * h["t_cpu_median_s"] + (ar_steps + 1) * L * l["t_cpu_median_s"] + (tiles * a["t_cpu_median_s"] if a else 0.0))
    return {"value": round(1.0 / t_img, 8), "unit": "images/s", "cores": os.cpu_count(), "threads": torch.get_num_threads(), "kind": "port",
            "value_from_medians": round(1.0 / t_med, 8), "repeats": REP,
            "legs_s": {"head_eval": {"min": round(h["t_cpu_s"], 3), "median": round(h["t_cpu_median_s"], 3)},
                       "llm_layer": {"min": round(l["t_cpu_s"], 3), "median": round(l["t_cpu_median_s"], 3)},
                       **({"ae_256": {"min": round(a["t_cpu_s"], 3), "median": round(a["t_cpu_median_s"], 3)}} if a else {})},
            "sample": f"the reference itself (/root/reference, pure Python + HF transformers) is not on the GPU box: this is the oracle, a CPU port "
                      f"pinned to the reference's outputs by tests/golden; {REP} repeats per leg, value from the minima, torch intra-op threads = "
                      f"{torch.get_num_threads()} of {os.cpu_count()} hardware threads; "
                      f"oracle (CPU port) at {'tiny' if tiny else 'true 14B'} dimensions, SURVEY 8(d): one FULL head evaluation ({nblk} blocks + "
                      f"the adaLN projections, M = {h['rows']} rows: {h['t_cpu_s']:.2f} s) + one LLM layer step of {l['rows']} tokens against "
                      f"~{'100' if tiny else '2k'} cached ({l['t_cpu_s']:.2f} s){ae_txt}; T = {ar_steps * n_eval} x t_head + "
                      f"{(ar_steps + 1) * L} x t_layer" + (f" + {tiles:.0f} x t_ae256" if a else "") + "; no sample is rescaled",
            "parity": par}


def cpu_baseline_imagenet(n_eval: int, ar_steps: int) -> dict:
    """BitDance-B head at its real dimensions on the host cores (one evaluation of the full 6-block head on 256 rows = 8
    images with CFG); extrapolated by rows and evaluation count; the transformer (5 % of the FLOPs) and the VAE excluded."""
    from oracle.true_dims import head_case
    h = head_case(D=768, Dz=768, C=32, P=16, B=8, branches=2, depth=6, nada=2, head_dim=64, sigmoid=False, seed=109)
    t_img = h["t_cpu_s"] / 8 * n_eval * ar_steps
    return {"value": round(1.0 / t_img, 6), "unit": "images/s", "cores": os.cpu_count(), "threads": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (CPU port): one evaluation of the BitDance-B head (6 blocks, width 768) on 256 rows = 8 images with CFG "
                      f"({h['t_cpu_s']:.2f} s); extrapolated x{n_eval * ar_steps} evaluations per image; transformer and VAE excluded",
            "parity": {"head_out_max_err": round(h["max_err"], 5), "head_out_mean_err": round(h["mean_err"], 6),
                       "ref_abs_mean": round(h["ref_abs_mean"], 4)}}


# ---------------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` from a plain interpreter: start the N ranks ourselves (one process per GPU, the launch shape
    of the reference's multi-GPU scripts, scripts/eval/eval_bitdance_14b_64x.sh:4-16, and of the driver's own
    ``python -m torch.distributed.run --nproc-per-node N`` line) and hand the job to them; rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("BD_BENCH_BACKEND", "nccl") == "nccl":
        print(f"[bench] --gpus {args.gpus} but only {n_dev} device(s) visible (RCCL needs one device per rank; "
              f"BD_BENCH_BACKEND=gloo runs several ranks on one GPU as a functional check)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what hipIpcGetMemHandle needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def setup_dist():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL ("nccl") on a real node.  BD_BENCH_BACKEND=gloo: several ranks sharing one GPU (functional check of the
        # tensor-parallel path on a single-GPU box; RCCL refuses two ranks on one device)
        backend = os.environ.get("BD_BENCH_BACKEND", "nccl")
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        local = 0
        torch.cuda.set_device(0)
    return dist, world, rank, f"cuda:{local}"


def token_checksum(tokens: torch.Tensor) -> torch.Tensor:
    """Two position-weighted sums of the (+-1) tokens in float64: exact, so equal tokens <=> equal checksums in practice."""
    t = tokens.double()
    w = torch.arange(1, t.numel() + 1, device=t.device, dtype=torch.float64)
    return torch.stack([t.sum(), (t.flatten() * w).sum()])


def checksums_agree(dist, mine: torch.Tensor) -> bool:
    """Tensor parallel: the replicated state must be bit-identical on every rank -- compare the checksums of every image."""
    if dist.get_backend() == "gloo":
        mine = mine.cpu()
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    return all(torch.equal(v, allv[0]) for v in allv)


def tokens_agree(dist, tokens: torch.Tensor, dev) -> bool:
    return checksums_agree(dist, token_checksum(tokens))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # plain `python bench.py --gpus N`: become the launcher
        sys.exit(spawn_ranks(args))
    dist, world, rank, dev = setup_dist()
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    from bitdance_amd import synthetic as syn
    from bitdance_amd.build import build
    from bitdance_amd.dist_util import max_over_ranks, rank_seed
    build(verbose=False)
    tune = {}
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        tune[k] = int(v)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    size, H0, W0, metric = WORKLOADS[args.workload]
    if args.workload.startswith("imagenet-"):
        return bench_imagenet(args, dist, world, rank, dev, metric, barrier, max_over_ranks, rank_seed)
    if args.workload.startswith("ae-"):
        return bench_ae_decode(args, dist, world, rank, dev, metric, barrier, max_over_ranks)

    # ---------------------------------------------------------------- T2I workloads
    H, W = args.height or H0, args.width or W0
    if size == "tiny":
        H, W = min(H, 256), min(W, 256)
    num_images = args.num_images or 1
    n_sampling = args.sampling_steps or 50
    guidance = args.guidance if args.guidance is not None else 7.5
    tp_mode = world > 1 and args.parallel == "tp"
    comm = None
    t0 = time.perf_counter()
    if tp_mode:
        from bitdance_amd.tp import TPComm
        rows_max = 2 * num_images * 64
        # (+ the gather region of the column-split adaLN projection: 14 x 5120 output columns per evaluation row, one group of evaluations)
        from bitdance_amd.tp import ada_gather_bytes, seq_hbuf_bytes
        comm = TPComm.from_process_group(max(rows_max, 128) * 5120, device=dev, backend=args.tp_comm,
                                         gather_bytes=ada_gather_bytes(max(rows_max, 128), 14 * 5120),
                                         hbuf_bytes=seq_hbuf_bytes(max(rows_max, 128), 5120))   # operand landing buffer of the sequence-parallel row kernels
    pipe = syn.build_pipeline(size, dev, with_ae=True, tp=comm, weights=args.weights)
    if args.weights == "fp8":
        metric += " (fp8-e4m3 weights)"
    elif args.weights == "fp8a":
        metric += " (fp8-e4m3 weights + activations, fp8 MFMA)"
    pipe.tune = tune or None
    if tp_mode:
        ei = {}
        if args.tp_ada_split != "auto":
            ei["tp.ada_split"] = int(args.tp_ada_split)
        if args.tp_seq != "auto":
            ei["tp.seq"] = int(args.tp_seq)
            ei["tp.llm_seq"] = int(args.tp_seq)
        pipe.extra_ints = ei or None
    if args.attn_splits:
        pipe.attn_splits = args.attn_splits
    pipe.use_graph = not args.no_graph
    if rank == 0:
        print(f"[bench] model built in {time.perf_counter() - t0:.1f} s"
              + (f" (tensor parallel x{world}, exchange backend {comm.backend})" if tp_mode else ""), file=sys.stderr)
    P = pipe.parallel_num
    prompt = "A close-up portrait in a cinematic photography style, capturing a girl-next-door look on a sunny daytime urban street."
    kw = dict(cond_prompt=f"<|im_start|>user\n{prompt}<|im_end|>\n<|im_start|>assistant\n",
              uncond_prompt="<|im_start|>assistant\n", guidance_scale=guidance,
              num_sampling_steps=n_sampling, num_images=num_images,
              image_size=[H, W], max_length=(H // 16) * (W // 16))

    def one_pass(i):
        torch.manual_seed(rank_seed(1234, 0 if tp_mode else rank, i))       # tensor parallel: every rank draws the same noise
        with torch.amp.autocast("cuda", enabled=True, dtype=torch.bfloat16):
            return pipe.gen_image(**kw)

    def tp_pass_ok(i) -> bool:
        """One pass; True iff it completed on every rank AND every rank holds bit-identical tokens."""
        ok = 1
        try:
            if os.environ.get("BD_BENCH_FAIL_TP"):             # test hook: exercise the fall-back chain on a healthy node
                raise RuntimeError("BD_BENCH_FAIL_TP is set")
            one_pass(i)
        except Exception as e:                                 # in-kernel wait budget exceeded etc.: all ranks must agree on what happens next
            print(f"[bench] rank {rank}: tensor-parallel pass failed: {e}", file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            return False
        return tokens_agree(dist, next(iter(pipe._engines.values())).tok_all, dev)

    tp_note = None
    tp_fused_check = None
    # tensor parallel: the first image is always an untimed, verified one (every rank finished, bit-identical tokens; else the
    # fall-back chain below) -- also with --warmup 0, where it is the only untimed pass
    for i in range(max(args.warmup, 1) if tp_mode else args.warmup):
        if tp_mode and i == 0:
            if not tp_pass_ok(i):
                why = None
                if comm.backend != "rccl":
                    if rank == 0:
                        print("[bench] the IPC exchange failed or diverged on this node: falling back to RCCL all-reduce", file=sys.stderr, flush=True)
                    pipe._engines.clear()
                    ok = 1
                    try:
                        comm.reset()
                        comm.use_rccl()
                    except Exception as e:                     # e.g. ncclCommInitRank refusing the topology
                        print(f"[bench] rank {rank}: switching to the RCCL exchange failed: {e}", file=sys.stderr, flush=True)
                        ok = 0
                    flag = torch.tensor([ok], device="cpu" if dist.get_backend() == "gloo" else dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    if int(flag.item()) == 0 or not tp_pass_ok(i):
                        why = "tensor-parallel ranks failed or diverged with the hand-written exchange AND with the RCCL exchange"
                else:
                    why = "tensor-parallel ranks failed or diverged (RCCL exchange)"
                if why is not None:
                    # last resort, so that the node still gets a measured line: every rank runs the whole model on its own GPU over
                    # its own images (what the reference's evaluation scripts do) -- reported as replicas, weak scaling, with the reason
                    if rank == 0:
                        print(f"[bench] {why}: falling back to independent replicas", file=sys.stderr, flush=True)
                    tp_note = why
                    tp_mode = False
                    pipe._engines.clear()
                    del pipe
                    comm = None
                    torch.cuda.empty_cache()
                    pipe = syn.build_pipeline(size, dev, with_ae=True, tp=None, weights=args.weights)
                    pipe.tune = tune or None
                    if args.attn_splits:
                        pipe.attn_splits = args.attn_splits
                    pipe.use_graph = not args.no_graph
                    one_pass(i)
            if tp_mode and comm.backend != "rccl" and "tp_fuse" not in tune:
                # The construction-time self-test covers the exchange kernel; the push fused into the GEMM epilogues and the push
                # all-gather of the column-split adaLN projection first meet real links here.  Both are bit-identical to the
                # conservative forms by construction (tests/test_gpu_tp.py), so the same image is generated once more with
                # tune.tp_fuse = 0 / tp.ada_split = 0 and the tokens compared on every rank: equal -> the fused forms are timed;
                # different -> the conservative forms are, and the line says so.
                eng0 = next(iter(pipe._engines.values()))
                fused_sum, was_split = token_checksum(eng0.tok_all).clone(), bool(getattr(eng0, "ada_split", False))
                keep = (pipe.tune, getattr(pipe, "extra_ints", None))
                was_seq = bool(getattr(eng0, "seq_parallel", False))
                pipe.tune = dict(tune, tp_fuse=0)
                pipe.extra_ints = {"tp.ada_split": 0, "tp.seq": 0, "tp.llm_seq": 0}
                pipe._engines.clear()
                same = tp_pass_ok(i) and bool(torch.equal(token_checksum(next(iter(pipe._engines.values())).tok_all), fused_sum))
                flag = torch.tensor([1 if same else 0], device="cpu" if dist.get_backend() == "gloo" else dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                tp_fused_check = bool(int(flag.item()))
                if tp_fused_check:
                    pipe.tune, pipe.extra_ints = keep
                    pipe._engines.clear()
                    one_pass(i)                                # engines and graphs of the timed configuration are built untimed
                elif rank == 0:
                    print("[bench] fused reduce-scatter push / column-split adaLN / sequence-parallel row kernels disagree with the unfused "
                          f"all-reduce exchange on this node (adaLN split was {'on' if was_split else 'off'}, sequence-parallel "
                          f"{'on' if was_seq else 'off'}): timing the conservative forms", file=sys.stderr, flush=True)
        else:
            one_pass(i)
    barrier()
    sums = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        img = one_pass(args.warmup + i)
        if tp_mode:                                            # every image's tokens, not only the first warm-up's (a few tiny
            sums.append(token_checksum(next(iter(pipe._engines.values())).tok_all))   # device ops; compared after the clock stops)
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(img).all()
    if tp_mode:
        # every rank must hold bit-identical tokens for every timed image.  A divergence (an ordering problem of the hand-written
        # exchange that the warm-up comparison did not catch) makes the timing meaningless as a result -- but a crash would leave the node
        # without any line: it is reported in the line ("tp.ranks_bit_identical": false, "value" kept for diagnosis) and on stderr
        ranks_identical = bool(checksums_agree(dist, torch.stack(sums)))
        if not ranks_identical and rank == 0:
            print("[bench] TENSOR-PARALLEL RANKS DIVERGED in the timed region: the reported value is NOT a valid result "
                  "(re-run with --tp-seq 0 / --tune tp_fuse=0 / --tp-comm rccl to localise)", file=sys.stderr, flush=True)
    dt = max_over_ranks(dt, dist, "cpu" if (dist is not None and dist.get_backend() == "gloo") else dev)

    # One line carries both answers: the tensor-parallel `value` above AND what the same N GPUs deliver as independent
    # replicas over disjoint images -- what the reference's evaluation scripts do (eval/eval_dpg.py:25-29) and, for images/s, the better use
    # of a node.  Timed after the tensor-parallel pass on the same lease: every rank builds the unsharded model beside
    # its shard, one untimed + one timed image.
    replicas = None
    if tp_mode and not args.no_replicas:
        ok, why = 1, ""
        try:
            pipe_r = syn.build_pipeline(size, dev, with_ae=True, tp=None, weights=args.weights)
            pipe_r.tune = {k: v for k, v in tune.items() if not k.startswith("tp")} or None
            if args.attn_splits:
                pipe_r.attn_splits = args.attn_splits
            pipe_r.use_graph = not args.no_graph

            def pass_r(i):
                torch.manual_seed(rank_seed(1234, rank, 1000 + i))
                with torch.amp.autocast("cuda", enabled=True, dtype=torch.bfloat16):
                    return pipe_r.gen_image(**kw)
            pass_r(0)
            barrier()
            tr = time.perf_counter()
            img_r = pass_r(1)
            barrier()
            dtr = time.perf_counter() - tr
            assert torch.isfinite(img_r).all()
            del pipe_r, img_r
            torch.cuda.empty_cache()
        except Exception as e:                                 # (out of memory beside the shard, ...): the line says so
            ok, why, dtr = 0, str(e)[:200], 0.0
            barrier(); barrier()
        flag = torch.tensor([ok], device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        dtr = max_over_ranks(dtr, dist, "cpu" if dist.get_backend() == "gloo" else dev)
        replicas = ({"value": round(world * num_images / dtr, 5), "unit": "images/s", "n_gpus": world, "steps": 1, "warmup": 1,
                     "ms_per_step": round(dtr * 1e3, 2), "scaling": "weak", "parallelism": f"replicas x{world}"}
                    if int(flag.item()) == 1 else {"value": None, "error": why or "a rank failed"})

    if rank == 0:
        n = world
        images = (1 if tp_mode else n) * num_images * args.steps
        ar_steps = kw["max_length"] // P
        eng = next(iter(pipe._engines.values()))
        out = {
            "metric": metric, "value": round(images / dt, 5), "unit": "images/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong" if tp_mode else "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp8": "bf16 activations / fp32 accumulate, fp8-e4m3 weights (per-channel scales)",
                      "fp8a": "fp8-e4m3 weights (per-channel scales) and activations (per-row scales) on the fp8 MFMA for adaLN / qkv / w1 / "
                              "LLM qkv / gate-up, bf16 activations elsewhere, fp32 accumulate"}[args.weights],
            "data": "synthetic (random weights at true shapes, fixed token ids)",
            "config": {"workload": f"BitDance-{size.upper()} T2I {H}x{W}, {n_sampling} sampling steps, cfg {guidance}, "
                                   f"num_images={num_images} per {'job' if tp_mode else 'GPU'}" if size != "tiny" else "tiny",
                       "ar_steps": ar_steps, "rows_per_pass": eng.M,
                       "parallelism": (f"tp{n}" if tp_mode else f"replicas x{n}") if n > 1 else "single GPU",
                       "hipgraph": pipe.use_graph, **({"tensor_parallel_fallback": tp_note} if tp_note else {})},
            "phases_ms_last_step": {k: round(v, 1) for k, v in pipe.timings().items()},
        }
        if replicas is not None:
            out["replicas"] = replicas
        if tp_mode:
            wbytes = sum(t.numel() * t.element_size() for w_ in (pipe.head_w, pipe.llm_w, pipe.proj_w) for t in w_.ptrs.values())
            nblk, L = pipe.head_w.nblocks, pipe.llm_w.cfg["num_hidden_layers"]
            n_x = ar_steps * (n_sampling + 1) * 2 * nblk + (ar_steps - 1) * 2 * L
            info = comm.info()
            out["tp"] = {"size": n, "world_size_seen": dist.get_world_size(), "process_group_backend": dist.get_backend(),
                         "exchange_backend": comm.backend, "exchange_fences": getattr(comm, "fences", 0),
                         "reduce_scatter_push": ("in the exchange kernel" if (tune.get("tp_fuse", 1) == 0 or tp_fused_check is False or comm.backend == "rccl")
                                                 else "fused into the row-split GEMM epilogue (tune.tp_fuse)"),
                         "fused_forms_equal_unfused_on_this_node": tp_fused_check,
                         "exchange_buffer_uncached": info["data_uncached"],
                         "flag_block_uncached": info["flags_uncached"], "fallback_reason": getattr(comm, "fallback_reason", None),
                         "images_checked_bit_identical": args.steps + 1,
                         "weight_bytes_per_rank": int(wbytes),
                         "adaln_projection": "column-split + push all-gather" if getattr(eng, "ada_split", False) else "replicated",
                         "head_row_kernels": ("sequence-parallel (csrc/bd_sp.hip: rows / tp rows per rank, no exchange kernel in an evaluation)"
                                              if getattr(eng, "seq_parallel", False) else "replicated behind an all-reduce kernel per row-split Linear"),
                         "llm_row_kernels": ("sequence-parallel (csrc/bd_sp.hip rms_sp: rows / tp rows of the residual stream per rank, no exchange kernel in a decode step)"
                                             if getattr(eng, "llm_seq_parallel", False) else "replicated behind an all-reduce kernel per row-split Linear"),
                         "exchanges_per_image": n_x, "exchange_payload_bytes_per_rank": int(eng.M * 5120 * 6 * (n - 1) / n),
                         "ranks_bit_identical": ranks_identical}
        if not args.no_roofline:
            st = pipe._stream
            with torch.cuda.stream(st):
                out["roofline"] = gemm_roofline(eng, lambda: (eng.head_sample(), eng.projector(), eng.llm_step()), eng.M)
            if tp_mode:
                out["roofline"]["note"] = "rank 0's launches: per-rank slices of the weights"
        if world == 1 and size == "14b-64x" and num_images == 1 and (H, W) == (1024, 1024):
            # "Report B=1 ... and B=4 (num_images=4, what the eval scripts do, eval/eval_dpg.py:44)".  `value` above
            # stays the B = 1 headline; each extra point is ONE untimed warm-up + ONE timed gen_image of that many images on the same
            # pipeline (the previous engine is dropped, one for the new row count built), then its GEMM launches profiled in situ
            # (MFMA-bound).  "b4" = the eval scripts' batch; "throughput" = the batch where this code base's images/s saturate on one
            # GPU (round 6: 0.505 / 0.529 images/s at 8 / 16 images, profiles/r06_bench_b8_first.json, _b16_first.json -- 8 keeps the
            # default run inside a few minutes).
            def extra_point(nimg: int, seed: int) -> dict:
                kwn = dict(kw, num_images=nimg)

                def passn(i):
                    torch.manual_seed(rank_seed(seed, rank, i))
                    with torch.amp.autocast("cuda", enabled=True, dtype=torch.bfloat16):
                        return pipe.gen_image(**kwn)
                passn(0)
                torch.cuda.synchronize()
                tn = time.perf_counter()
                imgn = passn(1)
                torch.cuda.synchronize()
                dtn = time.perf_counter() - tn
                assert torch.isfinite(imgn).all() and imgn.shape[0] == nimg
                engn = next(iter(pipe._engines.values()))
                pt = {"value": round(nimg / dtn, 5), "unit": "images/s", "num_images": nimg, "steps": 1, "warmup": 1, "ms_per_step": round(dtn * 1e3, 2),
                      "rows_per_pass": engn.M, "phases_ms": {k: round(v, 1) for k, v in pipe.timings().items()}}
                if not args.no_roofline:
                    with torch.cuda.stream(pipe._stream):
                        rn = gemm_roofline(engn, lambda: (engn.head_sample(), engn.projector(), engn.llm_step()), engn.M)
                    pt["roofline"] = {k: rn.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "mfma_busy", "eff_clock_ghz", "pmc_source",
                                                              "flop_per_launch", "launches", "avg_launch_us")}
                    pt["roofline"]["per_gemm"] = [{k: q[k] for k in ("name", "launches", "avg_us", "TFLOPs", "frac_of_mfma_peak", "mfma_busy", "eff_clock_ghz",
                                                                      "splitk", "nwaves") if k in q} for q in rn["per_gemm"]]
                del engn, imgn
                pipe._engines.clear()
                torch.cuda.empty_cache()
                return pt
            if not args.no_b4:
                out["b4"] = extra_point(4, 4321)
            if not args.no_throughput:
                out["throughput"] = extra_point(args.throughput_images, 8765)
                out["throughput"]["note"] = ("images/s of ONE gen_image call at the batch where one GPU saturates (MFMA-bound: every head GEMM at >= 1024 rows); "
                                             "`value` above stays the num_images = 1 headline")
        if world == 1 and not args.no_cpu_baseline:
            del pipe
            out["cpu_baseline"] = cpu_baseline_t2i(args, P, ar_steps, n_sampling + 1, px=int((H * W) ** 0.5))
        print(json.dumps(out), flush=True)
    elif tp_mode and not args.no_roofline:
        # rank 0's eager profiling pass launches exchanges: every rank has to take part in them
        eng = next(iter(pipe._engines.values()))
        with torch.cuda.stream(pipe._stream):
            eng.head_sample(); eng.projector(); eng.llm_step()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_imagenet(args, dist, world, rank, dev, metric, barrier, max_over_ranks, rank_seed):
    """BASELINE config 2: one ``BitDance.sample`` call over ``--num-images`` (default 384) classes, 100 sampling steps,
    linear CFG 6.1, VAE decode in chunks; N > 1 = replicas over disjoint class batches (what sample_ddp_parallel.py does)."""
    from bitdance_amd import synthetic as syn
    variant = args.workload.split("-", 1)[1]
    n_cls = args.num_images or IMAGENET_DEFAULTS[variant][0]
    n_sampling = args.sampling_steps or 100
    guidance = args.guidance if args.guidance is not None else IMAGENET_DEFAULTS[variant][1]
    mcfg = syn.IMAGENET_MODELS[variant]
    P = mcfg["parallel_num"]
    m = syn.build_imagenet(dev, cfg=mcfg, with_vae=not args.no_decode)
    tune = {k: int(v) for k, v in (kv.split("=") for kv in filter(None, args.tune.split(",")))}
    m.tune = tune or None
    ids = (torch.arange(n_cls) + rank * n_cls) % 1000

    def one_pass(i):
        torch.manual_seed(rank_seed(99, rank, i))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return m.sample(ids, n_sampling, cfg_scale=guidance, cfg_schedule="linear", chunk_size=48)

    for i in range(args.warmup):
        one_pass(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out_img = one_pass(args.warmup + i)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dist, "cpu" if (dist is not None and dist.get_backend() == "gloo") else dev)
    assert torch.isfinite(out_img.float()).all()
    if rank == 0:
        images = world * n_cls * args.steps
        ar_steps = (256 // 16) ** 2 // P
        out = {"metric": metric, "value": round(images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random weights at true shapes, class ids arange % 1000)",
               "config": {"workload": f"imagenet_gen BitDance-{variant[0].upper()}-{variant[1:]} 256x256, batch {n_cls} classes per call, {n_sampling} sampling steps, "
                                      f"linear CFG {guidance}, VAE decode {'off' if args.no_decode else 'on (chunks of 48)'}",
                          "ar_steps": ar_steps, "rows_per_pass": 2 * n_cls * P,
                          "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                          "hipgraph": bool(m.combined_engine and m.use_graph)}}
        if not args.no_roofline:
            if (n_cls, 2) in m._comb:                            # the combined engine of AR steps 1..: projector + transformer + head
                eng = m._comb[(n_cls, 2)]
                # back to the state right after the first step: the KV cache is full after a whole sample, one more decode step
                # would append past its end
                eng.reset([mcfg["cls_token_num"] + P - 1] * min(2 * n_cls, 16))
                out["roofline"] = gemm_roofline(eng, lambda: (eng.projector(), eng.llm_step(), eng.head_sample()), eng.M)
            else:
                eng = m._eng[(n_cls, 2)]
                out["roofline"] = gemm_roofline(eng, eng.head_sample, eng.M)
        if world == 1 and not args.no_cpu_baseline and variant == "b16x":
            out["cpu_baseline"] = cpu_baseline_imagenet(n_sampling + 1, ar_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_ae_decode(args, dist, world, rank, dev, metric, barrier, max_over_ranks):
    """Standalone conv decoder (VQModel.decode, modeling/vision_encoder/autoencoder.py:129-196,514-516): a step = one decode of
    ``--num-images`` (default 1) random +-1 latents to ``--height`` x ``--width`` (default 1024 x 1024) on the native kernels
    (csrc/bd_conv.hip); N > 1 = replicas.  roofline = the convolution kernel (MFMA-bound): algorithmic FLOPs of every
    convolution of the decode / the HIP-event time of the whole decode (GroupNorm passes included: a lower bound for the kernel)."""
    from bitdance_amd import synthetic as syn
    from bitdance_amd.autoencoder import VQModel
    cfg = syn.AE_D32C256 if args.workload.startswith("ae-d32c256") else syn.AE_D16C32
    patch = 2 ** (len(cfg["ddconfig"]["ch_mult"]) - 1)
    H, W = args.height or 1024, args.width or 1024
    n_img = args.num_images or 1
    ae = VQModel(**cfg).eval()
    ae.load_state_dict(syn.random_ae_state(cfg, dev), strict=True, assign=True)
    ae.to(dev)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    z = torch.sign(torch.randn(n_img, cfg["ddconfig"]["z_channels"], H // patch, W // patch, device=dev, generator=g))

    def one_pass():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return ae.decode(z)

    for _ in range(max(1, args.warmup)):
        out = one_pass()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        out = one_pass()
    e1.record()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dist, "cpu" if (dist is not None and dist.get_backend() == "gloo") else dev)
    assert torch.isfinite(out.float()).all() and out.shape == (n_img, 3, H, W)
    if rank == 0:
        # algorithmic FLOPs of the decoder's convolutions (2 * pixels * taps * Cin * Cout each), from the module itself
        flops = 0
        dec = ae.decoder
        hh, ww = H // patch, W // patch
        def conv_fl(c, h_, w_):
            return 2.0 * h_ * w_ * c.kernel_size[0] * c.kernel_size[1] * c.in_channels * c.out_channels
        flops += conv_fl(dec.conv_in, hh, ww)
        for b in dec.mid_block:
            flops += conv_fl(b.conv1, hh, ww) + conv_fl(b.conv2, hh, ww)
        for lv in reversed(range(dec.nlev)):
            for b in dec.up[lv].block:
                flops += conv_fl(b.conv1, hh, ww) + conv_fl(b.conv2, hh, ww) + (conv_fl(b.nin_shortcut, hh, ww) if b.cin != b.cout else 0)
            if lv > 0:
                flops += conv_fl(dec.up[lv].upsample.conv1, hh, ww)
                hh, ww = 2 * hh, 2 * ww
        flops += conv_fl(dec.conv_out, hh, ww)
        flops *= n_img
        ms = e0.elapsed_time(e1) / args.steps
        ach = flops / (ms * 1e-3) / 1e12
        out_j = {"metric": metric, "value": round(world * n_img * args.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                 "warmup": max(1, args.warmup), "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                 "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random weights at the decoder's shapes, random +-1 latent)",
                 "config": {"workload": f"{args.workload}: VQModel.decode of {n_img} x [{cfg['ddconfig']['z_channels']},{H // patch},{W // patch}] -> {H}x{W}, "
                                        f"ch_mult {cfg['ddconfig']['ch_mult']}, native gfx950 kernels",
                            "parallelism": f"replicas x{world}" if world > 1 else "single GPU"},
                 "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFS, 4),
                              "traffic": None, "kernel": "conv_tile_kernel (bd_conv.hip): convolution FLOPs of one decode / HIP-event time of the whole "
                              "decode (GroupNorm passes included)", "flop_per_decode": int(flops), "decode_ms": round(ms, 2)}}
        print(json.dumps(out_j), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
