#!/usr/bin/env python
"""Benchmark: images/sec at 1024 px, BitDance-14B-64x shapes, on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one whole pass of the hot path: one ``gen_image`` call (prefill, 64 AR steps x [51 diffusion-head
evaluations + sign + projector + cond/uncond LLM forward], AE decode) for ``--num-images`` images on synthetic
(random-weight, true-shape) models with inputs resident in HBM.  N > 1 runs one replica per GPU over disjoint
images (what the reference's own multi-GPU evaluation does, eval/eval_dpg.py:25-29): weak scaling, no data-path
collective; value = images of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline"     : achieved HBM GB/s of the dominant kernel family (the weight-streaming GEMM), measured live
                   with HIP events on the pipeline's stream, vs the 8 TB/s HBM3E peak
  "cpu_baseline" : the CPU oracle (a port of the reference algorithm) timed on this box's host cores on a bounded
                   sample (one head evaluation + one LLM layer step at true shapes), extrapolated to one image
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="14b-64x", choices=["14b-64x", "tiny"])
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--num-images", type=int, default=1)
    ap.add_argument("--sampling-steps", type=int, default=50)
    ap.add_argument("--guidance", type=float, default=7.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tune", default="", help="comma list name.S=4,name.nw=2 overriding GEMM launch configs")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
def gemm_roofline(pipe) -> dict:
    """In-situ roofline of the dominant kernel family (the weight-streaming GEMM, bd_gemm.hip): one extra AR step is
    run eagerly on the pipeline's stream with every GEMM launch bracketed by HIP events (bd_prof_*), so weights are
    NOT cache-resident between launches.  achieved = algorithmic weight bytes (N*K*2 per launch) / event time."""
    eng = next(iter(pipe._engines.values()))
    st = pipe._stream
    with torch.cuda.stream(st):
        prof = eng.profile_gemms(lambda: (eng.head_sample(), eng.projector(), eng.llm_step()))
    tot_b = sum(r["bytes"] for r in prof.values())
    tot_ms = sum(r["ms"] for r in prof.values())
    n_launch = sum(r["count"] for r in prof.values())
    ach = tot_b / tot_ms / 1e6
    per = []
    for name, r in sorted(prof.items()):
        S, nw = eng.gemm_config(name)
        per.append({"name": name, "launches": r["count"], "avg_us": round(r["ms"] / r["count"] * 1e3, 2),
                    "GBs": round(r["bytes"] / r["ms"] / 1e6, 1), "splitk": S, "nwaves": nw & 15, "ring": nw >> 4})
    # HBM bytes per launch from the PMC pass (rocprofv3 --pmc FETCH_SIZE in its own run, x2 gfx950 correction:
    # tools/pmc_gemm_traffic.py -> profiles/r01_pmc_gemm_traffic.json), weighted by this step's launch mix; only
    # valid for the shapes / launch configs that pass measured, else null
    traffic, traffic_src = None, None
    pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_gemm_traffic.json")
    if os.path.exists(pj) and eng.M == 128:
        pm = json.load(open(pj))["gemms"]
        cfgs = {q["name"]: q for q in per}
        if all(n in pm and pm[n]["splitk"] == cfgs[n]["splitk"] and pm[n]["nwaves"] == cfgs[n]["nwaves"] and
               pm[n]["N"] * pm[n]["K"] * 2 * r["count"] == int(r["bytes"]) for n, r in prof.items()):
            traffic = int(sum(pm[n]["hbm_read_bytes"] * r["count"] for n, r in prof.items()) / n_launch)
            traffic_src = "profiles/r01_pmc_gemm_traffic.json (FETCH_SIZE, read bytes per launch, launch-mix weighted)"
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "gemm_kernel<NW,MB,EPI> (bd_gemm.hip): every weight-streaming GEMM launch of one AR step, in situ",
            "launches": n_launch, "bytes_per_launch": int(tot_b / n_launch),
            "avg_launch_us": round(tot_ms / n_launch * 1e3, 2), "per_gemm": per}


# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(args) -> dict:
    """The CPU oracle (port of the reference algorithm, oracle/) on the host cores, bounded sample (~20 s): ONE of the
    head's 6 transformer blocks + the stacked adaLN Linear (M = 128 rows) and ONE LLM decoder layer over a 2 x 64-token
    block with 1k cached tokens, all at true 14B shapes; extrapolated to one image:
    T = AR * (N+1) * (nblocks * t_block + t_ada) + (AR-1) * L * t_layer   (prefill, final layer, AE decode excluded)."""
    from oracle import diff_head, qwen3
    from oracle.numerics import Policy
    from bitdance_amd import synthetic as syn
    torch.set_num_threads(os.cpu_count() or 1)
    big = args.size == "14b-64x"
    hc, lc = (syn.HEAD_14B_64X, syn.QWEN3_14B) if big else (syn.TINY_HEAD, syn.TINY_LLM)
    pol = Policy("autocast")
    D = hc["ch_latent"]
    H = int(D * 1.5)
    bf = torch.bfloat16
    cw = lambda *s: torch.full(s, 0.01, dtype=bf)
    w = {}
    p = "net.res_blocks.0."
    for nn_ in ("norm1", "norm2"):
        w[p + nn_ + ".weight"], w[p + nn_ + ".bias"] = torch.ones(D), torch.zeros(D)
    for nme, n, k in [("attn.wqkv", 3 * D, D), ("attn.wo", D, D), ("w1", 2 * H, D), ("w2", D, H)]:
        w[p + nme + ".weight"], w[p + nme + ".bias"] = cw(n, k), cw(n)
    nada = hc["depth_adanln"] * 6 * D + 2 * D
    w_ada, b_ada = cw(nada, D), cw(nada)
    x = torch.randn(2, 64, D).to(bf)
    y = torch.randn(2, 64, D).to(bf)
    mods = [torch.randn(2, 64, D).to(bf) * 0.1 for _ in range(6)]
    with torch.no_grad():
        t0 = time.perf_counter()
        diff_head.trans_block(w, 0, x, mods, D // 128, pol)
        t_block = time.perf_counter() - t0
        t0 = time.perf_counter()
        pol.linear(y, w_ada, b_ada)
        t_ada = time.perf_counter() - t0
    del w, w_ada
    one = dict(lc, num_hidden_layers=1)
    Dl, nh, nkv, hd, ff = lc["hidden_size"], lc["num_attention_heads"], lc["num_key_value_heads"], lc["head_dim"], lc["intermediate_size"]
    lw = {"model.norm.weight": torch.ones(Dl, dtype=bf)}
    p = "model.layers.0."
    for nme, n, k in [("self_attn.q_proj", nh * hd, Dl), ("self_attn.k_proj", nkv * hd, Dl), ("self_attn.v_proj", nkv * hd, Dl),
                      ("self_attn.o_proj", Dl, nh * hd), ("mlp.gate_proj", ff, Dl), ("mlp.up_proj", ff, Dl), ("mlp.down_proj", Dl, ff)]:
        lw[p + nme + ".weight"] = cw(n, k)
    for nme, n in [("self_attn.q_norm", hd), ("self_attn.k_norm", hd), ("input_layernorm", Dl), ("post_attention_layernorm", Dl)]:
        lw[p + nme + ".weight"] = torch.ones(n, dtype=bf)
    past = 1024 if big else 128
    cache = [[torch.randn(2, nkv, past, hd).to(bf), torch.randn(2, nkv, past, hd).to(bf)]]
    xin = torch.randn(2, 64, Dl)
    ones = torch.ones(2, 1, 64, past + 64, dtype=torch.bool)
    t0 = time.perf_counter()
    with torch.no_grad():
        qwen3.model_forward(lw, one, xin, cache, ones, pol)
    t_layer = time.perf_counter() - t0
    ar = (args.height // 16) * (args.width // 16) // 64
    n_ev = args.sampling_steps + 1
    L = lc["num_hidden_layers"]
    t_head = hc["depth_latent"] * t_block + t_ada
    t_img = ar * n_ev * t_head + (ar - 1) * L * t_layer
    return {"value": round(1.0 / t_img, 8), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (CPU port), {'true 14B' if big else 'tiny'} shapes, M=128 rows: 1 head block ({t_block:.2f} s) + adaLN Linear "
                      f"({t_ada:.2f} s) + 1 LLM layer step 2x64 tokens / {past} cached ({t_layer:.2f} s); extrapolated "
                      f"x{ar * n_ev} evals, x{(ar - 1) * L} layer steps; excludes prefill, final layer, AE decode"}


# ---------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = f"cuda:{local if world > 1 else 0}"

    from bitdance_amd import synthetic as syn
    from bitdance_amd.build import build
    build(verbose=False)
    t0 = time.perf_counter()
    pipe = syn.build_pipeline(args.size, dev, with_ae=True)
    tune = {}
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        tune[k] = int(v)
    pipe.tune = tune or None
    pipe.use_graph = not args.no_graph
    if rank == 0:
        print(f"[bench] model built in {time.perf_counter() - t0:.1f} s", file=sys.stderr)
    if args.size == "tiny":
        args.height, args.width = min(args.height, 256), min(args.width, 256)
    prompt = "A close-up portrait in a cinematic photography style, capturing a girl-next-door look on a sunny daytime urban street."
    kw = dict(cond_prompt=f"<|im_start|>user\n{prompt}<|im_end|>\n<|im_start|>assistant\n",
              uncond_prompt="<|im_start|>assistant\n", guidance_scale=args.guidance,
              num_sampling_steps=args.sampling_steps, num_images=args.num_images,
              image_size=[args.height, args.width], max_length=(args.height // 16) * (args.width // 16))

    from bitdance_amd.dist_util import job_throughput, max_over_ranks, rank_seed

    def one_pass(i):
        torch.manual_seed(rank_seed(1234, rank, i))
        with torch.amp.autocast("cuda", enabled=True, dtype=torch.bfloat16):
            img = pipe.gen_image(**kw)
        return img

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_pass(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        img = one_pass(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(img).all()
    dt = max_over_ranks(dt, dist, dev)

    out = None
    if rank == 0:
        n = world
        images = n * args.num_images * args.steps
        assert abs(images / dt - job_throughput(args.num_images, args.steps, dt, n)) < 1e-9
        out = {
            "metric": "images/sec @1024px BitDance-14B-64x" if args.size == "14b-64x" else "images/sec (tiny smoke config)",
            "value": round(images / dt, 5), "unit": "images/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random weights at true shapes, fixed token ids)",
            "config": {"workload": f"BitDance-14B-64x T2I {args.height}x{args.width}, {args.sampling_steps} sampling steps, "
                                   f"cfg {args.guidance}, num_images={args.num_images} per GPU"
                                   if args.size == "14b-64x" else "tiny",
                       "ar_steps": kw["max_length"] // 64, "parallelism": f"replicas x{n}" if n > 1 else "single GPU",
                       "hipgraph": pipe.use_graph},
            "phases_ms_last_step": {k: round(v, 1) for k, v in pipe.timings().items()},
        }
        if not args.no_roofline:
            out["roofline"] = gemm_roofline(pipe)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
