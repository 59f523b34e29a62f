"""A/B sweep of the launch configurations of the Qwen3-14B decode step's GEMMs (128 rows = one image with CFG), all in ONE process
on ONE box: a 12-layer slice of the model at true dimensions (7.9 GB of packed weights: every layer's weights come from HBM),
caches ~1k tokens long, the step run eagerly and timed with events on the launch stream, plus the in-situ time of each GEMM.

python tools/llm_sweep.py [reps] ["k=v,k=v;k=v,..." configs] [cached tokens]      keys: llm.<gemm>.{S,nw,kw,ring}, slab3, kparts8, ...,
                                                                                    splits (flash-decode splits of the attention)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from bitdance_amd import synthetic as syn                   # noqa: E402

DEFAULT = [
    {},
    {"llm.down.nw": 4, "llm.down.kw": 2, "llm.down.S": 3},
    {"llm.down.nw": 8, "llm.down.kw": 2, "llm.down.S": 6},
    {"llm.down.nw": 4, "llm.down.kw": 1, "llm.down.S": 6, "llm.down.ring": 3},
    {"llm.gu.nw": 9, "llm.gu.S": 2},
    {"llm.gu.nw": 4, "llm.gu.S": 1, "llm.gu.ring": 3},
    {"llm.gu.nw": 8, "llm.gu.S": 1, "llm.gu.ring": 3},
    {"llm.qkv.nw": 4, "llm.qkv.kw": 1, "llm.qkv.S": 4, "llm.qkv.ring": 3},
    {"llm.qkv.nw": 8, "llm.qkv.kw": 1, "llm.qkv.S": 8},
    {},
]
LAYERS = 12


def main():
    with torch.cuda.stream(torch.cuda.Stream()):
        run()


def run():
    shard = None
    if "--tp-shard" in sys.argv:                             # ONE rank's critical path at the tensor-parallel shard, in loop-back (tools/head_sweep.py)
        i = sys.argv.index("--tp-shard")
        shard = tuple(int(v) for v in sys.argv[i + 1].split("/"))
        del sys.argv[i:i + 2]
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfgs = list(DEFAULT)
    if len(sys.argv) > 2:
        cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in sys.argv[2].split(";")]
    dev = "cuda"
    cfg = dict(syn.QWEN3_14B)
    cfg["num_hidden_layers"] = LAYERS
    cfg["vocab_size"] = 1024
    if shard:
        lw = E.LlmWeights.from_state_dict(syn.random_llm_state(cfg, dev), cfg, dev, keep_for_prefill=False, tp_rank=shard[0], tp_size=shard[1])
        print(f"# rank {shard[0]} of {shard[1]} in loop-back: this rank's weight slices, peers' buffers are scratch, flags written locally "
              f"(values are meaningless: the peers contribute zeros)", flush=True)
        if len(sys.argv) <= 2:
            cfgs = [{"tp.llm_seq": 0}, {"tp.llm_seq": 1}, {"tp.llm_seq": 0}, {"tp.llm_seq": 1}]
    else:
        lw = E.LlmWeights.from_state_dict(syn.random_llm_state(cfg, dev), cfg, dev, keep_for_prefill=False)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(128, cfg["hidden_size"], device=dev, generator=g)
    ref = None
    past = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    for tune in cfgs:
        tune = dict(tune)
        splits = tune.pop("splits", 8)
        extra = {k: tune.pop(k) for k in list(tune) if k.startswith("tp.")}
        comm = None
        if shard:
            from bitdance_amd.tp import TPComm, seq_hbuf_bytes
            comm = TPComm.loopback_rank(shard[0], shard[1], 128 * 5120, dev, hbuf_bytes=seq_hbuf_bytes(128, 5120))
            comm.set_timeout(5.0)
        eng = E.Engine(None, None, lw, num_images=1, branches=2, device=dev, max_tokens=64, max_kv=past + 256, attn_splits=splits,
                       tune=tune, comm=comm, extra_ints=extra or None)
        tune["splits"] = splits
        tune.update(extra)
        if shard:
            tune["llm_seq"] = int(eng.llm_seq_parallel)
        eng.set_int("rt.emit_cond", 0)                      # no head in this context
        st = torch.cuda.current_stream()

        def step():
            eng.reset([past, past - 5])
            eng.residual()[:128].copy_(x)
            eng.llm_step()

        step()
        torch.cuda.synchronize()
        hid = eng.hidden().float().clone()
        if ref is None:
            ref = hid
        err = float((hid - ref).abs().max())
        ts = []
        for _ in range(reps):
            eng.reset([past, past - 5])
            eng.residual()[:128].copy_(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            eng.llm_step()
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        prof = eng.profile_gemms(step)
        per = "  ".join(f"{k.split('.')[1]} {v['ms'] / v['count'] * 1e3:6.1f}" for k, v in sorted(prof.items()))
        cf = "  ".join(f"{n}:S{eng.gemm_config('llm.' + n)[0]}w{eng.gemm_config('llm.' + n)[1]}" for n in ("qkv", "o", "gu", "down"))
        print(f"{str(tune):75s} {min(ts) / LAYERS * 1e3:7.1f} us/layer (median {sorted(ts)[len(ts) // 2] / LAYERS * 1e3:7.1f})  GEMM us: {per}  "
              f"[{cf}]  max|d hidden| vs first {err:.3g}", flush=True)
        if comm is not None:
            comm.check()
        del eng, comm
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
