"""A/B sweep of the structural options of the head sampling chain at true 14B dimensions (one AR step = N + 1 evaluations of the
6-block head, M = 128 rows), all in ONE process on ONE box (box-to-box spread is +-5 %):

  tune.ada_group  evaluations whose adaLN projections run as one GEMM (bd_api.hip head_ada_group)
  tune.<gemm>.{S,nw,kw,ring}, tune.slab3, ...   launch configurations (bd_api.hip choose_cfg)
(the run-ahead weight prefetch this tool also swept in round 3 -- profiles/r03_head_sweep1.log -- measured negative and is gone)

Every configuration is timed as a hipGraph replay and its sampled latent is compared bit for bit with the first one's.
python tools/head_sweep.py [reps] [n_steps] ["k=v,k=v;k=v,..." extra configs] [bf16|fp8|fp8a]

  --tp-shard r/N [--loopback]   ONE rank's critical path at the tensor-parallel shard shapes on ONE GPU: rank r of N packs its weight
      slices, the communicator runs in loop-back (bd_comm_set_loopback: the peers' buffers are scratch copies, every flag a peer would
      write is written locally), so the captured graph holds exactly the launches, pushes and waits of that rank on a node -- minus the
      links.  Configs may carry tp.seq=0/1 (all-reduce form / sequence-parallel row kernels), tp.ada_split=0/1, tune.* keys.  The
      sampled latent is meaningless there (the peers contribute zeros) and is not compared."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from oracle import tiny_models as tm                        # noqa: E402  (shape table + seeded random weights only)
from oracle.true_dims import device_seeded_state            # noqa: E402

DEFAULT = [
    dict(ada_group=1),
    dict(ada_group=4),
    dict(ada_group=8),
    dict(ada_group=16),
    dict(ada_group=26),
    dict(ada_group=52),
    dict(ada_group=1),
]


def main():
    with torch.cuda.stream(torch.cuda.Stream()):
        run()


def run():
    shard = None
    if "--tp-shard" in sys.argv:
        i = sys.argv.index("--tp-shard")
        shard = tuple(int(v) for v in sys.argv[i + 1].split("/"))
        del sys.argv[i:i + 2]
    if "--loopback" in sys.argv:
        sys.argv.remove("--loopback")
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    cfgs = list(DEFAULT)
    if len(sys.argv) > 3:
        cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in sys.argv[3].split(";")]
    elif shard:
        cfgs = [{"tp.seq": 0}, {"tp.seq": 1}, {"tp.seq": 1, "sp_wait": 0}, {"tp.seq": 0}]
    dev = "cuda"
    B = int(os.environ.get("BD_SWEEP_B", "1"))             # images per pass (rows = 128 B): 4 = the eval scripts' batch
    cfgd = dict(ch_target=32, ch_cond=5120, ch_latent=5120, depth_latent=6, depth_adanln=2)
    sd = device_seeded_state(tm.head_shapes(cfgd), 101, dev)
    wmode = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    if shard:
        hw = E.HeadWeights.from_state_dict(sd, dev, weights=wmode, tp_rank=shard[0], tp_size=shard[1])
        print(f"# rank {shard[0]} of {shard[1]} in loop-back: weight slices of this rank, peers' buffers are scratch, flags written locally", flush=True)
    else:
        hw = E.HeadWeights.from_state_dict(sd, dev, weights=wmode)
    del sd
    g = torch.Generator(device=dev).manual_seed(7)
    cond = torch.randn(2 * B, 64, 5120, device=dev, generator=g)
    noise = torch.randn(1, n + 1, B, 64, 32, device=dev, generator=g)
    ref = None
    from bitdance_amd._lib import check, lib
    for tune in cfgs:
        tune = dict(tune)
        opts = {k: tune.pop(k) for k in list(tune) if k.startswith(("wide.", "red.", "rows.", "tile.", "half.")) or k == "tile"}      # process-wide GEMM options
        for k, v in opts.items():
            check(lib().bd_set_gemm_option(k.encode(), v))
        extra = {k: tune.pop(k) for k in list(tune) if k.startswith("tp.")}                         # context keys outside tune.*
        tune_shown = dict(tune, **opts, **extra)
        comm = None
        if shard:
            from bitdance_amd.tp import TPComm, ada_gather_bytes, seq_hbuf_bytes
            comm = TPComm.loopback_rank(shard[0], shard[1], 128 * 5120, dev, gather_bytes=ada_gather_bytes(128, 14 * 5120),
                                        hbuf_bytes=seq_hbuf_bytes(128, 5120))
            comm.set_timeout(5.0)
        eng = E.Engine(hw, None, None, num_images=B, branches=2, device=dev, max_tokens=64, parallel_num=64, tune=tune, comm=comm,
                       extra_ints=extra or None)
        if shard:
            tune_shown = dict(tune_shown, seq=int(eng.seq_parallel), ada_split=int(eng.ada_split))
            if ref is None:
                print("# launch configurations at this shard (split-K, waves + 16 ring + 256 (kparts - 1)): "
                      + ", ".join(f"{n} {eng.gemm_config('head.' + n)}" for n in ("qkv", "wo", "w1", "w2", "ada")), flush=True)
                ref = 0
        eng.set_schedule(n, 7.5, 1)
        eng.load_noise(noise)
        eng.reset([0] * (2 * B))
        eng.set_cond(cond)
        eng.head_sample()
        torch.cuda.synchronize()
        pred = eng.pred().clone()
        if ref is None:
            ref = pred
        same = bool(torch.equal(pred, ref)) if not shard else None
        eng.capture(0)
        eng.reset([0] * (2 * B))
        eng.launch(0)
        torch.cuda.synchronize()
        same = (same and bool(torch.equal(eng.pred(), ref))) if not shard else None
        if comm is not None:
            comm.check()
        ts = []
        for _ in range(reps):
            eng.reset([0] * (2 * B))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.launch(0)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        print(f"{str(tune_shown):60s} graph {dt * 1e3:8.2f} ms/AR step  {dt / (n + 1) * 1e6:8.1f} us/eval  (median {sorted(ts)[len(ts) // 2] * 1e3:.2f})  "
              f"bit-identical to first: {same}", flush=True)
        if comm is not None:
            comm.check()
        del eng, comm
        for k in opts:
            check(lib().bd_set_gemm_option(k.encode(), {"wide.ring": 2, "tile": 1, "red.first": 1, "rows.ln_occ": 5, "rows.swiglu_t": 512, "tile.minrb": 32, "half.form": 1}.get(k, -1)))
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
