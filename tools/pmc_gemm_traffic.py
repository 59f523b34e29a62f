"""HBM traffic of every weight-streaming GEMM shape of one AR step, from the FETCH_SIZE PMC counter.

  1. on the GPU box, counters in their own pass (MI355X_MICROARCH.md, HBM section):
       rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o p -- python tools/pmc_gemm_traffic.py run
  2. python tools/pmc_gemm_traffic.py parse /tmp/pmc/p_results.db profiles/r01_pmc_gemm_traffic.json

`run` launches each (shape, split-K, waves) exactly as the engine configures it at M = 128, REPS times in a row;
`parse` takes the gemm_kernel dispatches in order, REPS per shape, and writes per-shape bytes per launch:
FETCH_SIZE is reported in KiB... of 64 B requests tallied for 128 B on gfx950 wide streaming reads, so the guide's
correction (x2) is applied and stated in the output.
"""
import json
import os
import sqlite3
import sys

REPS = 4
M = 128
# name, N, K, split-K, waves: the in-situ configuration of bench.py's default workload (roofline.per_gemm)
SHAPES = [("head.ada", 71680, 5120, 1, 10), ("head.qkv", 15360, 5120, 2, 4), ("head.wo", 5120, 5120, 6, 4),
          ("head.w1", 15360, 5120, 2, 4), ("head.w2", 5120, 7680, 6, 4), ("head.cond", 5120, 5120, 6, 4),
          ("proj.fc2", 5120, 5120, 6, 4), ("llm.qkv", 7168, 5120, 4, 4), ("llm.o", 5120, 5120, 6, 4),
          ("llm.gu", 34816, 5120, 1, 8), ("llm.down", 5120, 17408, 9, 8)]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bitdance_amd import engine as E
    from bitdance_amd._lib import check, lib
    st = torch.cuda.current_stream().cuda_stream
    for name, N, K, S, nw in SHAPES:
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        wp = E.pack_linear([w], "cuda")
        del w
        x = torch.randn(M, K, device="cuda")
        xf = torch.zeros(M * K, dtype=torch.bfloat16, device="cuda")
        check(lib().bd_rows_to_frag(xf.data_ptr(), x.data_ptr(), 1, M, K, M // 32, st))
        out = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
        for _ in range(REPS):
            check(lib().bd_gemm_partial(xf.data_ptr(), M // 32, wp.data_ptr(), N, K, S, nw, out.data_ptr(), st))
        torch.cuda.synchronize()
        del wp, out


def parse(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select dispatch_id, name, counter_value, duration from pmc_events where counter_name = 'FETCH_SIZE' "
                       "order by dispatch_id").fetchall()
    g = [r for r in rows if "gemm_kernel" in r[1]]
    assert len(g) == REPS * len(SHAPES), (len(g), REPS * len(SHAPES))
    res = {}
    for i, (name, N, K, S, nw) in enumerate(SHAPES):
        grp = g[i * REPS:(i + 1) * REPS][1:]                 # drop the first launch of each shape (cold TLB / code)
        kib = sum(r[2] for r in grp) / len(grp)
        fetched = 2.0 * kib * 1024.0                         # gfx950: FETCH_SIZE tallies 128 B requests as 64 B
        alg = N * K * 2
        res[name] = dict(N=N, K=K, splitk=S, nwaves=nw, fetch_size_kib_raw=round(kib, 1), hbm_read_bytes=round(fetched),
                         algorithmic_bytes=alg, ratio=round(fetched / alg, 4))
    out = dict(counter="FETCH_SIZE", unit="KiB raw; x2 gfx950 wide-read correction applied (MI355X_MICROARCH.md, HBM section)",
               rows_M=M, reps_averaged=REPS - 1, gemms=res)
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:10s} fetched {v['hbm_read_bytes'] / 1e6:8.1f} MB  algorithmic {v['algorithmic_bytes'] / 1e6:8.1f} MB  x{v['ratio']}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3])
