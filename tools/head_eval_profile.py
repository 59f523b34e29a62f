"""One AR step's head sampling (N + 1 evaluations of the 6-block BitDance-14B head, M = 128 rows) at true dimensions with
random weights, a few repetitions: the cheap target for `rocprofv3 --kernel-trace` when only the per-kernel times of the
evaluation chain are wanted (the full bench adds 30 GB of LLM weights and the MIOpen find pass).
python tools/head_eval_profile.py [reps] [n_steps] [num_images]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import engine as E                       # noqa: E402
from oracle import tiny_models as tm                        # noqa: E402  (shape table + seeded random weights only)
from oracle.true_dims import device_seeded_state            # noqa: E402


def main():
    with torch.cuda.stream(torch.cuda.Stream()):          # graph capture needs a non-default stream (as the pipeline uses)
        run()


def run():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = "cuda"
    cfgd = dict(ch_target=32, ch_cond=5120, ch_latent=5120, depth_latent=6, depth_adanln=2)
    sd = device_seeded_state(tm.head_shapes(cfgd), 101, dev)
    hw = E.HeadWeights.from_state_dict(sd, dev)
    del sd
    eng = E.Engine(hw, None, None, num_images=B, branches=2, device=dev, max_tokens=64, parallel_num=64)
    eng.set_schedule(n, 7.5, 1)
    eng.draw_noise(1)
    eng.reset([0] * (2 * B))
    eng.set_cond(torch.randn(2 * B, 64, 5120, device=dev))
    eng.head_sample()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.reset([0] * (2 * B))
        eng.head_sample()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"head_sample (eager): {dt * 1e3:.2f} ms per AR step, {dt / (n + 1) * 1e6:.1f} us per evaluation, rows {2 * B * 64}", flush=True)
    eng.capture(0)
    eng.reset([0] * (2 * B))
    eng.launch(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.reset([0] * (2 * B))
        eng.launch(0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"head_sample (hipGraph): {dt * 1e3:.2f} ms per AR step, {dt / (n + 1) * 1e6:.1f} us per evaluation", flush=True)


if __name__ == "__main__":
    main()
