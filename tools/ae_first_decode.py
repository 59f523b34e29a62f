"""First-call latency of the AE decode (MIOpen find) at 1024 px: python tools/ae_first_decode.py [naive0]
naive0: MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 (keep MIOpen's find pass from benchmarking the naive direct solver)."""
import os
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "naive0":
    os.environ["MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD"] = "0"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import synthetic as syn          # noqa: E402
from bitdance_amd.autoencoder import VQModel       # noqa: E402

ae = VQModel(**syn.AE_D16C32).eval()
ae.load_state_dict(syn.random_ae_state(syn.AE_D16C32, "cuda"), strict=True, assign=True)
ae.to("cuda")
x = torch.sign(torch.randn(1, 32, 64, 64, device="cuda"))
torch.backends.cudnn.benchmark = True
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = ae.decode(x)
    torch.cuda.synchronize()
    print(f"decode call {i}: {time.perf_counter() - t0:.2f} s  finite={bool(torch.isfinite(y.float()).all())}", flush=True)
