"""The native conv decoder alone, a few repetitions at one size: the target for `rocprofv3 --kernel-trace` when the per-kernel
split of a decode is wanted.  python tools/ae_native_profile.py [px] [batch] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import synthetic as syn          # noqa: E402
from bitdance_amd.ae_native import NativeDecoder   # noqa: E402
from bitdance_amd.autoencoder import VQModel       # noqa: E402

px = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ae = VQModel(**syn.AE_D16C32).eval()
ae.load_state_dict(syn.random_ae_state(syn.AE_D16C32, "cuda"), strict=True, assign=True)
ae.to("cuda")
nat = NativeDecoder(ae.decoder, "cuda")
x = torch.sign(torch.randn(B, 32, px // 16, px // 16, device="cuda"))
nat.decode(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    y = nat.decode(x)
torch.cuda.synchronize()
print(f"native decode {px} px batch {B}: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms")
