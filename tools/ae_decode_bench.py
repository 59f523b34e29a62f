"""Steady-state time of the conv decoder (autoencoder.py Decoder.forward) under the layout / dtype options torch + MIOpen offer:
NCHW under bf16 autocast (what the pipeline runs), channels-last, and bf16 weights without autocast.
python tools/ae_decode_bench.py [1024|256] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import synthetic as syn          # noqa: E402
from bitdance_amd.autoencoder import VQModel       # noqa: E402

px = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ae = VQModel(**syn.AE_D16C32).eval()
ae.load_state_dict(syn.random_ae_state(syn.AE_D16C32, "cuda"), strict=True, assign=True)
ae.to("cuda")
x = torch.sign(torch.randn(B, 32, px // 16, px // 16, device="cuda"))
torch.backends.cudnn.benchmark = True


def timed(fn, tag):
    with torch.no_grad():
        t0 = time.perf_counter(); y = fn(); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    print(f"{tag:50s} first {t_first:7.2f} s   steady {dt * 1e3:8.1f} ms   finite {bool(torch.isfinite(y.float()).all())}", flush=True)
    return y


def nchw():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return ae.decode(x)


y0 = timed(nchw, f"NCHW, bf16 autocast ({px} px, batch {B})")
ae_cl = VQModel(**syn.AE_D16C32).eval()
ae_cl.load_state_dict(ae.state_dict(), assign=False)
ae_cl = ae_cl.to("cuda").to(memory_format=torch.channels_last)
xcl = x.contiguous(memory_format=torch.channels_last)


def cl():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return ae_cl.decode(xcl)


y1 = timed(cl, "channels-last, bf16 autocast")
ae16 = VQModel(**syn.AE_D16C32).eval()
ae16.load_state_dict(ae.state_dict())
ae16 = ae16.to("cuda", torch.bfloat16).to(memory_format=torch.channels_last)
x16 = xcl.to(torch.bfloat16)
y2 = timed(lambda: ae16.decode(x16), "channels-last, bf16 weights and activations")
print("max |d| cl vs nchw:", (y1.float() - y0.float()).abs().max().item(), " bf16-all vs nchw:", (y2.float() - y0.float()).abs().max().item())
from bitdance_amd.ae_native import NativeDecoder   # noqa: E402
nat = NativeDecoder(ae.decoder, "cuda")
y3 = timed(lambda: nat.decode(x), "native gfx950 kernels (csrc/bd_conv.hip)")
d = (y3.float() - y0.float()).abs()
print(f"native vs MIOpen NCHW: max |d| {d.max().item():.4f} mean {d.mean().item():.5f} (mean |x| {y0.float().abs().mean().item():.3f})")
