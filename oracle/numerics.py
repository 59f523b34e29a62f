"""Dtype policies for the oracle (test infrastructure, see oracle/__init__.py).

The reference runs its hot loop under ``torch.amp.autocast("cuda", dtype=bfloat16)``
(t2i_pipeline.py:130).  The oracle restates that flow with *explicit* casts so the
rounding points are visible and run identically on any CPU:

  * ``Policy("fp32")``     -- no casts at all: what the reference computes when it is
                              executed on CPU with fp32 weights (autocast("cuda") is
                              inert there).  Used to pin the restatement against the
                              reference bit-for-bit / to 1e-5.
  * ``Policy("autocast")`` -- the CUDA/HIP autocast policy: ``F.linear`` / ``matmul``
                              inputs are cast to bf16, accumulate in fp32, round the
                              result once to bf16; ``layer_norm`` and ``softmax`` run
                              in fp32 (autocast's fp32 list); everything else follows
                              normal type promotion of its operands.

  * ``Policy("fp8w")``     -- the framework's own fp8 weight mode (BASELINE config 5; no counterpart in the reference):
                              the autocast flow with the STREAMED Linear weights stored as OCP e4m3 with one fp32 scale per
                              output channel (scale = max|row| / 448, computed from the bf16 weights): products of bf16
                              activations with the exactly-converted e4m3 values accumulate in fp32, the scale multiplies
                              the fp32 sum, then bias and the single bf16 rounding.  Small Linears the product keeps in bf16
                              (time embedding, input_proj, final Linear, projector fc1) pass ``quant=False``.

Elementwise bf16 ops on CPU tensors (compute in fp32, round to bf16) follow the same
eager semantics as on the GPU, so the oracle keeps real ``torch.bfloat16`` tensors
where the reference would and lets torch's type promotion do the rest.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16
F32 = torch.float32


class Policy:
    def __init__(self, name: str = "autocast"):
        if name not in ("fp32", "autocast", "fp8w", "fp8wa"):
            raise ValueError(f"unknown policy {name!r}")
        self.name = name

    @property
    def amp(self) -> bool:
        return self.name in ("autocast", "fp8w", "fp8wa")

    # -- F.linear under the policy -------------------------------------------------
    def linear(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None, quant: bool = True, act8: bool = False) -> torch.Tensor:
        """``quant``: this Linear's weights are streamed (fp8 modes quantise them per output channel).  ``act8``: its input comes
        straight from a row kernel (LayerNorm-modulate, RMSNorm, the silu of the adaLN input): policy "fp8wa" quantises that input
        per ROW to e4m3 as well -- scale = amax / 448, q = e4m3(x * (448 / amax)) from the value the row kernel holds (no bf16 step
        in between) -- and the product runs on the fp8 matrix pipe (csrc/bd_gemm_kernel.h WT = 2)."""
        if not self.amp:
            return F.linear(x, w, b)
        xb = x.to(BF16).to(F32)
        wb = w.to(BF16).to(F32)
        if self.name == "fp8wa" and quant and act8:
            xs = x.to(F32)
            am = xs.abs().amax(dim=-1, keepdim=True)
            inv = torch.where(am > 0, 448.0 / am, torch.zeros_like(am))
            xq = (xs * inv).to(torch.float8_e4m3fn).to(F32)
            s = (wb.abs().amax(dim=1) / 448.0).clamp_min(1e-12)
            q = (wb / s[:, None]).to(torch.float8_e4m3fn).to(F32)
            acc = ((xq @ q.t()) * s) * (am / 448.0)
        elif self.name in ("fp8w", "fp8wa") and quant:
            s = (wb.abs().amax(dim=1) / 448.0).clamp_min(1e-12)
            q = (wb / s[:, None]).to(torch.float8_e4m3fn).to(F32)
            acc = (xb @ q.t()) * s
        else:
            acc = xb @ wb.t()
        if b is not None:
            acc = acc + b.to(BF16).to(F32)
        return acc.to(BF16)

    # -- torch.matmul under the policy ---------------------------------------------
    def matmul(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        if not self.amp:
            return a @ b
        return (a.to(BF16).to(F32) @ b.to(BF16).to(F32)).to(BF16)

    # -- F.layer_norm: autocast runs it in fp32 and returns fp32 --------------------
    def layer_norm(self, x, weight, bias, eps: float) -> torch.Tensor:
        if not self.amp:
            return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)
        w = None if weight is None else weight.to(F32)
        b = None if bias is None else bias.to(F32)
        return F.layer_norm(x.to(F32), (x.shape[-1],), w, b, eps)

    def cast_in(self, x: torch.Tensor) -> torch.Tensor:
        """What an autocast op does to a floating input."""
        return x.to(BF16) if self.amp else x


class fused_kernel:
    """The body of a FUSED attention kernel (flash_attn_func, F.scaled_dot_product_attention) restated with torch ops: a real
    ``torch.autocast`` context active around the caller must not re-cast the restated internals (the fused kernel is opaque to
    autocast: fp32 scores and statistics whatever the policy).  Inert when no autocast context is active -- every CPU golden
    run -- and what lets tests/test_gpu_autocast.py run this oracle under the DEVICE's own autocast with ``Policy("fp32")``."""

    def __init__(self, t: torch.Tensor):
        self._ctx = torch.autocast(t.device.type, enabled=False) if torch.is_autocast_enabled(t.device.type) else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def autocast_lower(*ts):
    """What an active device autocast does to the inputs of an op on its lower-precision list (F.scaled_dot_product_attention):
    cast floating inputs to the autocast dtype.  Identity without an active autocast context."""
    dev = ts[0].device.type
    if not torch.is_autocast_enabled(dev):
        return ts
    dt = torch.get_autocast_dtype(dev)
    return tuple(t.to(dt) if t.is_floating_point() else t for t in ts)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16, returned as fp32 (value-preserving)."""
    return x.to(BF16).to(F32)
