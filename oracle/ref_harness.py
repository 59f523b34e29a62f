"""Import the UNMODIFIED reference from /root/reference for pinning the oracle.

Test infrastructure only, and only usable where /root/reference exists (this container,
never the GPU box).  Applies the three version shims of SURVEY.md section 8(c):

  1. ``flash_attn`` is not installed: register a stub module exposing
     ``flash_attn_func(q, k, v, causal=False)`` = the published FlashAttention-2 forward
     (fp32 statistics, P rounded to the input dtype before P.V) on [B,S,H,D]
     (import ``transformers`` first or its availability probe dies on the stub).
  2. transformers 5.x ``DynamicCache`` is not subscriptable but the reference does
     ``pkv[0][0].shape[2]`` (t2i_pipeline.py:207,257): add ``__getitem__``.
  3. the reference reuses the cond-sized all-True mask for the shorter uncond branch
     (t2i_pipeline.py:233,266); transformers<=4.5x sliced 4-D masks to the key length,
     5.x does not: slice it in a forward-pre-hook (semantics unchanged, mask is all ones).
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF_ROOT = os.environ.get("BITDANCE_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "modeling"))


_done = False


def install() -> None:
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    import transformers  # noqa: F401  (must precede the stub, see module docstring)
    from transformers import Qwen3ForCausalLM  # noqa: F401
    from transformers.activations import ACT2FN  # noqa: F401
    from transformers.cache_utils import DynamicCache

    if "flash_attn" not in sys.modules:
        stub = types.ModuleType("flash_attn")

        def flash_attn_func(q, k, v, causal=False, softmax_scale=None, **_):
            # FlashAttention-2 forward as published (Dao 2023, Alg. 1): fp32 scores and running
            # statistics, un-normalised P cast to the input dtype for P.V, fp32 output accumulator,
            # one normalisation and one rounding at the end.  Layout [B,S,H,D].
            assert not causal
            cd = q.dtype
            scale = q.shape[-1] ** -0.5 if softmax_scale is None else softmax_scale
            with torch.autocast("cpu", enabled=False):        # a fused kernel is opaque to autocast
                qf, kf, vf = (t.transpose(1, 2).float() for t in (q, k, v))
                s = (qf @ kf.transpose(-1, -2)) * scale
                m = s.amax(dim=-1, keepdim=True)
                p = torch.exp(s - m)
                l = p.sum(dim=-1, keepdim=True)
                o = (p.to(cd).float() @ vf) / l
                return o.to(cd).transpose(1, 2).contiguous()

        stub.flash_attn_func = flash_attn_func
        sys.modules["flash_attn"] = stub
    if not hasattr(DynamicCache, "__getitem__"):
        DynamicCache.__getitem__ = lambda self, i: (self.layers[i].keys, self.layers[i].values)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _done = True


def add_mask_slice_hook(qwen3_model) -> None:
    """Shim 3: slice an oversize 4-D mask to past+T keys before Qwen3Model.forward."""

    def hook(module, args, kwargs):
        m = kwargs.get("attention_mask")
        pkv = kwargs.get("past_key_values")
        if m is not None and m.dim() == 4:
            past = pkv.get_seq_length() if pkv is not None else 0
            t = kwargs["inputs_embeds"].shape[1]
            kwargs["attention_mask"] = m[..., : past + t]
        return args, kwargs

    qwen3_model.register_forward_pre_hook(hook, with_kwargs=True)


class ReplayNoise:
    """Monkey-patch torch.randn / randn_like to replay (and record) an injected sequence so CPU
    and GPU runs consume identical noise (SURVEY.md section 7 'Hard parts')."""

    def __init__(self, seq=None, seed: int = 0):
        self.seq = None if seq is None else list(seq)
        self.gen = torch.Generator().manual_seed(seed)
        self.record: list[torch.Tensor] = []
        self.calls = 0

    def _next(self, shape):
        self.calls += 1
        if self.seq is not None:
            t = self.seq.pop(0)
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        else:
            t = torch.empty(*shape, dtype=torch.float32).normal_(generator=self.gen)
        self.record.append(t.clone())
        return t.clone()

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)):
                size = tuple(size[0])
            return self._next(size)

        torch.randn = randn
        torch.randn_like = lambda x, **kw: self._next(tuple(x.shape))
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like
        return False


class CudaAutocastOnCpu:
    """Emulate the CUDA/HIP bf16 autocast policy on CPU for the *reference* code: CPU autocast
    (linear/matmul/sdpa -> bf16) plus layer_norm forced to fp32 as CUDA autocast's fp32 list
    does.  (softmax is only reached inside SDPA / the flash stub on the P=64 path.)"""

    def __enter__(self):
        self._ln = torch.nn.functional.layer_norm
        orig = self._ln

        def layer_norm32(x, shape, weight=None, bias=None, eps=1e-5):
            with torch.autocast("cpu", enabled=False):
                return orig(x.float(), shape, None if weight is None else weight.float(),
                            None if bias is None else bias.float(), eps)

        torch.nn.functional.layer_norm = layer_norm32
        self._ac = torch.autocast("cpu", dtype=torch.bfloat16)
        self._ac.__enter__()
        return self

    def __exit__(self, *exc):
        self._ac.__exit__(*exc)
        torch.nn.functional.layer_norm = self._ln
        return False
