"""Qwen3 decoder for the BitDance loop: native decode step + prefill on hipBLASLt/SDPA.

The decode step (64 new tokens against the KV cache, the part that runs 64x per image) is the HIP path in
engine.Engine.llm_step.  The prefill (once per image, SURVEY.md section 8f rank 1 = "next") runs here with
torch ops in exactly the dtype flow HF's Qwen3 has under bf16 autocast with bf16 hidden states
(HF modeling_qwen3.py:59-64,140-170,241-323), and writes post-RoPE K / V straight into the engine's static
KV cache (K [seq][kvh][pos][128], V transposed [seq][kvh][128][pos]).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .engine import Engine, LlmWeights

BF16 = torch.bfloat16


def _rms(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    h = x.float()
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(x.dtype)


def _rot(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


@torch.no_grad()
def prefill_block(eng: Engine, lw: LlmWeights, x: torch.Tensor, seq0: int, past: int, causal: bool) -> torch.Tensor:
    """One Qwen3Model.forward over x [B,T,D] (bf16) for sequences seq0..seq0+B-1 with `past` cached tokens.
    causal=True: standard causal mask (t2i_pipeline.py:199-203); False: every query sees all past+T keys
    (the all-True mask of :206-218).  Appends K/V to the engine cache, returns last_hidden_state (bf16)."""
    c = lw.cfg
    tp = getattr(lw, "tp_size", 1)                      # tensor parallel: this rank's heads / FFN slice, partial sums all-reduced
    L, nh, nkv, hd, eps = (c["num_hidden_layers"], c["num_attention_heads"] // tp, c["num_key_value_heads"] // tp,
                           c["head_dim"], c["rms_norm_eps"])
    reduce_ = eng.comm.all_reduce_ if (tp > 1 and eng.comm is not None) else (lambda t: t)
    B, T, _ = x.shape
    nseq = eng.branches * eng.B
    kc = eng.ws["llm.k_cache"].view(BF16).view(L, nseq, nkv, eng.Lmax, hd)
    vc = eng.ws["llm.vt_cache"].view(BF16).view(L, nseq, nkv, hd, eng.Lmax)
    cos = eng.cos[past:past + T].to(x.dtype)[None, None]          # cast to the hidden dtype (HF:137)
    sin = eng.sin[past:past + T].to(x.dtype)[None, None]
    sd = lw.sd
    h = x
    for li in range(L):
        p = f"model.layers.{li}."
        r = h
        a = _rms(h, sd[p + "input_layernorm.weight"], eps)
        q = F.linear(a, sd[p + "self_attn.q_proj.weight"]).view(B, T, nh, hd)
        k = F.linear(a, sd[p + "self_attn.k_proj.weight"]).view(B, T, nkv, hd)
        v = F.linear(a, sd[p + "self_attn.v_proj.weight"]).view(B, T, nkv, hd).transpose(1, 2)
        q = _rms(q, sd[p + "self_attn.q_norm.weight"], eps).transpose(1, 2)
        k = _rms(k, sd[p + "self_attn.k_norm.weight"], eps).transpose(1, 2)
        q = (q * cos) + (_rot(q) * sin)
        k = (k * cos) + (_rot(k) * sin)
        kc[li, seq0:seq0 + B, :, past:past + T] = k
        vc[li, seq0:seq0 + B, :, :, past:past + T] = v.transpose(2, 3)
        kk = kc[li, seq0:seq0 + B, :, :past + T]
        vv = vc[li, seq0:seq0 + B, :, :, :past + T].transpose(2, 3)
        rep = nh // nkv
        kk = kk.repeat_interleave(rep, dim=1)
        vv = vv.repeat_interleave(rep, dim=1)
        if causal and past == 0:
            o = F.scaled_dot_product_attention(q, kk, vv, is_causal=True)
        elif causal:
            i = torch.arange(T, device=x.device)[:, None] + past
            j = torch.arange(past + T, device=x.device)[None, :]
            o = F.scaled_dot_product_attention(q, kk, vv, attn_mask=(j <= i))
        else:
            o = F.scaled_dot_product_attention(q, kk, vv)
        o = o.transpose(1, 2).reshape(B, T, nh * hd)
        h = r + reduce_(F.linear(o, sd[p + "self_attn.o_proj.weight"]))
        r = h
        a = _rms(h, sd[p + "post_attention_layernorm.weight"], eps)
        g = F.linear(a, sd[p + "mlp.gate_proj.weight"])
        u = F.linear(a, sd[p + "mlp.up_proj.weight"])
        h = r + reduce_(F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"]))
    return _rms(h, sd["model.norm.weight"], eps)
