"""Oracle: decoder-only LLM forward as the reference drives it -- test infrastructure only.

The arithmetic lives in third-party HF ``transformers`` (``Qwen3Model``), pinned to 4.57.0 by
/root/reference/requirements.txt:1 and NOT vendored; the container ships 5.15.0 whose source
is the algorithm restated here (``HF:`` = transformers/models/qwen3/modeling_qwen3.py):

  Qwen3RMSNorm.forward        HF:59-64      Qwen3MLP.forward            HF:81-83
  Qwen3RotaryEmbedding        HF:124-137    rotate_half / apply_rotary  HF:140-170
  Qwen3Attention.forward      HF:241-280    Qwen3DecoderLayer.forward   HF:294-323
  Qwen3Model.forward          HF:367-427    DynamicLayer.update (cat)   cache_utils.py

Reference call sites anchoring parity: /root/reference/modeling/t2i_pipeline.py:199-236
(prefill: causal call on the prompt, then one block-bidirectional call with an all-True
4-D mask) and :261-268 (decode: P tokens per call, all-True mask, fp32 inputs_embeds).

Weights: flat dict with HF checkpoint names (``model.layers.0.self_attn.q_proj.weight`` ...).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .numerics import BF16, F32, Policy, autocast_lower, fused_kernel


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """HF:59-64 -- fp32 statistics, cast back to the INPUT dtype, then times weight."""
    dt = x.dtype
    h = x.to(F32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def rope_tables(cfg: dict, position_ids: torch.Tensor, dtype) -> tuple[torch.Tensor, torch.Tensor]:
    """HF:94-137 (default rope): fp32 cos/sin, cast to the hidden-state dtype."""
    hd = cfg["head_dim"]
    inv_freq = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float, device=position_ids.device) / hd))
    freqs = position_ids.float()[:, None] * inv_freq[None, :]       # == the K=1 matmul at HF:131
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def sdpa(q, k, v, mask, scale: float, pol: Policy):
    """HF sdpa_attention_forward -> F.scaled_dot_product_attention with GQA (HF:sdpa_attention.py).
    Restated as softmax(q k^T scale + mask) v: fp32 scores/softmax, the un-normalised P rounded to
    the compute dtype before P.V, fp32 accumulation, one rounding of the output."""
    if pol.amp:
        q, k, v = q.to(BF16), k.to(BF16), v.to(BF16)
    q, k, v = autocast_lower(q, k, v)                      # (a real autocast context: tests/test_gpu_autocast.py; inert otherwise)
    cd = q.dtype
    with fused_kernel(q):
        rep = q.shape[1] // k.shape[1]
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
        s = (q.to(F32) @ k.to(F32).transpose(-1, -2)) * scale
        if mask is not None:
            s = s.masked_fill(~mask, float("-inf"))
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        out = (p.to(cd).to(F32) @ v.to(F32)) / l
        return out.to(cd)


def model_forward(w: dict, cfg: dict, inputs_embeds: torch.Tensor, cache: list | None,
                  attention_mask: torch.Tensor | None, pol: Policy, trace: dict | None = None):
    """Qwen3Model.forward HF:367-427 with use_cache=True.

    inputs_embeds [B,T,D]; cache = list of [K,V] per layer ([B,kvh,L,hd]) or None;
    attention_mask None -> causal (prefill, t2i_pipeline.py:199-203), or a bool
    [B,1,T,>=past+T] tensor (all True in the reference, :206-218, :256-268).
    Returns (last_hidden_state [B,T,D], cache)."""
    L = cfg["num_hidden_layers"]
    nh, nkv, hd = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    eps = cfg["rms_norm_eps"]
    B, T, _ = inputs_embeds.shape
    if cache is None:
        cache = [None] * L
    past = 0 if cache[0] is None else cache[0][0].shape[2]
    dev = inputs_embeds.device
    pos = torch.arange(T, device=dev) + past
    h = inputs_embeds
    cos, sin = rope_tables(cfg, pos, h.dtype)            # [T,hd], dtype of the hidden states
    cos, sin = cos[None, None], sin[None, None]
    if attention_mask is None:
        i = torch.arange(T, device=dev)[:, None] + past
        j = torch.arange(past + T, device=dev)[None, :]
        mask = (j <= i)[None, None]
    else:
        mask = attention_mask[..., : past + T]             # shim 3 of SURVEY 8(c): slice to key length
    for li in range(L):
        p = f"model.layers.{li}."
        resid = h
        x = rms_norm(h, w[p + "input_layernorm.weight"], eps)
        q = pol.linear(x, w[p + "self_attn.q_proj.weight"], act8=True).view(B, T, nh, hd)
        k = pol.linear(x, w[p + "self_attn.k_proj.weight"], act8=True).view(B, T, nkv, hd)
        v = pol.linear(x, w[p + "self_attn.v_proj.weight"], act8=True).view(B, T, nkv, hd)
        q = rms_norm(q, w[p + "self_attn.q_norm.weight"], eps).transpose(1, 2)
        k = rms_norm(k, w[p + "self_attn.k_norm.weight"], eps).transpose(1, 2)
        v = v.transpose(1, 2)
        q = (q * cos) + (rotate_half(q) * sin)
        k = (k * cos) + (rotate_half(k) * sin)
        if cache[li] is not None:                        # DynamicLayer.update: torch.cat (type-promoting)
            k = torch.cat([cache[li][0], k], dim=2)
            v = torch.cat([cache[li][1], v], dim=2)
        cache[li] = [k, v]
        a = sdpa(q, k, v, mask, hd ** -0.5, pol)
        a = a.transpose(1, 2).reshape(B, T, nh * hd)
        if trace is not None and li == 0:
            trace["attn0"] = a
        h = resid + pol.linear(a, w[p + "self_attn.o_proj.weight"])
        resid = h
        x = rms_norm(h, w[p + "post_attention_layernorm.weight"], eps)
        g = pol.linear(x, w[p + "mlp.gate_proj.weight"], act8=True)
        u = pol.linear(x, w[p + "mlp.up_proj.weight"], act8=True)
        h = resid + pol.linear(F.silu(g) * u, w[p + "mlp.down_proj.weight"])
        if trace is not None:
            trace[f"h{li}"] = h
    return rms_norm(h, w["model.norm.weight"], eps), cache
