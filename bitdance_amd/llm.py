"""Qwen3 decoder for the BitDance loop: the prompt passes (prefill).

The decode step (P new tokens against the KV cache, the part that runs 64x per image) is the HIP path in
engine.Engine.llm_step.  The prefill (once per image: a causal call over the prompt, then one block-bidirectional call over
the last P tokens, t2i_pipeline.py:199-217) runs

  * natively (``prefill_native`` / ``native_block``, the default): the SAME step kernels over blocks of P prompt tokens with
    three switches -- causal masking inside the block, a bf16 residual stream and bf16 RoPE tables -- i.e. the dtype flow
    HF's Qwen3 has with bf16 hidden states (HF modeling_qwen3.py:59-64,137,140-170,294-323).  No second copy of the LLM
    weights, and under tensor parallelism the row-split Linears use the same in-step exchange as the decode;
  * or with torch ops (``prefill_block``: hipBLASLt + SDPA on the original-layout weights, kept only when the weights were
    packed with ``keep_for_prefill=True``): the round-1 path, still used as a cross-check in the tests.

Both write post-RoPE K / V into the engine's static KV cache (K [seq][kvh][pos][128], V transposed [seq][kvh][128][pos]).
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


# ---------------------------------------------------------------------------------------------------------------------
def _run_block(eng: Engine, rows: list, kv: list[int], causal: bool) -> torch.Tensor:
    """One native step over a block of <= P PROMPT tokens per sequence.  rows[i]: [t_i, D] bf16 (t_i <= P) or None; kv[i]: tokens
    already cached for sequence i.  Rows past t_i are zero padding: their K / V land behind the real tokens and are
    overwritten by the next block (every later block starts at kv[i] + t_i), causal queries never see them.
    Returns last_hidden_state of the block [nseq, P, D] (bf16 values in fp32 storage; rows >= t_i are meaningless)."""
    P, D = eng.P, eng.llm.cfg["hidden_size"]
    nseq = len(rows)
    R = eng.residual()
    R[: nseq * P].zero_()
    for i, r in enumerate(rows):
        if r is not None and r.shape[0]:
            R[i * P: i * P + r.shape[0]].copy_(r)
    eng.reset(kv)
    for k, v in (("rt.llm_causal", int(causal)), ("rt.llm_bf16", 1), ("rt.no_advance", 1), ("rt.emit_cond", 0)):
        eng.set_int(k, v)
    try:
        eng.llm_step()
    finally:
        for k, v in (("rt.llm_causal", 0), ("rt.llm_bf16", 0), ("rt.no_advance", 0), ("rt.emit_cond", 1)):
            eng.set_int(k, v)
    return eng.hidden().view(nseq, P, D).clone()


@torch.no_grad()
def prefill_native(eng: Engine, seq_embeds: list) -> tuple[torch.Tensor, list[int]]:
    """The reference's two prompt calls for EVERY sequence of the engine at once (sequences may differ in length: cond /
    uncond prompts): causal over tokens [0, T_i - P), then all-visible over the last P tokens.  seq_embeds[i]: [T_i, D] bf16.
    Returns (last_hidden_state of the last P tokens [nseq, P, D], cache lengths T_i)."""
    P = eng.P
    nseq = eng.branches * eng.B
    if len(seq_embeds) != nseq or any(e.shape[0] < P for e in seq_embeds):
        raise ValueError("prefill_native: one [T >= P, D] embedding tensor per sequence")
    T0 = [e.shape[0] - P for e in seq_embeds]
    for c in range((max(T0) + P - 1) // P):
        rows = [e[c * P: min((c + 1) * P, t0)] if c * P < t0 else None for e, t0 in zip(seq_embeds, T0)]
        _run_block(eng, rows, [min(c * P, t0) for t0 in T0], causal=True)
    hid = _run_block(eng, [e[t0:] for e, t0 in zip(seq_embeds, T0)], T0, causal=False)
    return hid, [e.shape[0] for e in seq_embeds]


@torch.no_grad()
def native_block(eng: Engine, x: torch.Tensor, past: int, causal: bool) -> torch.Tensor:
    """``Qwen3Model.forward`` over x [B, T, D] (bf16) for all B sequences of the engine with ``past`` cached tokens each:
    causal (any T, in blocks of P) or all-visible (T == P).  Returns last_hidden_state [B, T, D] bf16."""
    B, T, D = x.shape
    P = eng.P
    if not causal and T != P:
        raise NotImplementedError("an all-visible prompt call must be exactly one block of parallel_num tokens")
    outs = []
    for c in range((T + P - 1) // P):
        t = min(P, T - c * P)
        h = _run_block(eng, [x[b, c * P: c * P + t] for b in range(B)], [past + c * P] * B, causal)
        outs.append(h[:, :t])
    return torch.cat(outs, dim=1).to(BF16)
