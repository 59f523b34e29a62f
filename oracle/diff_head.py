"""Oracle: binary diffusion head (adaLN DiT "TransEncoder") -- test infrastructure only.

Restates /root/reference/modeling/vision_head/flow_head_parallel_x.py:
  timestep_embedding :12-27      TimestepEmbedder.forward :140-143
  FinalLayer.forward :169-173    Attention.forward :192-220
  TransBlock.forward :242-252    TransEncoder.forward :325-342
  DiffHead.sample    :107-120

Weights are a flat dict keyed exactly like ``vision_head.safetensors``
(``net.input_proj.weight`` ...), see SURVEY.md section 8(b).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .numerics import BF16, F32, Policy, fused_kernel
from . import sampler


def timestep_features(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0,
                      time_factor: float = 1000.0) -> torch.Tensor:
    """flow_head_parallel_x.py:12-27 (cos first, then sin)."""
    half = dim // 2
    t = time_factor * t.float()
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=F32, device=t.device) / half)
    args = t[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb.to(t)


def time_embed(w: dict, t: torch.Tensor, pol: Policy) -> torch.Tensor:
    """TimestepEmbedder.forward :140-143 (Linear -> SiLU -> Linear)."""
    f = timestep_features(t, w["net.time_embed.mlp.0.weight"].shape[1])
    h = pol.linear(f, w["net.time_embed.mlp.0.weight"], w["net.time_embed.mlp.0.bias"], quant=False)
    h = F.silu(h)
    return pol.linear(h, w["net.time_embed.mlp.2.weight"], w["net.time_embed.mlp.2.bias"], quant=False)


def attention(w: dict, pre: str, x: torch.Tensor, n_head: int, pol: Policy) -> torch.Tensor:
    """Attention.forward :192-220.  seq<=32: explicit softmax path (:203-208); otherwise the
    flash-attention call (:210-215), restated as softmax(q k^T * scale) v with fp32
    accumulation, un-normalised P rounded to the compute dtype before P.V (what a
    flash kernel does) and the output rounded once."""
    bsz, seqlen, dim = x.shape
    hd = dim // n_head
    scale = hd ** -0.5
    qkv = pol.linear(x, w[pre + "wqkv.weight"], w[pre + "wqkv.bias"], act8=True)
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.view(bsz, seqlen, n_head, hd).transpose(1, 2)
    k = k.view(bsz, seqlen, n_head, hd).transpose(1, 2)
    v = v.view(bsz, seqlen, n_head, hd).transpose(1, 2)
    if seqlen <= 32:
        q = q * scale
        att = pol.matmul(q, k.transpose(-1, -2))
        att = F.softmax(att.float() if pol.amp else att, dim=-1)
        out = pol.matmul(att, v)
    else:
        cd = q.dtype
        with fused_kernel(q):
            s = (q.to(F32) @ k.to(F32).transpose(-1, -2)) * scale
            m = s.amax(dim=-1, keepdim=True)
            p = torch.exp(s - m)
            l = p.sum(dim=-1, keepdim=True)
            out = (p.to(cd).to(F32) @ v.to(F32)) / l
            out = out.to(cd)
    out = out.transpose(1, 2).contiguous().view(bsz, seqlen, dim)
    return pol.linear(out, w[pre + "wo.weight"], w[pre + "wo.bias"])


def trans_block(w: dict, i: int, x, mods, n_head: int, pol: Policy):
    """TransBlock.forward :242-252 (SwiGLU variant, use_swiglu=True as in bitdance_14b_64x.yaml:33)."""
    s1, b1, g1, s2, b2, g2 = mods
    pre = f"net.res_blocks.{i}."
    h = pol.layer_norm(x, w[pre + "norm1.weight"], w[pre + "norm1.bias"], 1e-6) * (1 + s1) + b1
    h = attention(w, pre + "attn.", h, n_head, pol)
    x = x + h * g1
    h = pol.layer_norm(x, w[pre + "norm2.weight"], w[pre + "norm2.bias"], 1e-6) * (1 + s2) + b2
    if (pre + "w1.weight") in w:
        h1, h2 = pol.linear(h, w[pre + "w1.weight"], w[pre + "w1.bias"], act8=True).chunk(2, dim=-1)
        h = pol.linear(F.silu(h1) * h2, w[pre + "w2.weight"], w[pre + "w2.bias"])
    else:  # non-SwiGLU MLP (:246-248)
        h = pol.linear(h, w[pre + "mlp.0.weight"], w[pre + "mlp.0.bias"])
        h = pol.linear(F.silu(h), w[pre + "mlp.2.weight"], w[pre + "mlp.2.bias"])
    return x + h * g2


def count(w: dict, prefix: str) -> int:
    idx = {int(k[len(prefix):].split(".")[0]) for k in w if k.startswith(prefix)}
    return max(idx) + 1 if idx else 0


def net_forward(w: dict, x: torch.Tensor, t: torch.Tensor, c: torch.Tensor, pol: Policy,
                final_sigmoid: bool = True, trace: dict | None = None, head_dim: int = 128) -> torch.Tensor:
    """TransEncoder.forward :325-342.  x [M',P,C] fp32, t [M'] fp32, c [M',P,Dz] fp32."""
    n_blocks = count(w, "net.res_blocks.")
    n_ada = count(w, "net.ada_ln_blocks.")
    switch = max(1, n_blocks // n_ada)
    dim = w["net.input_proj.weight"].shape[0]
    n_head = dim // head_dim                  # TransBlock.__init__ :227 (128); imagenet diff_head_parallel.py:207 (64)
    x = pol.linear(x, w["net.input_proj.weight"], w["net.input_proj.bias"], quant=False)
    te = time_embed(w, t, pol).unsqueeze(1)
    ce = pol.linear(c, w["net.cond_embed.weight"], w["net.cond_embed.bias"])
    y = F.silu(te + ce)
    if trace is not None:
        trace["x0"], trace["y"] = x, y
    mods = pol.linear(y, w["net.ada_ln_blocks.0.weight"], w["net.ada_ln_blocks.0.bias"], act8=True).chunk(6, dim=-1)
    for i in range(n_blocks):
        if i > 0 and i % switch == 0:
            j = i // switch
            mods = pol.linear(y, w[f"net.ada_ln_blocks.{j}.weight"], w[f"net.ada_ln_blocks.{j}.bias"], act8=True).chunk(6, dim=-1)
        x = trans_block(w, i, x, mods, n_head, pol)
        if trace is not None:
            trace[f"x{i + 1}"] = x
    scale, shift = pol.linear(y, w["net.final_layer.ada_ln_modulation.weight"],
                              w["net.final_layer.ada_ln_modulation.bias"], act8=True).chunk(2, dim=-1)
    h = pol.layer_norm(x, None, None, 1e-6) * (1.0 + scale) + shift
    out = pol.linear(h, w["net.final_layer.linear.weight"], w["net.final_layer.linear.bias"], quant=False)
    if not final_sigmoid:                     # imagenet variant, diff_head_parallel.py:310
        return out
    return 2 * torch.sigmoid(out) - 1


def is_mlp_head(w: dict) -> bool:
    return "net.res_blocks.0.norm.weight" in w and "net.res_blocks.0.attn.wqkv.weight" not in w


def mlp_net_forward(w: dict, x: torch.Tensor, t: torch.Tensor, c: torch.Tensor, pol: Policy) -> torch.Tensor:
    """The 1x ImageNet models' head: /root/reference/imagenet_gen/src/diff_head.py MlpEncoder.forward :228-253 with
    ResBlock.forward :133-137 and FinalLayer.forward :147-151 (no output squash).  x [N, C] fp32, t [N], c [N, Dz] fp32
    (any leading shape: every op is row-wise)."""
    n_blocks = count(w, "net.res_blocks.")
    n_ada = count(w, "net.ada_ln_blocks.")
    switch = max(1, n_blocks // n_ada)
    x = pol.linear(x, w["net.input_proj.weight"], w["net.input_proj.bias"], quant=False)
    te = time_embed(w, t, pol)
    if c.dim() == 3:
        te = te.unsqueeze(1)
    ce = pol.linear(c, w["net.cond_embed.weight"], w["net.cond_embed.bias"])
    y = F.silu(te + ce)
    scale, shift, gate = pol.linear(y, w["net.ada_ln_blocks.0.weight"], w["net.ada_ln_blocks.0.bias"], act8=True).chunk(3, dim=-1)
    for i in range(n_blocks):
        if i > 0 and i % switch == 0:
            j = i // switch
            scale, shift, gate = pol.linear(y, w[f"net.ada_ln_blocks.{j}.weight"], w[f"net.ada_ln_blocks.{j}.bias"], act8=True).chunk(3, dim=-1)
        pre = f"net.res_blocks.{i}."
        h = pol.layer_norm(x, w[pre + "norm.weight"], w[pre + "norm.bias"], 1e-6) * (1 + scale) + shift
        h1, h2 = pol.linear(h, w[pre + "w1.weight"], w[pre + "w1.bias"], act8=True).chunk(2, dim=-1)
        h = pol.linear(F.silu(h1) * h2, w[pre + "w2.weight"], w[pre + "w2.bias"])
        x = x + h * gate
    scale, shift = pol.linear(y, w["net.final_layer.ada_ln_modulation.weight"],
                              w["net.final_layer.ada_ln_modulation.bias"], act8=True).chunk(2, dim=-1)
    h = pol.layer_norm(x, None, None, 1e-6) * (1.0 + scale) + shift
    return pol.linear(h, w["net.final_layer.linear.weight"], w["net.final_layer.linear.bias"], quant=False)


def sample(w: dict, z: torch.Tensor, cfg: float, num_sampling_steps: int, noise, pol: Policy,
           time_shift: float = 1.0, trace: list | None = None) -> torch.Tensor:
    """DiffHead.sample :107-120 -> euler_maruyama.  z [cfg_mult*B, P, Dz] fp32."""
    ch_target = w["net.input_proj.weight"].shape[1]
    fwd = lambda x, t, c: net_forward(w, x, t, c, pol)
    return sampler.euler_maruyama(ch_target, fwd, z, cfg, num_sampling_steps, noise,
                                  time_shift=time_shift, trace=trace)
