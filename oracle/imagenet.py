"""Oracle: class-conditional ImageNet BitDance (parallel 16x / 4x variants and the 1x variant) -- test infrastructure only.

Restates /root/reference/imagenet_gen/src:
  model_parallel.py   MLPConnector.forward :73-75     get_block_causal_mask :90-101
                      RoPE table + mask buffers :197-215   forward_model :342-350
                      head_sample :352-369 (linear CFG ramp)   sample :371-419
  layers_parallel.py  Attention.forward :135-168  naive_attention :120-133  update_kv_cache :110-118
                      FeedForward.forward :187-189  TransformerBlock.forward_onestep :229-241
                      get_2d_pos :235-252  precompute_freqs_cis_2d :255-270  apply_rotary_emb :273-290
  utils.py            patchify_raster_2d :91-113  unpatchify_raster :76-88
  diff_head_parallel.py  = the T2I head with head_dim 64, always the explicit-softmax attention, no final sigmoid
                      (:192-200, :296-310) -> oracle.diff_head with head_dim=64, final_sigmoid=False
  sampling_parallel.py   = sampling_x.py (same arithmetic) -> oracle.sampler

The 1x models (imagenet_gen/src/model.py, layers.py, diff_head.py, sampling.py) are the same
loop with parallel_num = 1: no query tokens (model.py:372-375), a purely causal mask (layers.py:126-129 == the block mask
with 1-token blocks), the RoPE table without re-ordering and ``[:-1]`` (model.py:181-190), plain-raster un-patchify
(model.py:246-255 with patch_size 1) and the MLP head (diff_head.py:228-253 -> oracle.diff_head.mlp_net_forward);
sampling.py == sampling_parallel.py up to the rank of the row tensors.

Weights: flat dict keyed like the reference's ``state_dict()`` minus ``vae.*``.
Dtype flow under the CUDA bf16 autocast policy (probed on the GPU box, tools/probe_autocast.py): ``rms_norm`` is in
neither autocast list (output dtype = input dtype), ``embedding`` and parameters are fp32, so step 0 carries an fp32
residual stream and every later step a bf16 one (``proj_in`` output is bf16); K/V are stored in an fp32 cache.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import diff_head
from .numerics import F32, Policy


# ------------------------------------------------------------------------------------------ tables
def get_2d_pos(resolution: int, patch: int) -> torch.Tensor:
    """layers_parallel.py:235-252 with num_scales=1: patch-centre (x, y) coordinates."""
    n = resolution // patch
    centers = (torch.arange(n, dtype=F32) + 0.5) * (float(n) / n)
    gy, gx = torch.meshgrid(centers, centers, indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=1)


def patchify_raster_2d(x: torch.Tensor, p: int, H: int, W: int) -> torch.Tensor:
    """utils.py:91-113: (H W) -> (H/p W/p p p) token order."""
    n, c1, c2 = x.shape
    y = x.reshape(H // p, p, W // p, p, c1 * c2).permute(0, 2, 1, 3, 4).reshape(n, c1 * c2)
    return y.view(n, c1, c2)


def rope_table(cfg: dict) -> torch.Tensor:
    """[cls + P-1 + h*w - P, head_dim/2, 2] (cos, sin): zeros-position for class/query tokens, 2-D positions (+1) for
    image tokens in patch-raster order, last P rows dropped.  model_parallel.py:197-212, layers_parallel.py:255-270."""
    hd = cfg["dim"] // cfg["n_head"]
    hw = cfg["resolution"] // (cfg["down_size"] * cfg["patch_size"])
    P = cfg["parallel_num"]
    pos = get_2d_pos(cfg["resolution"], cfg["down_size"] * cfg["patch_size"])
    half = hd // 2
    freqs = 1.0 / (10000 ** (torch.arange(0, half, 2)[: half // 2].float() / half))
    t = pos + 1.0
    t = torch.cat([torch.zeros(cfg["cls_token_num"] + P - 1, 2), t], dim=0)
    fr = torch.outer(t.flatten(), freqs).view(t.shape[0], -1)
    fc = torch.stack([torch.cos(fr), torch.sin(fr)], dim=-1)
    fc[-hw * hw:] = patchify_raster_2d(fc[-hw * hw:], int(P ** 0.5), hw, hw)
    return fc[:-P]


def block_causal_mask(total: int, causal: int, block: int) -> torch.Tensor:
    """model_parallel.py:90-101: causal, with each `block`-token group after the first `causal` tokens bidirectional."""
    m = torch.zeros(total, total)
    m.masked_fill_(torch.triu(torch.ones(total, total), diagonal=1).bool(), float("-inf"))
    for i in range(causal, total, block):
        m[i:i + block, i:i + block] = 0
    return m


def apply_rope(x: torch.Tensor, fc: torch.Tensor) -> torch.Tensor:
    """layers_parallel.py:273-290: interleaved pairs, fp32 arithmetic, result cast back to x's dtype."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    fc = fc.view(1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack([xs[..., 0] * fc[..., 0] - xs[..., 1] * fc[..., 1],
                       xs[..., 1] * fc[..., 0] + xs[..., 0] * fc[..., 1]], dim=-1)
    return out.flatten(3).type_as(x)


# ------------------------------------------------------------------------------------------ transformer
def rms(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return F.rms_norm(x, (x.shape[-1],), w, 1e-6)           # nn.RMSNorm(dim, eps=1e-6): output dtype = input dtype


def attention(w: dict, pre: str, x, mask, fc, cache, start: int, end: int, n_head: int, pol: Policy):
    """Attention.forward :135-168 on the KV-cache path (update_kv_cache :110-118, naive_attention :120-133)."""
    bsz, T, dim = x.shape
    hd = dim // n_head
    q, k, v = pol.linear(x, w[pre + "wqkv.weight"]).chunk(3, dim=-1)
    q, k, v = (t.view(bsz, T, n_head, hd) for t in (q, k, v))
    q, k = apply_rope(q, fc), apply_rope(k, fc)
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    cache[0][:, :, start:end] = k
    cache[1][:, :, start:end] = v
    keys, vals = cache[0][:, :, :end], cache[1][:, :, :end]
    q = q * hd ** -0.5
    att = pol.matmul(q, keys.transpose(-1, -2))
    if T > 1:
        att = att + mask
    att = torch.softmax(att.float() if pol.amp else att, dim=-1)
    out = pol.matmul(att, vals).transpose(1, 2).contiguous().view(bsz, T, dim)
    return pol.linear(out, w[pre + "wo.weight"])


def ffn(w: dict, pre: str, x, pol: Policy):
    h1, h2 = pol.linear(x, w[pre + "w1.weight"]).chunk(2, dim=-1)       # :187-189
    return pol.linear(F.silu(h1) * h2, w[pre + "w2.weight"])


def forward_model(w: dict, cfg: dict, x, mask, fc, caches, start: int, end: int, pol: Policy):
    """forward_model :342-350 = emb_norm -> forward_onestep per layer -> norm."""
    x = rms(x, w["emb_norm.weight"])
    for i in range(cfg["n_layer"]):
        p = f"layers.{i}."
        h = x + attention(w, p + "attention.", rms(x, w[p + "attention_norm.weight"]), mask, fc, caches[i], start, end,
                          cfg["n_head"], pol)
        x = h + ffn(w, p + "feed_forward.", rms(h, w[p + "ffn_norm.weight"]), pol)
    return rms(x, w["norm.weight"])


def proj_in(w: dict, x, pol: Policy):
    h1, h2 = pol.linear(x, w["proj_in.w1.weight"], w["proj_in.w1.bias"]).chunk(2, dim=-1)   # MLPConnector :73-75
    return pol.linear(F.silu(h1) * h2, w["proj_in.w2.weight"], w["proj_in.w2.bias"])


def cfg_at(cfg_scale: float, schedule: str, diff_pos: int, seq_len: int) -> float:
    """head_sample :356-365."""
    if cfg_scale > 1.0:
        if schedule == "constant":
            return cfg_scale
        if schedule == "linear":
            return 1.0 + (cfg_scale - 1.0) * diff_pos / seq_len
        raise NotImplementedError(schedule)
    return 1.0


def head_weights(w: dict) -> dict:
    return {k[len("head."):]: v for k, v in w.items() if k.startswith("head.")}


def unpatchify_raster(x: torch.Tensor, p: int, hw: tuple[int, int]) -> torch.Tensor:
    """utils.py:76-88: [B, N, C] in patch-raster order -> [B, C, H, W]."""
    B, N, C = x.shape
    H, W = hw
    return x.view(B, H // p, W // p, p, p, C).permute(0, 5, 1, 3, 2, 4).contiguous().view(B, C, H, W)


# ------------------------------------------------------------------------------------------ sample
def sample(w: dict, cfg: dict, class_ids: torch.Tensor, sample_steps: int, cfg_scale: float, noise,
           pol: Policy, cfg_schedule: str = "linear", force_tokens=None, trace: dict | None = None):
    """BitDance.sample :371-419 up to (not including) ``vae.decode``.

    ``noise``: iterator of the reference's randn / randn_like draws in call order.  ``force_tokens`` [bsz, hw, C]
    (teacher forcing): when given, each step's binarised prediction is replaced by these before it is fed back.
    Returns (latent [n, C, h, w] in {-1, 0, +1}, tokens [bsz, hw, C], preds [bsz, hw, C] pre-sign)."""
    P, D = cfg["parallel_num"], cfg["dim"]
    noise = iter(noise) if P > 1 else (n.reshape(-1, n.shape[-1]) for n in noise)     # 1x: the head samples rows [N, C]
    hw = cfg["resolution"] // (cfg["down_size"] * cfg["patch_size"])
    n_cls = cfg["cls_token_num"]
    total = hw * hw + n_cls
    if cfg_scale > 1.0:
        ids = torch.cat([class_ids, torch.ones_like(class_ids) * cfg["num_classes"]])
    else:
        ids = class_ids
    bsz = ids.shape[0]
    act = bsz // 2 if cfg_scale > 1.0 else bsz
    hd = D // cfg["n_head"]
    caches = [(torch.zeros(bsz, cfg["n_head"], total, hd), torch.zeros(bsz, cfg["n_head"], total, hd))
              for _ in range(cfg["n_layer"])]
    fc_all = rope_table(cfg)
    mask_all = block_causal_mask(hw * hw + n_cls - 1, n_cls - 1, P)[None, None]
    hwt = head_weights(w)
    c = F.embedding(ids, w["cls_embedding.weight"]).view(bsz, n_cls, -1)
    seq_len = hw * hw // P
    toks, preds = [], []
    last = None
    for i in range(seq_len):
        if i == 0:
            T0 = n_cls + P - 1
            x = torch.cat([c, w["query_token"].repeat(bsz, 1, 1)], dim=1) if P > 1 else c    # model.py:372-375: class tokens only
            x = forward_model(w, cfg, x, mask_all[:, :, :T0, :T0], fc_all[0:T0], caches, 0, T0, pol)
        else:
            x = proj_in(w, last, pol)
            s0 = P * (i - 1) + n_cls + P - 1
            x = forward_model(w, cfg, x, mask_all[:, :, s0:s0 + P, :s0 + P], fc_all[s0:s0 + P], caches, s0, s0 + P, pol)
        z = x[:, -P:, :] + w["pos_for_diff.weight"][i * P:(i + 1) * P, :]
        ci = cfg_at(cfg_scale, cfg_schedule, i, seq_len)
        if trace is not None:
            trace.setdefault("z", []).append(z.float().clone())
        if P == 1:
            z = z.reshape(-1, z.shape[-1])                                # model.py:332: the MLP head sees rows
        if diff_head.is_mlp_head(hwt):                  # 1x models: MLP head over rows (model.py:331-332: x.view(-1, D))
            fwd = lambda xx, tt, cc: diff_head.mlp_net_forward(hwt, xx, tt, cc, pol)
        else:
            fwd = lambda xx, tt, cc: diff_head.net_forward(hwt, xx, tt, cc, pol, final_sigmoid=False, head_dim=64)
        from . import sampler
        pred = sampler.euler_maruyama(hwt["net.input_proj.weight"].shape[1], fwd, z, ci, sample_steps, noise,
                                      time_shift=cfg.get("time_shift", 1.0))
        if P == 1:
            pred = pred.view(-1, 1, pred.shape[-1])                       # model.py:346
        preds.append(pred.clone())
        tok = torch.sign(pred)                                            # LFQ :367-368
        toks.append(tok)
        last = tok if force_tokens is None else force_tokens[:, i * P:(i + 1) * P].to(tok.dtype)
    tokens = torch.cat(toks, dim=-2)
    used = tokens if force_tokens is None else force_tokens.to(tokens.dtype)
    latent = unpatchify_raster(used[:act], int(P ** 0.5), (hw, hw))
    return latent, tokens, torch.cat(preds, dim=-2)
