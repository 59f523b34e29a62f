"""Class-conditional ImageNet BitDance (parallel "16x" variant) on the native diffusion head.

Mirrors the inference surface of ``imagenet_gen/src/model_parallel.py``:
``BitDance(...).sample(cond, sample_steps, cfg_scale, cfg_schedule)`` (:371-419) with ``head_sample``'s linear CFG ramp
(:352-369), the un-mixed first AR step (cfg_iter == 1.0 there, so cond and uncond rows are sampled independently),
LFQ ``sign`` and the patch-raster un-patchify.  State-dict keys are the reference's (minus ``vae.*``).

What runs where (round 1):
  * I3 ``diff_head_parallel.TransEncoder`` + ``sampling_parallel.euler_maruyama`` (>= 70 % of the model's FLOPs): the HIP
    head -- the T2I kernels with head_dim 64 attention and no final sigmoid (engine.HeadWeights(head_dim=64,
    final_sigmoid=False));
  * I2 the KV-cached block-causal transformer (4 % of the FLOPs): torch ops under bf16 autocast, i.e. the reference's own
    arithmetic on hipBLASLt; its native kernels (head_dim-64 attention, interleaved 2-D RoPE) are the next row;
  * the conv decoder stays on MIOpen (north star).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .engine import Engine, HeadWeights

__all__ = ["BitDance"]


def _pos_2d(resolution: int, patch: int) -> torch.Tensor:
    n = resolution // patch                                   # layers_parallel.py:235-252, one scale: cell centres
    c = torch.arange(n, dtype=torch.float32) + 0.5
    gy, gx = torch.meshgrid(c, c, indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=1)


def _patch_raster(x: torch.Tensor, p: int, H: int, W: int) -> torch.Tensor:
    """Rows of x in (H W) order -> (H/p W/p p p) order (utils.py:91-113)."""
    tail = x.shape[1:]
    return x.reshape(H // p, p, W // p, p, -1).permute(0, 2, 1, 3, 4).reshape(H * W, *tail)


class BitDance:
    def __init__(self, state_dict: dict, *, dim: int, n_layer: int, n_head: int, latent_dim: int, resolution: int = 256,
                 down_size: int = 16, patch_size: int = 1, cls_token_num: int = 64, num_classes: int = 1000,
                 parallel_num: int = 16, time_shift: float = 1.0, device="cuda", vae=None, **_unused):
        if not torch.cuda.is_available():
            raise RuntimeError("bitdance_amd.imagenet.BitDance needs a GPU (HIP head); there is no CPU fallback")
        if parallel_num != 16:
            raise NotImplementedError("native imagenet path: parallel_num must be 16 (the 16x checkpoints)")
        if time_shift != 1.0:
            raise NotImplementedError("time_shift != 1 is not wired into the native sampler schedule")
        self.device = torch.device(device)
        self.dim, self.n_layer, self.n_head = dim, n_layer, n_head
        self.P, self.cls_token_num, self.num_classes = parallel_num, cls_token_num, num_classes
        self.h = self.w = resolution // (down_size * patch_size)
        self.latent_dim = latent_dim
        self.total_tokens = self.h * self.w + cls_token_num
        self.vae = vae
        sd = {k: v.detach().to(self.device, torch.float32) for k, v in state_dict.items() if not k.startswith("vae.")}
        self.w_ = sd
        head_sd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
        self.head_w = HeadWeights.from_state_dict(head_sd, self.device, head_dim=64, final_sigmoid=False)
        self._eng: dict = {}
        # RoPE table [cls + P-1 + h*w - P, hd/2, 2] and block-causal mask (model_parallel.py:197-215)
        hd = dim // n_head
        half = hd // 2
        freqs = 1.0 / (10000 ** (torch.arange(0, half, 2)[: half // 2].float() / half))
        t = torch.cat([torch.zeros(cls_token_num + parallel_num - 1, 2), _pos_2d(resolution, down_size * patch_size) + 1.0])
        fr = torch.outer(t.flatten(), freqs).view(t.shape[0], -1)
        fc = torch.stack([torch.cos(fr), torch.sin(fr)], dim=-1)
        n_img = self.h * self.w
        fc[-n_img:] = _patch_raster(fc[-n_img:], int(parallel_num ** 0.5), self.h, self.w)
        self.freqs_cis = fc[:-parallel_num].to(self.device)
        tot, causal = n_img + cls_token_num - 1, cls_token_num - 1
        m = torch.zeros(tot, tot)
        m.masked_fill_(torch.triu(torch.ones(tot, tot), diagonal=1).bool(), float("-inf"))
        for i in range(causal, tot, parallel_num):
            m[i:i + parallel_num, i:i + parallel_num] = 0
        self.attn_mask = m[None, None].to(self.device)

    # ------------------------------------------------------------------ transformer (torch, reference arithmetic)
    def _rope(self, x, fc):
        xs = x.float().reshape(*x.shape[:-1], -1, 2)
        fc = fc.view(1, xs.size(1), 1, xs.size(3), 2)
        out = torch.stack([xs[..., 0] * fc[..., 0] - xs[..., 1] * fc[..., 1],
                           xs[..., 1] * fc[..., 0] + xs[..., 0] * fc[..., 1]], dim=-1)
        return out.flatten(3).type_as(x)

    def _forward_model(self, x, mask, start, end, caches):
        w, H = self.w_, self.n_head
        rms = lambda t, k: F.rms_norm(t, (t.shape[-1],), w[k], 1e-6)
        fc = self.freqs_cis[start:end]
        x = rms(x, "emb_norm.weight")
        for i in range(self.n_layer):
            p = f"layers.{i}."
            a = rms(x, p + "attention_norm.weight")
            B, T, D = a.shape
            q, k, v = F.linear(a, w[p + "attention.wqkv.weight"]).chunk(3, dim=-1)
            q, k, v = (t.view(B, T, H, D // H) for t in (q, k, v))
            q, k = self._rope(q, fc), self._rope(k, fc)
            q, k, v = (t.transpose(1, 2) for t in (q, k, v))
            kc, vc = caches[i]
            kc[:, :, start:end] = k
            vc[:, :, start:end] = v
            att = (q * (D // H) ** -0.5) @ kc[:, :, :end].transpose(-1, -2)
            if T > 1:
                att = att + mask
            out = (torch.softmax(att, dim=-1) @ vc[:, :, :end]).transpose(1, 2).contiguous().view(B, T, D)
            h = x + F.linear(out, w[p + "attention.wo.weight"])
            h1, h2 = F.linear(rms(h, p + "ffn_norm.weight"), w[p + "feed_forward.w1.weight"]).chunk(2, dim=-1)
            x = h + F.linear(F.silu(h1) * h2, w[p + "feed_forward.w2.weight"])
        return rms(x, "norm.weight")

    def _proj_in(self, x):
        w = self.w_
        h1, h2 = F.linear(x, w["proj_in.w1.weight"], w["proj_in.w1.bias"]).chunk(2, dim=-1)
        return F.linear(F.silu(h1) * h2, w["proj_in.w2.weight"], w["proj_in.w2.bias"])

    # ------------------------------------------------------------------ head (HIP)
    def _head_sample(self, z: torch.Tensor, cfg: float, steps: int, noise=None) -> torch.Tensor:
        """DiffHead.sample on the native head.  z [rows, P, D] fp32; returns [rows, P, C] like the reference
        (the CFG-mixed sample repeated for both halves when cfg > 1)."""
        mult = 2 if cfg > 1.0 else 1
        rows = z.shape[0]
        B = rows // mult
        key = (B, mult)
        if key not in self._eng:
            self._eng[key] = Engine(self.head_w, None, None, num_images=B, branches=mult, device=self.device,
                                    max_tokens=self.P, parallel_num=self.P)
        eng = self._eng[key]
        eng.set_schedule(steps, cfg, 1)
        if noise is None:
            eng.draw_noise(1)                                # randn + N x randn_like: the reference's RNG order
        else:
            eng.load_noise(noise.view(1, steps + 1, B, self.P, -1))
        eng.reset([0] * rows)
        eng.set_cond(z)
        eng.head_sample()
        x = eng.pred().clone()
        return torch.cat([x] * mult, dim=0)

    # ------------------------------------------------------------------ BitDance.sample
    @torch.no_grad()
    def sample(self, cond: torch.Tensor, sample_steps: int, cfg_scale: float = 1.0, cfg_schedule: str = "linear",
               chunk_size: int = 0, *, noise=None, force_tokens=None, return_tokens: bool = False):
        """model_parallel.py:371-419.  ``noise`` (tests): list of per-AR-step tensors [N+1, rows_i, P, C] replacing the
        RNG draws; ``force_tokens`` [bsz, h*w, C] teacher-forces the fed-back tokens; ``return_tokens`` returns
        (latent, tokens, preds) instead of decoding."""
        dev, P, n_cls = self.device, self.P, self.cls_token_num
        cond = cond.to(dev)
        ids = torch.cat([cond, torch.ones_like(cond) * self.num_classes]) if cfg_scale > 1.0 else cond
        bsz = ids.shape[0]
        act = bsz // 2 if cfg_scale > 1.0 else bsz
        hd = self.dim // self.n_head
        caches = [(torch.zeros(bsz, self.n_head, self.total_tokens, hd, device=dev),
                   torch.zeros(bsz, self.n_head, self.total_tokens, hd, device=dev)) for _ in range(self.n_layer)]
        seq_len = self.h * self.w // P
        w = self.w_
        toks, preds, last = [], [], None
        for i in range(seq_len):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if i == 0:
                    T0 = n_cls + P - 1
                    c = F.embedding(ids, w["cls_embedding.weight"]).view(bsz, n_cls, -1)
                    x = torch.cat([c, w["query_token"].repeat(bsz, 1, 1)], dim=1)
                    x = self._forward_model(x, self.attn_mask[:, :, :T0, :T0], 0, T0, caches)
                else:
                    s0 = P * (i - 1) + n_cls + P - 1
                    x = self._forward_model(self._proj_in(last), self.attn_mask[:, :, s0:s0 + P, :s0 + P], s0, s0 + P, caches)
                z = x[:, -P:, :] + w["pos_for_diff.weight"][i * P:(i + 1) * P, :]
            if cfg_scale > 1.0:
                if cfg_schedule == "constant":
                    ci = cfg_scale
                elif cfg_schedule == "linear":
                    ci = 1.0 + (cfg_scale - 1.0) * i / seq_len
                else:
                    raise NotImplementedError(f"unknown cfg_schedule {cfg_schedule}")
            else:
                ci = 1.0
            pred = self._head_sample(z.float(), ci, sample_steps, None if noise is None else noise[i].to(dev))
            preds.append(pred)
            tok = torch.sign(pred)
            toks.append(tok)
            last = tok if force_tokens is None else force_tokens[:, i * P:(i + 1) * P].to(dev, tok.dtype)
        tokens = torch.cat(toks, dim=-2)
        used = tokens if force_tokens is None else force_tokens.to(dev, tokens.dtype)
        p = int(P ** 0.5)
        C = used.shape[-1]
        latent = used[:act].view(act, self.h // p, self.w // p, p, p, C).permute(0, 5, 1, 3, 2, 4).contiguous() \
            .view(act, C, self.h, self.w)                      # unpatchify_raster (utils.py:76-88)
        if return_tokens:
            return latent, tokens, torch.cat(preds, dim=-2)
        if self.vae is None:
            return latent
        if chunk_size > 0:
            return torch.cat([self.vae.decode(latent[j:j + chunk_size]).cpu() for j in range(0, act, chunk_size)], dim=0)
        return self.vae.decode(latent)
