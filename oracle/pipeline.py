"""Oracle: next-patch-diffusion AR loop -- test infrastructure only.

Restates /root/reference/modeling/t2i_pipeline.py:
  _get_1d_sincos_pos_embed :85-96     get_2d_embed :98-107
  gen_image                :157-272   decode_image (un-raster only) :274-283
and /root/reference/modeling/utils.py MLPconnector :9-20, and the image-generating part of
/root/reference/modeling/mllm.py ``forward_inference_block_causal`` :695-897 (interleaved text + image context,
``encode_image`` :899-930, ``remove_first_user_block`` utils.py:206-216) on top of the same loop.

Tokenisation is outside the arithmetic path: the loop takes token-id lists where the
reference calls ``tokenizer.encode`` / ``convert_tokens_to_ids`` (:175-194).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .numerics import F32, Policy
from . import diff_head, qwen3


def sincos_1d(dim: int, max_len: int) -> torch.Tensor:
    """:85-96 -> [max_len, dim] = cat(sin, cos)."""
    omega = torch.arange(dim // 2, dtype=F32)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = torch.einsum("m,d->md", torch.arange(max_len, dtype=F32), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def pos_embed_2d(table: torch.Tensor, h: int, w: int, ps: int) -> torch.Tensor:
    """:98-107 -> [h*w, D]; channel order (emb_h | emb_v), token order '(h w p1 p2)'."""
    half = table.shape[1]
    gv = table[:h].view(h, 1, half).expand(h, w, half)
    gh = table[:w].view(1, w, half).expand(h, w, half)
    pe = torch.cat([gh, gv], dim=-1)
    pe = pe.reshape(h // ps, ps, w // ps, ps, 2 * half).permute(0, 2, 1, 3, 4)
    return pe.reshape(h * w, 2 * half)


def projector(w: dict, x: torch.Tensor, pol: Policy) -> torch.Tensor:
    """MLPconnector.forward utils.py:16-20 with ``gelu_pytorch_tanh``."""
    h = pol.linear(x, w["fc1.weight"], w["fc1.bias"], quant=False)
    h = F.gelu(h, approximate="tanh")
    return pol.linear(h, w["fc2.weight"], w["fc2.bias"])


def unraster(tokens: torch.Tensor, h: int, w: int, ps: int) -> torch.Tensor:
    """decode_image :274-280: 'b (h w p1 p2) c -> b c (h p1) (w p2)'."""
    b, _, c = tokens.shape
    x = tokens.view(b, h // ps, w // ps, ps, ps, c).permute(0, 5, 1, 3, 2, 4)
    return x.reshape(b, c, h, w)


def gen_tokens(llm_w: dict, llm_cfg: dict, head_w: dict, proj_w: dict, embed: torch.Tensor,
               cond_ids, uncond_ids, start_ids, query_ids, *, h: int, w: int, parallel_num: int,
               guidance_scale: float, num_sampling_steps: int, num_images: int, noise, pol: Policy,
               max_patch: int = 256, trace: dict | None = None,
               force_tokens: torch.Tensor | None = None) -> torch.Tensor:
    """gen_image :157-270 up to (not including) the AE decode.  Returns [num_images, h*w, C] in {-1,0,+1}.

    embed      : the LLM input-embedding matrix [V, D]
    start_ids  : [<|vision_start|>, <|res_h|>, <|res_w|>]            (:181-184)
    query_ids  : [<|query_1|> .. <|query_{P-1}|>]                    (:190-194)
    noise      : iterator over the RNG draws in reference call order  (sampling_x.py:60,40)
    force_tokens: teacher forcing for tolerance tests -- [num_images, h*w, C] tokens fed back to the
                 LLM instead of the loop's own sign(pred) (the returned tokens are still the loop's own)
    """
    dev = embed.device
    tail = F.embedding(torch.tensor(list(start_ids) + list(query_ids), device=dev), embed)
    ctx = [torch.cat([F.embedding(torch.tensor(list(ids), device=dev), embed), tail], dim=0) if ids is not None else None
           for ids in (cond_ids, uncond_ids if guidance_scale > 1.0 else None)]
    return gen_tokens_from_context(llm_w, llm_cfg, head_w, proj_w, ctx[0], ctx[1], h=h, w=w, parallel_num=parallel_num,
                                   guidance_scale=guidance_scale, num_sampling_steps=num_sampling_steps, num_images=num_images,
                                   noise=noise, pol=pol, max_patch=max_patch, trace=trace, force_tokens=force_tokens)


def gen_tokens_from_context(llm_w: dict, llm_cfg: dict, head_w: dict, proj_w: dict, cond_ctx: torch.Tensor,
                            uncond_ctx: torch.Tensor | None, *, h: int, w: int, parallel_num: int, guidance_scale: float,
                            num_sampling_steps: int, num_images: int, noise, pol: Policy, max_patch: int = 256,
                            trace: dict | None = None, force_tokens: torch.Tensor | None = None) -> torch.Tensor:
    """The AR loop over an arbitrary context: ``cond_ctx`` / ``uncond_ctx`` [T, D] are the input EMBEDDINGS of everything
    before the first patch, query tokens included (t2i_pipeline.py:195-236 with text ids; mllm.py:745-805 with an interleaved
    text + image context).  Prefill = causal over ctx[:-P], all-visible over the last P; then :241-270 == mllm.py:806-864."""
    P = parallel_num
    ps = int(P ** 0.5)
    D = cond_ctx.shape[1]
    cfg_on = guidance_scale > 1.0
    noise = iter(noise)
    dev = cond_ctx.device
    pos = pos_embed_2d(sincos_1d(D // 2, max_patch), h, w, ps).unsqueeze(0).to(dev)  # fp32

    def prefill(x):
        x = x.unsqueeze(0).repeat(num_images, 1, 1)
        _, cache = qwen3.model_forward(llm_w, llm_cfg, x[:, :-P], None, None, pol)
        past = cache[0][0].shape[2]
        ones = torch.ones(num_images, 1, P, P + past, dtype=torch.bool, device=dev)
        hid, cache = qwen3.model_forward(llm_w, llm_cfg, x[:, -P:], cache, ones, pol)
        return hid[:, -P:], cache

    hid_c, cache_c = prefill(cond_ctx)
    if cfg_on:
        hid_u, cache_u = prefill(uncond_ctx)
    out = []
    for step in range((h * w) // P):
        sl = slice(step * P, (step + 1) * P)
        hf = torch.cat([hid_c, hid_u], dim=0) if cfg_on else hid_c
        hf = hf + pos[:, sl]
        pred = diff_head.sample(head_w, hf, guidance_scale, num_sampling_steps, noise, pol)
        tok = torch.sign(pred)                                                     # :248 (sign(0)=0)
        if trace is not None:
            trace.setdefault("pred", []).append(pred[:num_images].clone())
            trace.setdefault("cond", []).append(hf.clone())
        out.append(tok[:num_images])
        if force_tokens is not None:
            tok = torch.cat([force_tokens[:, sl].to(dev)] * (2 if cfg_on else 1), dim=0)
        x = projector(proj_w, tok, pol) + pos[:, sl]
        ones = torch.ones(x.shape[0], 1, P, P + cache_c[0][0].shape[2], dtype=torch.bool, device=dev)
        hid_c, cache_c = qwen3.model_forward(llm_w, llm_cfg, x[:num_images], cache_c, ones[:num_images], pol)
        hid_c = hid_c[:, -P:]
        if cfg_on:
            hid_u, cache_u = qwen3.model_forward(llm_w, llm_cfg, x[num_images:], cache_u, ones[num_images:], pol)
            hid_u = hid_u[:, -P:]
    return torch.cat(out, dim=1)


# ------------------------------------------------------------------------------------------------ interleaved context
def remove_first_user_block(x: str) -> str:
    """modeling/utils.py:206-216: the unconditional branch's text = the text without its first user turn."""
    a, b = "<|im_start|>user\n", "<|im_end|>\n"
    i = x.find(a)
    if i == -1:
        return x
    j = x.find(b, i + len(a))
    return x if j == -1 else x[:i] + x[j + len(b):]


def image_latents_to_tokens(quant: torch.Tensor, ps: int) -> torch.Tensor:
    """VQModel.vt_forward's re-ordering (vision_encoder/autoencoder.py:418-422): [C, h, w] -> '(h w p1 p2) c'."""
    C, H, W = quant.shape
    return quant.view(C, H // ps, ps, W // ps, ps).permute(1, 3, 2, 4, 0).reshape(H * W, C)


def encode_image(proj_w: dict, latents: torch.Tensor, hw: tuple[int, int], D: int, ps: int, pol: Policy,
                 max_patch: int = 256) -> torch.Tensor:
    """MLLModel.encode_image (mllm.py:899-930) after the tokenizer: ``latents`` [h*w, C] binary tokens in patch order ->
    embed_vision_mlp -> += get_2d_embed(h, w, ps) (an in-place add: the sum keeps the projector's dtype)."""
    e = projector(proj_w, latents, pol)
    pe = pos_embed_2d(sincos_1d(D // 2, max_patch), hw[0], hw[1], ps)
    return (e.float() + pe).to(e.dtype)                   # `x += pos` on a bf16 x: fp32 sum, one rounding back to bf16


def interleaved_context(embed: torch.Tensor, plan: list, texts: list, image_embeds: list, encode, *, start_of_image: int,
                        end_of_image: int, res_ids: tuple[int, int], query_ids, cfg_on: bool):
    """The context a model-generated IMAGE item sees in ``forward_inference_block_causal`` (mllm.py:719-745,865-895): per plan
    item, user text -> token embeddings (uncond: the text minus its first user block); every image item -> the
    [start_of_image, res_h, res_w] embeddings (res ids from the GENERATION size, :730-733); a user image -> its encode_image
    embeddings + end_of_image; the generated image -> the query tokens.  Returns (cond [T, D], uncond [T', D] or None) up to
    and including the first model-generated image's query tokens."""
    E = lambda ids: F.embedding(torch.tensor(list(ids), dtype=torch.long), embed)
    texts, image_embeds = list(texts), list(image_embeds)
    c, u = [], []
    for item in plan:
        if item["type"] == "image":
            s = E([start_of_image, *res_ids])
            c.append(s)
            u.append(s)
        if item["from"] == "model":
            if item["type"] != "image":
                raise NotImplementedError("text generation")
            q = E(query_ids)
            c.append(q)
            u.append(q)
            break
        if item["type"] == "text":
            t = texts.pop(0)
            c.append(E(encode(t)))
            if cfg_on:
                u.append(E(encode(remove_first_user_block(t))))
        else:
            e = image_embeds.pop(0)
            end = E([end_of_image])
            c += [e, end]
            if cfg_on:
                u += [e, end]
    cat = lambda xs: torch.cat([x.to(xs[-1].dtype) if x.dtype != xs[-1].dtype else x for x in xs], dim=0)
    return cat(c), (cat(u) if cfg_on else None)


# ------------------------------------------------------------------------------------------------ token sampler
def filter_logits(logits: torch.Tensor, top_k: int = 0, top_p: float = 1.0, min_tokens_to_keep: int = 1) -> torch.Tensor:
    """modeling/utils.py:64-91 (top_k_top_p_filtering) as set arithmetic, row by row with numpy-style loops (small cases):
    top-k keeps every logit >= the k-th largest (ties survive); top-p then keeps, in descending order, the shortest prefix
    whose softmax mass exceeds top_p (the crossing token included, at least ``min_tokens_to_keep``).  Removed entries = -inf."""
    out = logits.clone().to(F32)
    V = out.shape[-1]
    for r in range(out.shape[0]):
        row = out[r]
        if top_k > 0:
            k = min(max(top_k, min_tokens_to_keep), V)
            kth = torch.sort(row, descending=True)[0][k - 1]
            row[row < kth] = float("-inf")
        if top_p < 1.0:
            vals, idx = torch.sort(row, descending=True)
            cum = torch.cumsum(torch.softmax(vals, dim=-1), dim=-1)
            drop = [False] * V
            for j in range(1, V):                                  # token j goes when the mass BEFORE it already exceeds top_p
                drop[j] = bool(cum[j - 1] > top_p)
            if min_tokens_to_keep > 1:
                for j in range(min(min_tokens_to_keep, V)):
                    drop[j] = False
                # the reference clears positions [0, min_keep) BEFORE shifting by one: position min_keep is then also kept
                if min_tokens_to_keep < V:
                    drop[min_tokens_to_keep] = False
            for j in range(V):
                if drop[j]:
                    row[idx[j]] = float("-inf")
    return out


def sample_codebook_greedy(pred_logits: torch.Tensor, codebook: torch.Tensor, temperature: float, top_k: int, top_p: float):
    """modeling/utils.py:94-124 with do_sample=False: argmax of softmax(filtered logits / max(T, 1e-5)) and its embedding."""
    lg = pred_logits / max(temperature, 1e-5)
    if top_k > 0 or top_p < 1.0:
        lg = filter_logits(lg, top_k, top_p)
    tok = torch.argmax(torch.softmax(lg, dim=-1), dim=-1)
    return tok, codebook[tok]
