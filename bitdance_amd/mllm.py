"""``MLLModel``-shaped entry to the native generation loop.

The reference has the same next-patch-diffusion loop twice: ``BitDanceT2IPipeline.gen_image`` (modeling/t2i_pipeline.py:
157-272, inference from a released model directory) and ``MLLModel.gen_image`` -> ``gen_image_block_causal``
(modeling/mllm.py:258-272,386-501, the training-side model object used by the evaluation scripts).  The golden
``tests/golden/mllm_equiv.npz`` pins that both produce identical tokens from identical components and noise, so one native
loop serves both surfaces; this class carries the attribute names code written against ``MLLModel`` reads
(``tokenizer``, ``llm_model.model``, ``vision_head``, ``embed_vision_mlp``, ``vision_encoder``, ``parallel_num``, ``ps``,
``hidden_size``, ``config.head.vision_pred``) and forwards ``gen_image`` / ``decode_image`` with the reference's signatures.

``forward_inference_block_causal`` (mllm.py:695-897) is served for the plans the reference itself
can run: any user text / user image items followed by ONE model-generated image (text-to-image, image editing, multi-image
conditioning).  Its text-generation branch does not run in the reference (the first decode step indexes ``past_key_values``
while it is still None, :796-800; later steps would feed 2-D embeddings), so a plan that asks the model for text raises
``NotImplementedError`` here too; the token sampler that branch would call (``sample_codebook`` / ``top_k_top_p_filtering``,
modeling/utils.py:64-124) is provided and pinned against the reference's outputs all the same.  ``encode_image`` (:899-930) = the tokenizer's conv encoder (native kernels under bf16 autocast: ae_native.NativeEncoder) -> binary tokens in patch
order -> the native projector -> + 2-D position embedding.

``gen_image_full_causal`` (mllm.py:274-384, parallel_num == 1 models: one token per AR step) is the same native loop at P = 1
(golden ``full_causal_*.npz``).  Out of scope (training): ``forward`` / losses.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .t2i_pipeline import BitDanceT2IPipeline


class MLLModel:
    def __init__(self, pipeline: BitDanceT2IPipeline):
        p = self._p = pipeline
        self.device = p.device
        self.tokenizer = p.tokenizer
        self.llm_config = p.llm_config
        self.llm_model = p.llm_model                     # .model(...) = the native Qwen3 seam (seams.NativeQwen3Model)
        self.vision_head = p.vision_head                 # .sample(z, cfg, num_sampling_steps)
        self.embed_vision_mlp = p.embed_vision_mlp
        self.vision_encoder = p.ae                       # the reference's name for the VQModel (mllm.py:62-66)
        self.vision_latent_dim = p.ae_config["ddconfig"]["z_channels"]
        self.vision_head_type = "diffusion_parallel_x"
        self.hidden_size = p.hidden_size
        self.parallel_num = p.parallel_num
        self.ps = p.ps
        self.vae_patch_size = p.vae_patch_size
        self.config = SimpleNamespace(head=SimpleNamespace(vision_pred=dict(p.vision_head_config, type=self.vision_head_type)),
                                      vit_patch_size=p.vae_patch_size)

    @classmethod
    def from_pretrained(cls, model_path: str, device="cuda", tp=None) -> "MLLModel":
        return cls(BitDanceT2IPipeline(model_path, device=device, tp=tp))

    def get_2d_embed(self, h, w, ps=1):
        return self._p.get_2d_embed(h, w, ps)

    @torch.no_grad()
    def gen_image(self, cond_prompt, uncond_prompt=None, guidance_scale: float = 1.0, num_sampling_steps: int = 50,
                  max_length: int = 64, num_images: int = 1, image_size=[256, 256], show_progress: bool = False):
        """mllm.py:258-272: parallel_num > 1 -> the block-causal loop."""
        if self.parallel_num > 1:
            return self.gen_image_block_causal(cond_prompt, uncond_prompt, guidance_scale, num_sampling_steps, max_length,
                                               num_images, image_size, show_progress)
        return self.gen_image_full_causal(cond_prompt, uncond_prompt, guidance_scale, num_sampling_steps, max_length,
                                          num_images, image_size, show_progress)

    @torch.no_grad()
    def gen_image_block_causal(self, cond_prompt, uncond_prompt=None, guidance_scale: float = 1.0, num_sampling_steps: int = 50,
                               max_length: int = 64, num_images: int = 1, image_size=[256, 256], show_progress: bool = False,
                               **native_kw):
        """mllm.py:386-501 == t2i_pipeline.py:157-272 (pinned by tests/golden/mllm_equiv.npz): the native AR loop."""
        return self._p.gen_image(cond_prompt, uncond_prompt, guidance_scale, num_sampling_steps, max_length, num_images,
                                 image_size, show_progress, **native_kw)

    @torch.no_grad()
    def gen_image_full_causal(self, cond_prompt, uncond_prompt=None, guidance_scale: float = 1.0, num_sampling_steps: int = 50,
                              max_length: int = 64, num_images: int = 1, image_size=[256, 256], show_progress: bool = False,
                              **native_kw):
        """mllm.py:274-384, what ``gen_image`` dispatches to when the head's parallel_num is 1 (:268-272): one token per AR step
        (``step_width`` = parallel_num), a plain causal prefill, no query tokens, ps = 1.  That is the block-causal loop with blocks
        of one token -- the reference's two loops produce identical tokens from identical components and noise at
        parallel_num = 1 (asserted when tests/golden/full_causal_*.npz are generated) -- so the same native loop serves it.  The
        reference can only run it with the diffusion_parallel_x head (the one head that builds ``vision_diffusion_head``,
        :133-150); with parallel_num > 1 its ``last_hidden_state[:, -step_width:]`` / query-token variant is
        gen_image_block_causal's prefill without the block mask, which no released model uses."""
        if self.parallel_num != 1:
            raise NotImplementedError("gen_image_full_causal is the loop of parallel_num == 1 models (mllm.py:268-272); "
                                      f"this model has parallel_num = {self.parallel_num}: use gen_image / gen_image_block_causal")
        return self._p.gen_image(cond_prompt, uncond_prompt, guidance_scale, num_sampling_steps, max_length, num_images,
                                 image_size, show_progress, **native_kw)

    def decode_image(self, image_latents, image_size=None, ps=1):
        """mllm.py:503-512."""
        return self._p.decode_image(image_latents, image_size, ps)

    def forward(self, *a, **k):
        raise NotImplementedError("training forward / losses are out of scope (SURVEY.md section 8: inference hot path only)")

    # ------------------------------------------------------------------ interleaved text + image context
    @torch.no_grad()
    def vt_forward(self, image_list, max_bs: int = 32, ps: int = 1) -> torch.Tensor:
        """VQModel.vt_forward (vision_encoder/autoencoder.py:402-424): images grouped by size, encoded in batches of ``max_bs``,
        each latent [C, h, w] re-ordered 'c (h p1) (w p2) -> (h w p1 p2) c'; concatenated in list order."""
        groups: dict = {}
        for i, img in enumerate(image_list):
            groups.setdefault(tuple(img.shape[-2:]), []).append((i, img))
        out = [None] * len(image_list)
        for items in groups.values():
            for s0 in range(0, len(items), max_bs):
                chunk = items[s0:s0 + max_bs]
                quant = self.vision_encoder.encode(torch.cat([x[1] for x in chunk], dim=0).to(self.device))
                for b, (idx, _) in enumerate(chunk):
                    C, H, W = quant[b].shape
                    out[idx] = quant[b].view(C, H // ps, ps, W // ps, ps).permute(1, 3, 2, 4, 0).reshape(H * W, C)
        return torch.cat(out, dim=0)

    @torch.no_grad()
    def encode_image(self, image_list, packed_label_indexes_vision=None):
        """mllm.py:899-930 (inference branch): -> (packed embeddings [sum h_i*w_i, D], packed binary latents [sum h_i*w_i, C])."""
        lat = self.vt_forward(image_list, max_bs=32, ps=self.ps)
        emb = self.embed_vision_mlp(lat)                                           # native projector, bf16 values
        pos = torch.cat([self.get_2d_embed(img.shape[-2] // self.vae_patch_size, img.shape[-1] // self.vae_patch_size, ps=self.ps)
                         for img in image_list], dim=0)
        emb = (emb.float() + pos.to(emb.device)).to(emb.dtype)                     # `+=` on the bf16 tensor: one rounding
        return emb, lat.clone()

    @staticmethod
    def remove_first_user_block(x: str) -> str:
        """modeling/utils.py:206-216."""
        a, b = "<|im_start|>user\n", "<|im_end|>\n"
        i = x.find(a)
        if i == -1:
            return x
        j = x.find(b, i + len(a))
        return x if j == -1 else x[:i] + x[j + len(b):]

    @staticmethod
    def top_k_top_p_filtering(logits: torch.Tensor, top_k: int = 0, top_p: float = 1.0, filter_value: float = -float("inf"),
                              min_tokens_to_keep: int = 1) -> torch.Tensor:
        """modeling/utils.py:64-91, vectorised over rows: keep the logits >= the k-th largest, then the shortest descending
        prefix whose softmax mass exceeds ``top_p`` (crossing token included); everything else becomes ``filter_value``."""
        V = logits.size(-1)
        if top_k > 0:
            kth = torch.topk(logits, min(max(top_k, min_tokens_to_keep), V), dim=-1).values[..., -1:]
            logits = logits.masked_fill(logits < kth, filter_value)
        if top_p < 1.0:
            vals, order = torch.sort(logits, descending=True, dim=-1)
            cum = torch.cumsum(torch.softmax(vals, dim=-1), dim=-1)
            gone = torch.zeros_like(vals, dtype=torch.bool)
            gone[..., 1:] = cum[..., :-1] > top_p                        # the reference's shift-right of (cum > top_p)
            if min_tokens_to_keep > 1:
                gone[..., : min_tokens_to_keep + 1] = False               # cleared before the shift there: one more survives
            logits = logits.masked_fill(torch.zeros_like(gone).scatter(-1, order, gone), filter_value)
        return logits

    @staticmethod
    def sample_codebook(pred_logits, cur_item_type, codebook, do_sample: bool = True, temperature: float = 1.0, top_k: int = 0,
                        top_p: float = 1.0):
        """modeling/utils.py:94-124: temperature, top-k / top-p, softmax, multinomial (or argmax) -> (tokens, codebook(tokens)).
        Host-side plumbing (one row per sequence over the vocabulary); the reference reaches it only from the text / standard-head
        branches of its interleaved loops."""
        logits = pred_logits / max(temperature, 1e-5)
        if top_k > 0 or top_p < 1.0:
            logits = MLLModel.top_k_top_p_filtering(logits, top_k=top_k, top_p=top_p)
        probs = torch.softmax(logits, dim=-1)
        tokens = torch.multinomial(probs, num_samples=1).squeeze(-1) if do_sample else torch.argmax(probs, dim=-1)
        return tokens, codebook(tokens)

    def _tok_id(self, name: str) -> int:
        t = self.tokenizer
        return getattr(t, name + "_id") if hasattr(t, name + "_id") else t.convert_tokens_to_ids(
            {"start_of_image": "<|vision_start|>", "end_of_image": "<|vision_end|>"}.get(name, f"<|{name}|>"))

    @torch.no_grad()
    def forward_inference_block_causal(self, sequence_plan, text_list, image_list, do_sample: bool = True,
                                       max_length_text: int = 128, max_length_vision: int = 64, temperature: float = 1.0,
                                       sample_steps: int = 50, image_size=[256, 256], cfg_scale=7.5, *args, noise=None,
                                       return_tokens: bool = False, force_tokens=None, trace=None, **kwargs):
        """mllm.py:695-897 for plans that end in ONE model-generated image (see the module docstring).  The context is
        assembled exactly as the reference does (:719-745,865-895): user text -> token embeddings (unconditional branch: the
        text without its first user block); every image item -> [start_of_image, res_h, res_w] with the res tokens of the
        GENERATION size; a user image -> ``encode_image`` + end_of_image; then the query tokens and the native AR loop.
        Returns {"generated_text": [], "generated_image": [image]} (``return_tokens``: the binary tokens instead)."""
        plan = list(sequence_plan)
        model_items = [i for i, it in enumerate(plan) if it["from"] == "model"]
        if model_items != [len(plan) - 1] or plan[-1]["type"] != "image":
            raise NotImplementedError("native forward_inference_block_causal: user text / image items followed by one model-generated "
                                      "image (the reference's text-generation branch does not run: mllm.py:796-800)")
        texts, images = list(text_list), list(image_list)
        use_cfg = cfg_scale > 1.0
        P, vp = self.parallel_num, self.vae_patch_size
        if max_length_vision != (image_size[0] // vp) * (image_size[1] // vp):
            # the reference stops after max_length_vision tokens and then un-rasters them as a square image (:806,861)
            raise ValueError(f"max_length_vision={max_length_vision} must equal the token count of image_size {image_size}: "
                             f"{(image_size[0] // vp) * (image_size[1] // vp)}")
        embed = self._p.llm_w.sd["model.embed_tokens.weight"]
        E = lambda ids: torch.nn.functional.embedding(torch.tensor(list(ids), device=self.device, dtype=torch.long), embed)
        start = E([self._tok_id("start_of_image"), self._tok_id(f"res_{image_size[0] // vp}"), self._tok_id(f"res_{image_size[1] // vp}")])
        c, u = [], []
        for item in plan:
            if item["type"] == "image":
                c.append(start)
                u.append(start)
            if item["from"] == "model":
                q = E([self._tok_id(f"query_{i}") for i in range(1, P)])
                c.append(q)
                u.append(q)
            elif item["type"] == "text":
                t = texts.pop(0)
                c.append(E(self.tokenizer.encode(t)))
                if use_cfg:
                    u.append(E(self.tokenizer.encode(self.remove_first_user_block(t))))
            else:
                e = self.encode_image([images.pop(0).to(self.device)])[0]
                end = E([self._tok_id("end_of_image")])
                c += [e, end]
                if use_cfg:
                    u += [e, end]
        out = self._p.gen_image_from_context(torch.cat([x.to(torch.bfloat16) for x in c]),
                                             torch.cat([x.to(torch.bfloat16) for x in u]) if use_cfg else None,
                                             guidance_scale=cfg_scale, num_sampling_steps=sample_steps, num_images=1,
                                             image_size=image_size, noise=noise, return_tokens=return_tokens,
                                             force_tokens=force_tokens, trace=trace)
        return {"generated_text": [], "generated_image": [out]}
