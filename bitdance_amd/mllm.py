"""``MLLModel``-shaped entry to the native generation loop.

The reference has the same next-patch-diffusion loop twice: ``BitDanceT2IPipeline.gen_image`` (modeling/t2i_pipeline.py:
157-272, inference from a released model directory) and ``MLLModel.gen_image`` -> ``gen_image_block_causal``
(modeling/mllm.py:258-272,386-501, the training-side model object used by the evaluation scripts).  The golden
``tests/golden/mllm_equiv.npz`` pins that both produce identical tokens from identical components and noise, so one native
loop serves both surfaces; this class carries the attribute names code written against ``MLLModel`` reads
(``tokenizer``, ``llm_model.model``, ``vision_head``, ``embed_vision_mlp``, ``vision_encoder``, ``parallel_num``, ``ps``,
``hidden_size``, ``config.head.vision_pred``) and forwards ``gen_image`` / ``decode_image`` with the reference's signatures.

Out of scope here (training): ``forward`` / losses, ``forward_inference_block_causal``
(interleaved text + image), ``gen_image_full_causal`` (parallel_num == 1 models) -- they raise ``NotImplementedError``.
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

    def gen_image_full_causal(self, *a, **k):
        raise NotImplementedError("parallel_num == 1 models (token-by-token loop, mllm.py:274-384) are not part of the native path")

    def decode_image(self, image_latents, image_size=None, ps=1):
        """mllm.py:503-512."""
        return self._p.decode_image(image_latents, image_size, ps)

    def forward(self, *a, **k):
        raise NotImplementedError("training forward / losses are out of scope (SURVEY.md section 8: inference hot path only)")

    def forward_inference_block_causal(self, *a, **k):
        raise NotImplementedError("interleaved text + image inference (mllm.py:695-897) is a 'next' row (SURVEY.md section 8f rank 3)")
