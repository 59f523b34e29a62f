"""The reference's internal operator seams, backed by the native engine:

  (1) ``pipe.llm_model.model(inputs_embeds=[B,T,D], past_key_values=cache|None, use_cache=True,
        attention_mask=None|bool[B,1,T,T+past]) -> obj.last_hidden_state, obj.past_key_values``  and
      ``pkv[0][0].shape[2]`` for the past length                         (t2i_pipeline.py:199-217,257-268)
  (2) ``pipe.vision_head.sample(z=[cfg*B,P,D], cfg=float, num_sampling_steps=int) -> [cfg*B,P,C] fp32``   (:246)
  (3) ``pipe.embed_vision_mlp(tokens [.,P,C]) -> [.,P,D]``                                                   (:249)
  (5) ``pipe.llm_model.model.embed_tokens(ids)``

``gen_image`` itself does not go through these objects (it replays the fused AR-step graphs); they exist so that code
written against the reference's attributes keeps working, one operator at a time.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .engine import Engine
from .llm import native_block, prefill_block


class NativeDiffHead:
    """DiffHead.sample (flow_head_parallel_x.py:107-120) on the HIP head."""

    def __init__(self, pipe):
        self._p = pipe
        self._eng: dict = {}

    @torch.no_grad()
    def sample(self, z: torch.Tensor, cfg: float, num_sampling_steps: int) -> torch.Tensor:
        pipe = self._p
        mult = 2 if cfg > 1.0 else 1
        rows, P, _ = z.shape
        B = rows // mult
        key = (B, mult, P)
        if key not in self._eng:
            self._eng[key] = Engine(pipe.head_w, None, None, num_images=B, branches=mult, device=pipe.device,
                                    max_tokens=P, parallel_num=P, comm=getattr(pipe, "tp", None))
        eng = self._eng[key]
        eng.set_schedule(num_sampling_steps, cfg, 1, time_shift=float(pipe.vision_head_config.get("time_shift", 1.0)))
        eng.draw_noise(1)                                   # randn + N x randn_like, the reference's RNG order
        eng.reset([0] * min(B * mult, 16))                  # the head reads only the step counter
        eng.set_cond(z.to(pipe.device))
        eng.head_sample()
        x = eng.pred().clone()
        return torch.cat([x] * mult, dim=0)


class NativeConnector:
    """MLPconnector.forward (modeling/utils.py:16-20): fc1 -> gelu(tanh) -> fc2, bf16 output."""

    def __init__(self, pipe):
        self._p = pipe
        self._eng: dict = {}

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        pipe = self._p
        lead = tokens.shape[:-1]
        t = tokens.reshape(-1, tokens.shape[-1]).to(pipe.device, torch.float32).contiguous()
        n = t.shape[0]
        P = pipe.parallel_num
        if n % P:
            raise NotImplementedError(f"native connector: token count must be a multiple of parallel_num={P}")
        B = n // P
        if B not in self._eng:
            eng = Engine(None, pipe.proj_w, None, num_images=B, branches=1, device=pipe.device, max_tokens=P,
                         parallel_num=P)
            eng._R = torch.zeros(eng.Mpad, pipe.proj_w.D, dtype=torch.float32, device=pipe.device)
            eng.set_ptr("llm.R", eng._R)
            self._eng[B] = eng
        eng = self._eng[B]
        eng.set_ptr("head.tok_cur", t)
        eng.pos.zero_()
        eng.reset([0] * min(B, 16))                         # the projector reads only the step counter
        eng.projector()
        return eng._R[:n].to(torch.bfloat16).view(*lead, -1)     # values are exact bf16 (pos == 0)


class NativeKVCache:
    """What the reference reads from a cache: ``pkv[0][0].shape[2]`` (t2i_pipeline.py:207,257)."""

    def __init__(self, eng: Engine, batch: int, length: int):
        self.eng, self.batch, self.length = eng, batch, length

    def get_seq_length(self) -> int:
        return self.length

    def __getitem__(self, i):
        shape = (self.batch, self.eng.llm.cfg["num_key_value_heads"], self.length, self.eng.llm.cfg["head_dim"])
        return (SimpleNamespace(shape=shape), SimpleNamespace(shape=shape))


class NativeQwen3Model:
    """Qwen3Model.forward as the reference calls it: fp32 P-token decode calls run the native step; bf16 prefill calls run
    the same step kernels in their prefill mode (llm.native_block) when the pipeline has ``native_prefill`` set, else the
    torch cross-check path (llm.prefill_block)."""

    def __init__(self, pipe, max_kv: int = 4608):
        self._p = pipe
        self.max_kv = max_kv

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return F.embedding(ids, self._p.llm_w.sd["model.embed_tokens.weight"])

    @torch.no_grad()
    def __call__(self, inputs_embeds=None, past_key_values=None, use_cache=True, attention_mask=None, **_):
        pipe = self._p
        B, T, _ = inputs_embeds.shape
        P = pipe.parallel_num
        if past_key_values is None:
            eng = Engine(None, None, pipe.llm_w, num_images=B, branches=1, device=pipe.device, max_kv=self.max_kv,
                         max_tokens=P, parallel_num=P, comm=getattr(pipe, "tp", None))
            eng.set_int("rt.emit_cond", 0)
            cache = NativeKVCache(eng, B, 0)
        else:
            cache = past_key_values
            eng = cache.eng
        past = cache.length
        if inputs_embeds.dtype == torch.float32 and T == P and attention_mask is not None:
            eng.reset([past] * B)                           # decode: all-True mask = block-bidirectional
            eng.residual()[: B * P].copy_(inputs_embeds.reshape(B * P, -1))
            eng.llm_step()
            hidden = eng.hidden().clone().view(B, P, -1)
        else:
            x = inputs_embeds.to(pipe.device, torch.bfloat16)
            if getattr(pipe, "native_prefill", False):
                hidden = native_block(eng, x, past, causal=attention_mask is None)
            else:
                hidden = prefill_block(eng, pipe.llm_w, x, 0, past, causal=attention_mask is None)
        cache.length = past + T
        return SimpleNamespace(last_hidden_state=hidden, past_key_values=cache)
