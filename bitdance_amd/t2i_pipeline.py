"""Drop-in ``BitDanceT2IPipeline`` on the MI355X-native engine.

Mirrors the call surface of /root/reference/modeling/t2i_pipeline.py:
``BitDanceT2IPipeline(model_path, device='cuda')``, ``.generate(...)`` (:110-155), ``.gen_image(...)`` (:157-272),
``.decode_image(...)`` (:274-283), attributes ``tokenizer / parallel_num / ps / vae_patch_size / hidden_size``,
the same model-directory layout and checkpoint keys, ``ValueError`` for unsupported sizes.

What differs is *how* the loop runs: the per-step body (51 diffusion-head evaluations, sign binarisation,
projector, cond+uncond LLM forward batched into one pass over the weights) is two hipGraph launches of
hand-written gfx950 kernels (engine.Engine) instead of thousands of eager torch ops.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from .autoencoder import VQModel
from .engine import Engine, HeadWeights, LlmWeights, ProjWeights
from .llm import prefill_block, prefill_native
from .seams import NativeConnector, NativeDiffHead, NativeQwen3Model

IMAGE_SIZE_LIST = [
    [2048, 512], [1920, 512], [1536, 640], [1280, 768], [1152, 896], [1024, 1024], [896, 1152], [768, 1280],
    [640, 1536], [512, 1920], [512, 2048],
    [1024, 256], [896, 256], [640, 384], [512, 512], [384, 640], [256, 896], [256, 1024],
]

LLM_CFG_KEYS = ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
                "intermediate_size", "rms_norm_eps", "rope_theta", "vocab_size")


def _load_sft(path):
    from safetensors.torch import load_file
    return load_file(path)


def _load_llm_state(model_path: str) -> dict:
    idx = os.path.join(model_path, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = ["model.safetensors"]
    sd = {}
    for fn in files:
        sd.update(_load_sft(os.path.join(model_path, fn)))
    return sd


def _llm_cfg_from_json(cfg: dict) -> dict:
    out = {k: cfg[k] for k in LLM_CFG_KEYS if k in cfg}
    out.setdefault("head_dim", cfg["hidden_size"] // cfg["num_attention_heads"])
    if "rope_theta" not in out:
        out["rope_theta"] = (cfg.get("rope_parameters") or {}).get("rope_theta", 10000.0)
    return out


def load_model_dir(model_path: str) -> dict:
    """Everything ``BitDanceT2IPipeline.__init__`` reads from a released model directory (t2i_pipeline.py:45-75): the HF
    tokenizer + ``config.json`` (Qwen3) + ``model*.safetensors`` (single file or the sharded index), ``ae_config.json`` /
    ``ae.safetensors``, ``vision_head_config.json`` / ``vision_head.safetensors``, ``projector.safetensors``.  Pure host
    code (no GPU), so the checkpoint-loading contract is testable on CPU (tests/test_host_cpu.py)."""
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(model_path)
    with open(os.path.join(model_path, "config.json")) as f:
        llm_cfg = _llm_cfg_from_json(json.load(f))
    with open(os.path.join(model_path, "ae_config.json")) as f:
        ae_config = json.load(f)
    with open(os.path.join(model_path, "vision_head_config.json")) as f:
        head_config = json.load(f)
    return dict(tokenizer=tokenizer, llm_cfg=llm_cfg, llm_sd=_load_llm_state(model_path), ae_config=ae_config,
                ae_sd=_load_sft(os.path.join(model_path, "ae.safetensors")), head_config=head_config,
                head_sd=_load_sft(os.path.join(model_path, "vision_head.safetensors")),
                proj_sd=_load_sft(os.path.join(model_path, "projector.safetensors")))


class BitDanceT2IPipeline:
    def __init__(self, model_path, device="cuda", tp=None, weights: str = "bf16", native_prefill: bool = True):
        """``tp``: a ``bitdance_amd.tp.TPComm`` -- this process is then one rank of a tensor-parallel group (every rank
        constructs the pipeline on its own GPU and makes the same calls with the same seed); None = one GPU."""
        self.device = device
        self._init_from(**load_model_dir(model_path), device=device, tp=tp, weights=weights, native_prefill=native_prefill)

    @classmethod
    def from_components(cls, *, tokenizer, llm_cfg, llm_sd, ae_config, ae_sd, head_config, head_sd, proj_sd,
                        device="cuda", tp=None, weights: str = "bf16", native_prefill: bool = True):
        """Same object from in-memory state dicts (tests, synthetic-weight benchmarks)."""
        self = object.__new__(cls)
        self._init_from(tokenizer, llm_cfg, llm_sd, ae_config, ae_sd, head_config, head_sd, proj_sd, device, tp=tp, weights=weights,
                        native_prefill=native_prefill)
        return self

    def _init_from(self, tokenizer, llm_cfg, llm_sd, ae_config, ae_sd, head_config, head_sd, proj_sd, device, tp=None,
                   weights: str = "bf16", native_prefill: bool = True):
        """``weights`` = "fp8": the streamed Linears are stored e4m3 + per-channel scales (a separate precision mode, BASELINE
        config 5; the once-per-image prefill keeps bf16 copies)."""
        self.weights = weights
        self.native_prefill = native_prefill               # False: torch prefill on a second, original-layout copy of the LLM
        if not torch.cuda.is_available():
            raise RuntimeError("BitDanceT2IPipeline (bitdance_amd) needs a ROCm GPU; there is no CPU fallback")
        self.device = device
        self.tp = tp if (tp is not None and tp.size > 1) else None
        tpr, tps = (self.tp.rank, self.tp.size) if self.tp else (0, 1)
        self.tokenizer = tokenizer
        self.llm_config = SimpleNamespace(**llm_cfg)
        self.hidden_size = llm_cfg["hidden_size"]
        self.llm_w = LlmWeights.from_state_dict(llm_sd, llm_cfg, device, keep_for_prefill=not native_prefill, tp_rank=tpr, tp_size=tps,
                                                weights=weights)
        self.ae_config = ae_config
        self.ae = VQModel(**ae_config).eval()
        if ae_sd is not None:
            self.ae.load_state_dict(ae_sd, strict=True, assign=True)
        self.ae.to(device)
        self.vae_patch_size = 2 ** (len(ae_config["ddconfig"]["ch_mult"]) - 1)
        self.vision_head_config = head_config
        self.head_w = HeadWeights.from_state_dict(head_sd, device, tp_rank=tpr, tp_size=tps, weights=weights)
        self.parallel_num = head_config["parallel_num"]
        if self.parallel_num not in (1, 4, 16, 64):
            raise NotImplementedError("the native path implements parallel_num 64 / 16 (the released 64x / 16x models), 4 and 1 "
                                      "(one token per AR step: the loop of MLLModel.gen_image_full_causal, mllm.py:274-384)")
        self.ps = int(self.parallel_num ** 0.5)
        self.proj_w = ProjWeights.from_state_dict(proj_sd, device, weights=weights)
        # the reference's operator seams (same attribute names), each backed by the native engine
        self.llm_model = SimpleNamespace(model=NativeQwen3Model(self))
        self.vision_head = NativeDiffHead(self)
        self.embed_vision_mlp = NativeConnector(self)
        self.build_pos_embed()
        self._engines: dict = {}
        self._stream = torch.cuda.Stream(device=device)
        self.use_graph = True
        self.last_timings: dict = {}

    # -- 2-D sincos position table (t2i_pipeline.py:79-107) -----------------------------------------------
    def build_pos_embed(self, max_len=4096):
        n = max_len // self.vae_patch_size
        half = self.hidden_size // 2
        omega = torch.arange(half // 2, dtype=torch.float32)
        omega /= half / 2.0
        omega = 1.0 / 10000 ** omega
        ang = torch.einsum("m,d->md", torch.arange(n, dtype=torch.float32), omega)
        self.pos_embed_1d = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).to(self.device)

    def get_2d_embed(self, h, w, ps=1):
        half = self.hidden_size // 2
        gv = self.pos_embed_1d[:h].view(h, 1, half).expand(h, w, half)
        gh = self.pos_embed_1d[:w].view(1, w, half).expand(h, w, half)
        pe = torch.cat([gh, gv], dim=-1)
        pe = pe.reshape(h // ps, ps, w // ps, ps, 2 * half).permute(0, 2, 1, 3, 4)
        return pe.reshape(h * w, 2 * half)

    # -- public API ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, prompt: str, height: int = 1024, width: int = 1024, num_sampling_steps: int = 50,
                 guidance_scale: float = 7.5, num_images: int = 1, seed: int = 1234):
        from PIL import Image
        if seed is not None:
            from transformers import set_seed
            set_seed(seed)
        max_length = (height // self.vae_patch_size) * (width // self.vae_patch_size)
        image_size = [height, width]
        if image_size not in IMAGE_SIZE_LIST:
            raise ValueError(f"image_size {image_size} is not supported. Please choose from {IMAGE_SIZE_LIST}")
        with torch.amp.autocast("cuda", enabled=True, dtype=torch.bfloat16):
            imgs = self.gen_image(
                cond_prompt=f"<|im_start|>user\n{prompt}<|im_end|>\n<|im_start|>assistant\n",
                uncond_prompt="<|im_start|>assistant\n", guidance_scale=guidance_scale,
                num_sampling_steps=num_sampling_steps, num_images=num_images, image_size=image_size,
                max_length=max_length, show_progress=True)
        arr = torch.clamp(127.5 * imgs + 128.0, 0, 255).permute(0, 2, 3, 1).to("cpu", dtype=torch.uint8).numpy()
        return [Image.fromarray(np.ascontiguousarray(a)) for a in arr]

    def _engine(self, num_images: int, branches: int, tokens: int, kv: int) -> Engine:
        lmax = ((kv + 255) // 256) * 256
        key = (num_images, branches, tokens, lmax)
        if key not in self._engines:
            self._engines.clear()                      # one resident engine (KV cache + workspaces) at a time
            torch.cuda.empty_cache()
            self._engines[key] = Engine(self.head_w, self.proj_w, self.llm_w, num_images=num_images,
                                        branches=branches, device=self.device, max_tokens=tokens, max_kv=lmax,
                                        tune=getattr(self, "tune", None), parallel_num=self.parallel_num, comm=self.tp,
                                        extra_ints=getattr(self, "extra_ints", None),      # e.g. {"tp.ada_split": 1} (engine.Engine)
                                        # flash-decode splits of the KV cache: 12 once the cache passes ~2k tokens (a 1024 px image ends
                                        # at 4.4k): 248 vs 265 us per layer at 4096 cached tokens, no difference below 1k
                                        # (profiles/r03_llm_attn_splits.log); more splits only add partial-output traffic
                                        # (a tensor-parallel rank holds 8 / tp kv heads: scale the splits so the grid keeps its size)
                                        attn_splits=getattr(self, "attn_splits", None)
                                        or min(32, (12 if lmax > 2048 else 8) * (self.tp.size if self.tp is not None else 1)))
        return self._engines[key]

    def _prompt_ids(self, cond_prompt, uncond_prompt, image_size, cfg_on):
        tok = self.tokenizer
        hp, wp = image_size[0] // self.vae_patch_size, image_size[1] // self.vae_patch_size
        tail = [tok.convert_tokens_to_ids("<|vision_start|>"), tok.convert_tokens_to_ids(f"<|res_{hp}|>"),
                tok.convert_tokens_to_ids(f"<|res_{wp}|>")]
        tail += [tok.convert_tokens_to_ids(f"<|query_{i}|>") for i in range(1, self.parallel_num)]
        cond = list(tok.encode(cond_prompt)) + tail
        uncond = (list(tok.encode(uncond_prompt)) + tail) if cfg_on else None
        return cond, uncond

    @torch.no_grad()
    def gen_image(self, cond_prompt, uncond_prompt=None, guidance_scale: float = 1.0, num_sampling_steps: int = 50,
                  max_length: int = 64, num_images: int = 1, image_size=[256, 256], show_progress: bool = False,
                  noise: torch.Tensor | None = None, return_tokens: bool = False):
        P = self.parallel_num
        num_steps = max_length // P
        cfg_on = guidance_scale > 1.0
        branches = 2 if cfg_on else 1
        h, w = image_size[0] // self.vae_patch_size, image_size[1] // self.vae_patch_size
        # the reference fails in its pos-embed slice / final rearrange when the token budget and the grid disagree
        # (t2i_pipeline.py:244,279-281); here the engine's token / position / noise buffers are sized from h*w
        if max_length != h * w or max_length % P:
            raise ValueError(f"max_length={max_length} must equal (H/{self.vae_patch_size})*(W/{self.vae_patch_size})={h * w} "
                             f"and be a multiple of parallel_num={P}")
        cond_ids, uncond_ids = self._prompt_ids(cond_prompt, uncond_prompt, image_size, cfg_on)
        embed = self.llm_w.sd["model.embed_tokens.weight"]
        ctx = [F.embedding(torch.tensor(ids, device=self.device, dtype=torch.long), embed) for ids in [cond_ids, uncond_ids][:branches]]
        return self.gen_image_from_context(ctx[0], ctx[1] if cfg_on else None, guidance_scale=guidance_scale,
                                           num_sampling_steps=num_sampling_steps, num_images=num_images, image_size=image_size,
                                           noise=noise, return_tokens=return_tokens)

    @torch.no_grad()
    def gen_image_from_context(self, cond_ctx: torch.Tensor, uncond_ctx: torch.Tensor | None, *, guidance_scale: float = 1.0,
                               num_sampling_steps: int = 50, num_images: int = 1, image_size=[256, 256],
                               noise: torch.Tensor | None = None, return_tokens: bool = False,
                               force_tokens: torch.Tensor | None = None, trace: dict | None = None):
        """The AR loop over an arbitrary context: ``cond_ctx`` / ``uncond_ctx`` [T, D] are the input EMBEDDINGS of everything
        before the first patch, query tokens included -- token embeddings of a prompt (``gen_image``, t2i_pipeline.py:175-198) or
        an interleaved text + image context (``MLLModel.forward_inference_block_causal``, mllm.py:719-745).  Prefill = causal
        over ctx[:-P], all-visible over the last P (:199-236); then the 64-step loop (:241-270).
        ``force_tokens`` [num_images, h*w, C] / ``trace`` (tests): teacher forcing -- the fed-back tokens are replaced by these
        (eager launches) and the pre-sign latents of every step are appended to ``trace["pred"]``."""
        P = self.parallel_num
        cfg_on = guidance_scale > 1.0
        branches = 2 if cfg_on else 1
        if cfg_on and uncond_ctx is None:
            raise ValueError("guidance_scale > 1 needs an unconditional context")
        h, w = image_size[0] // self.vae_patch_size, image_size[1] // self.vae_patch_size
        if (h * w) % P:
            raise ValueError(f"(H/{self.vae_patch_size})*(W/{self.vae_patch_size})={h * w} must be a multiple of parallel_num={P}")
        num_steps = (h * w) // P
        ctxs = [c.to(self.device, torch.bfloat16) for c in [cond_ctx, uncond_ctx][:branches]]
        if any(c.dim() != 2 or c.shape[0] < P or c.shape[1] != self.hidden_size for c in ctxs):
            raise ValueError("a context is [T >= parallel_num, hidden_size] input embeddings")
        kv_need = max(c.shape[0] for c in ctxs) + num_steps * P + P
        eng = self._engine(num_images, branches, h * w, kv_need)
        dev = self.device
        st = self._stream
        st.wait_stream(torch.cuda.current_stream())
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(st):
            eng.set_schedule(num_sampling_steps, guidance_scale, num_steps,
                             time_shift=float(self.vision_head_config.get("time_shift", 1.0)))
            if noise is None:
                eng.draw_noise(num_steps)
            else:
                eng.load_noise(noise.to(dev))
            pos = self.get_2d_embed(h, w, ps=self.ps)
            eng.pos[: h * w].copy_(pos)
            ev[0].record(st)
            hid = []
            kv = []
            if self.tp is not None:
                # the first in-kernel exchanges run inside the prefill: ranks whose engine build / checkpoint load took
                # different times must meet on the host first, or the early rank spends its wait budget on the late one
                st.synchronize()
                self.tp.barrier()
            if self.native_prefill:
                embs = []
                for x in ctxs:
                    embs += [x] * num_images
                hid_last, kv = prefill_native(eng, embs)
                hid = [hid_last.to(torch.bfloat16)]
            else:
                for br, x in enumerate(ctxs):
                    x = x.unsqueeze(0).repeat(num_images, 1, 1)
                    T0 = x.shape[1] - P
                    prefill_block(eng, self.llm_w, x[:, :T0], br * num_images, 0, causal=True)
                    hid.append(prefill_block(eng, self.llm_w, x[:, T0:], br * num_images, T0, causal=False))
                    kv += [x.shape[1]] * num_images
            if self.tp is not None:
                st.synchronize()
                self.tp.check()                            # a failed prefill exchange raises here, on every rank, not after the loop
            cond0 = torch.cat(hid, dim=0)[:, -P:] + pos[None, :P]          # bf16 + fp32 -> fp32 (t2i:244-245)
            eng.set_cond(cond0.reshape(eng.M, -1))
            eng.reset(kv)
            if self.use_graph:
                eng.capture(0)
                if num_steps > 1:
                    eng.capture(1)
            ev[1].record(st)
            if self.tp is not None:                        # ranks enter the exchange loop together (in-kernel waits are bounded)
                st.synchronize()
                self.tp.barrier()
            for step in range(num_steps):
                if force_tokens is not None or trace is not None:
                    eng.head_sample()
                    if trace is not None:
                        trace.setdefault("pred", []).append(eng.pred().clone())
                    if force_tokens is not None:
                        eng.tok_cur().copy_(force_tokens[:, step * P:(step + 1) * P].to(dev))
                    if step + 1 < num_steps:
                        eng.projector()
                        eng.llm_step()
                elif self.use_graph:
                    eng.launch(0)
                    if step + 1 < num_steps:
                        eng.launch(1)
                else:
                    eng.head_sample()
                    if step + 1 < num_steps:
                        eng.projector()
                        eng.llm_step()
            ev[2].record(st)
            if self.tp is not None:
                st.synchronize()
                self.tp.check()                            # raises if a peer never arrived
            tokens = eng.tok_all[:, : h * w].clone()
            if return_tokens:
                out = tokens
            else:
                out = self.decode_image(tokens, [h, w], ps=self.ps)
            ev[3].record(st)
        torch.cuda.current_stream().wait_stream(st)
        self._events = ev
        return out

    def timings(self) -> dict:
        """Milliseconds of the last gen_image call: prefill (+graph capture on first use), AR loop, AE decode."""
        ev = self._events
        ev[3].synchronize()
        return {"prefill_ms": ev[0].elapsed_time(ev[1]), "ar_loop_ms": ev[1].elapsed_time(ev[2]),
                "decode_ms": ev[2].elapsed_time(ev[3])}

    def decode_image(self, image_latents, image_size=None, ps=1):
        """Un-raster the tokens (t2i_pipeline.py:274-283) and run the conv decoder: the native implicit-GEMM kernels under bf16
        autocast (autoencoder.VQModel.decode -> ae_native.NativeDecoder); outside autocast, or for a configuration the native
        kernels do not cover, the torch module on MIOpen -- whose immediate mode has no tuned gfx950 entries, hence
        cudnn.benchmark (MIOpen Find, cached per shape) around the call.
        Under tensor parallelism every rank holds the same tokens; with several images per call each rank decodes its share of
        the batch and the images are summed into place across the group (``tp_split_decode``; 4 images at tp 4: 109 -> 27 ms + a
        48 MB all-reduce).  One image stays replicated: a spatial split would all-reduce every GroupNorm's statistics and exchange
        a halo row per convolution for 33 ms of a multi-second image."""
        if image_size is None:
            h = w = int(image_latents.size(1) ** 0.5)
        else:
            h, w = image_size
        b, _, c = image_latents.shape
        x = image_latents.view(b, h // ps, w // ps, ps, ps, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, h, w)
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = True
        try:
            # (a GAN decoder draws torch.randn_like from the GLOBAL device generator: ranks decoding different numbers of images
            # would leave the group with diverged generator offsets, and the next image's sampling noise -- engine.draw_noise, same
            # generator -- would differ across ranks.  Every rank decodes the whole batch there: the replicated state stays replicated.)
            gan = bool(getattr(getattr(self.ae, "decoder", None), "gan", False))
            # large batches in chunks (per-image results are independent: convolutions and per-sample GroupNorm): the native decoder's
            # padded NHWC work buffers at 1024 px are ~0.5 GB per image and layer -- 32 images at once exceeded the 288 GB beside the
            # model, its KV caches and workspaces.  A GAN decoder draws noise per call, so its batch stays whole.
            chunk = int(getattr(self, "decode_chunk", 8))

            def dec(t):
                if t.shape[0] > chunk and not gan:
                    return torch.cat([self.ae.decode(t[i:i + chunk]) for i in range(0, t.shape[0], chunk)])
                return self.ae.decode(t)
            if self.tp is not None and b > 1 and getattr(self, "tp_split_decode", True) and not gan:
                per = (b + self.tp.size - 1) // self.tp.size
                lo = self.tp.rank * per
                mine = x[lo:lo + per]
                part = dec(mine if mine.shape[0] else x[:1])                 # a rank without a share still learns shape / dtype
                full = torch.zeros((b,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
                if mine.shape[0]:
                    full[lo:lo + mine.shape[0]] = part
                return self.tp.all_reduce_(full)                              # disjoint shares + zeros: exact
            return dec(x)
        finally:
            torch.backends.cudnn.benchmark = prev
