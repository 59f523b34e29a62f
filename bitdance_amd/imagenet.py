"""Class-conditional ImageNet BitDance on the native engine: the parallel variants (16x, 4x) and the 1x models.

``parallel_num`` = 16 / 4: ``imagenet_gen/src/model_parallel.py`` (BitDance-B-16x / -4x).  ``parallel_num`` = 1:
``imagenet_gen/src/model.py`` (BitDance-B/L/H-1x) -- the same loop with one token per AR step: no
query tokens (model.py:372-375), purely causal attention (layers.py:126-129), the RoPE table in raster order (:181-190) and the
MLP diffusion head (diff_head.py:228-253; engine.HeadWeights detects it from the checkpoint keys, ``head.variant`` = 1).

Mirrors the inference surface of ``imagenet_gen/src/model_parallel.py``:
``BitDance(...).sample(cond, sample_steps, cfg_scale, cfg_schedule)`` (:371-419) with ``head_sample``'s linear CFG ramp
(:352-369), the un-mixed first AR step (cfg_iter == 1.0 there, so cond and uncond rows are sampled independently),
LFQ ``sign`` and the patch-raster un-patchify.  State-dict keys are the reference's (minus ``vae.*``).

What runs where:
  * I3 ``diff_head_parallel.TransEncoder`` + ``sampling_parallel.euler_maruyama`` (>= 70 % of the model's FLOPs): the HIP
    head -- the T2I kernels with head_dim 64 attention and no final sigmoid (engine.HeadWeights(head_dim=64,
    final_sigmoid=False));
  * I2 the KV-cached block-causal transformer: every 16-token decode step (``proj_in`` -> ``forward_model``) on the HIP
    engine (``bd_projector`` / ``bd_llm_step`` in their imagenet variant: bf16 residual stream, fp32 norm weights,
    interleaved 2-D RoPE, head_dim-64 attention with the reference's rounding points, static K/V cache).  The FIRST call
    (class tokens + query tokens under the mixed causal / block mask, once per batch) runs on the same step kernels since round
    4 (``_first_step_native``: causal blocks of P class tokens, then the last P tokens as a bidirectional block; fp32 residual
    stream, emb_norm on the raw rows) -- like the T2I prefill; the torch form stays behind ``native_first_step = False`` /
    ``native_transformer = False`` as a cross-check;
  * the conv decoder runs on the native kernels under bf16 autocast (autoencoder.VQModel.decode, round 3).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .engine import Engine, HeadWeights, pack_linear, pack_swiglu, _bf16

__all__ = ["BitDance"]


class _InProj:
    """MLPConnector (model_parallel.py:62-75) for the engine: w1 row-major bf16 (VALU, K = latent channels), w2 packed."""

    def __init__(self, sd: dict, device):
        w1, w2 = sd["proj_in.w1.weight"], sd["proj_in.w2.weight"]
        self.D, self.hid, self.C = w2.shape[0], w2.shape[1], w1.shape[1]
        self.ptrs = {"proj.w1": _bf16(w1, device), "proj.b1": _bf16(sd["proj_in.w1.bias"], device),
                     "proj.w2": pack_linear([w2], device), "proj.b2": _bf16(sd["proj_in.w2.bias"], device)}

    def ints(self) -> dict:
        return {"proj.D": self.D, "proj.C": self.C, "proj.hid": self.hid, "proj.variant": 1}


class _InTransformer:
    """layers_parallel.TransformerBlock stack for the engine (llm.variant = 1)."""
    variant = 1

    def __init__(self, sd: dict, n_layer: int, n_head: int, rope: torch.Tensor, device):
        D = sd["norm.weight"].shape[0]
        F_ = sd["layers.0.feed_forward.w2.weight"].shape[1]
        self.cfg = {"hidden_size": D, "rms_norm_eps": 1e-6}
        self.D, self.F, self.L, self.nh = D, F_, n_layer, n_head
        f32 = lambda t: t.detach().to(device, torch.float32).contiguous()
        p = {"llm.final_norm": f32(sd["norm.weight"]), "llm.emb_norm": f32(sd["emb_norm.weight"]),
             "llm.rope2d": f32(rope)}
        for i in range(n_layer):
            s_, d = f"layers.{i}.", f"llm.l{i}."
            p[d + "in_norm"] = f32(sd[s_ + "attention_norm.weight"])
            p[d + "post_norm"] = f32(sd[s_ + "ffn_norm.weight"])
            p[d + "wqkv"] = pack_linear([sd[s_ + "attention.wqkv.weight"]], device)
            p[d + "wo"] = pack_linear([sd[s_ + "attention.wo.weight"]], device)
            w1 = sd[s_ + "feed_forward.w1.weight"]
            p[d + "wgu"] = pack_swiglu(w1[:F_], w1[F_:], device)
            p[d + "wdown"] = pack_linear([sd[s_ + "feed_forward.w2.weight"]], device)
        self.ptrs = p

    def ints(self, Lmax: int, splits: int) -> dict:
        return {"llm.D": self.D, "llm.L": self.L, "llm.nh": self.nh, "llm.nkv": self.nh, "llm.F": self.F,
                "llm.Lmax": Lmax, "llm.splits": splits, "llm.head_dim": 64, "llm.variant": 1}

    def rope_tables(self, max_pos: int, device):           # the rotate-half tables of the Qwen3 path are unused here
        return self.ptrs["llm.rope2d"], self.ptrs["llm.rope2d"]


def _pos_2d(resolution: int, patch: int) -> torch.Tensor:
    n = resolution // patch                                   # layers_parallel.py:235-252, one scale: cell centres
    c = torch.arange(n, dtype=torch.float32) + 0.5
    gy, gx = torch.meshgrid(c, c, indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=1)


def _patch_raster(x: torch.Tensor, p: int, H: int, W: int) -> torch.Tensor:
    """Rows of x in (H W) order -> (H/p W/p p p) order (utils.py:91-113)."""
    tail = x.shape[1:]
    return x.reshape(H // p, p, W // p, p, -1).permute(0, 2, 1, 3, 4).reshape(H * W, *tail)


def rope_table_2d(head_dim: int, resolution: int, patch: int, cls_token_num: int, parallel_num: int) -> torch.Tensor:
    """[cls + P-1 + h*w - P, head_dim/2, 2] (cos, sin): position 0 for class/query tokens, (x+1, y+1) cell centres for the
    image tokens in patch-raster order, the last P rows dropped (model_parallel.py:197-212, layers_parallel.py:255-270)."""
    n = resolution // patch
    half = head_dim // 2
    freqs = 1.0 / (10000 ** (torch.arange(0, half, 2)[: half // 2].float() / half))
    t = torch.cat([torch.zeros(cls_token_num + parallel_num - 1, 2), _pos_2d(resolution, patch) + 1.0])
    fr = torch.outer(t.flatten(), freqs).view(t.shape[0], -1)
    fc = torch.stack([torch.cos(fr), torch.sin(fr)], dim=-1)
    fc[-n * n:] = _patch_raster(fc[-n * n:], int(parallel_num ** 0.5), n, n)
    return fc[:-parallel_num]


def block_causal_mask(total: int, causal: int, block: int) -> torch.Tensor:
    """Additive mask: causal, each `block`-token group after the first `causal` tokens bidirectional (model_parallel.py:90-101)."""
    m = torch.zeros(total, total)
    m.masked_fill_(torch.triu(torch.ones(total, total), diagonal=1).bool(), float("-inf"))
    for i in range(causal, total, block):
        m[i:i + block, i:i + block] = 0
    return m


class BitDance:
    def __init__(self, state_dict: dict, *, dim: int, n_layer: int, n_head: int, latent_dim: int, resolution: int = 256,
                 down_size: int = 16, patch_size: int = 1, cls_token_num: int = 64, num_classes: int = 1000,
                 parallel_num: int = 16, time_shift: float = 1.0, device="cuda", vae=None, **_unused):
        if not torch.cuda.is_available():
            raise RuntimeError("bitdance_amd.imagenet.BitDance needs a GPU (HIP head); there is no CPU fallback")
        if parallel_num not in (1, 4, 16):
            raise NotImplementedError("native imagenet path: parallel_num must be 16, 4 (parallel checkpoints) or 1 (1x checkpoints)")
        self.time_shift = float(time_shift)
        self.device = torch.device(device)
        self.dim, self.n_layer, self.n_head = dim, n_layer, n_head
        self.P, self.cls_token_num, self.num_classes = parallel_num, cls_token_num, num_classes
        self.h = self.w = resolution // (down_size * patch_size)
        self.latent_dim = latent_dim
        self.total_tokens = self.h * self.w + cls_token_num
        self.vae = vae
        self.native_transformer = True                       # False: torch ops for every step (reference arithmetic)
        self.native_first_step = True                        # False: the class / query-token call on torch ops (cross-check)
        sd = {k: v.detach().to(self.device, torch.float32) for k, v in state_dict.items() if not k.startswith("vae.")}
        self.w_ = sd
        head_sd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
        self.head_w = HeadWeights.from_state_dict(head_sd, self.device, head_dim=64, final_sigmoid=False)
        self._eng: dict = {}
        self.freqs_cis = rope_table_2d(dim // n_head, resolution, down_size * patch_size, cls_token_num, parallel_num).to(self.device)
        self.attn_mask = block_causal_mask(self.h * self.w + cls_token_num - 1, cls_token_num - 1, parallel_num)[None, None] \
            .to(self.device)
        if dim // n_head != 64:
            raise NotImplementedError("native imagenet transformer: head_dim must be 64")
        self.proj_w = _InProj(sd, self.device)
        self.tr_w = _InTransformer(sd, n_layer, n_head, self.freqs_cis, self.device)
        self._tr: dict = {}
        # AR steps 1 .. on ONE engine (head + proj_in + transformer, the T2I loop's structure) as replays of two captured hipGraphs
        # per step; combined_engine = False: the per-step path over separate head / transformer engines (cross-check),
        # use_graph = False: the combined engine with eager launches
        self.combined_engine = True
        self.use_graph = True
        self._comb: dict = {}
        self._stream = None

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

    # ------------------------------------------------------------------ transformer decode step (HIP)
    def _tr_engine(self, bsz: int) -> Engine:
        if bsz not in self._tr:
            eng = Engine(None, self.proj_w, self.tr_w, num_images=bsz, branches=1, device=self.device,
                         max_tokens=self.P, max_kv=self.total_tokens, parallel_num=self.P)
            eng.set_int("rt.emit_cond", 0)
            eng._tok = torch.zeros(bsz * self.P, self.proj_w.C, dtype=torch.float32, device=self.device)
            eng.set_ptr("head.tok_cur", eng._tok)
            self._tr[bsz] = eng
        return self._tr[bsz]

    def _load_cache(self, eng: Engine, caches, T0: int) -> None:
        """K/V of the torch first step -> the engine's static cache [layer][seq][head][Lmax][64] bf16."""
        bsz = caches[0][0].shape[0]
        shape = (self.n_layer, bsz, self.n_head, eng.Lmax, 64)
        kc = eng.view("llm.k_cache", torch.bfloat16, shape)
        vc = eng.view("llm.vt_cache", torch.bfloat16, shape)
        for l, (k, v) in enumerate(caches):
            kc[l, :, :, :T0].copy_(k[:, :, :T0])
            vc[l, :, :, :T0].copy_(v[:, :, :T0])
        eng.reset([T0] * min(bsz, 16))

    def _cache_views(self, eng: Engine, T0: int) -> list:
        """The first T0 cached positions of every layer of ``eng`` as (K, V) views [bsz, heads, T0, 64] (what _load_cache takes)."""
        bsz = eng.B * eng.branches
        shape = (self.n_layer, bsz, self.n_head, eng.Lmax, 64)
        kc = eng.view("llm.k_cache", torch.bfloat16, shape)
        vc = eng.view("llm.vt_cache", torch.bfloat16, shape)
        return [(kc[l][:, :, :T0], vc[l][:, :, :T0]) for l in range(self.n_layer)]

    def _first_step_native(self, eng: Engine, ids: torch.Tensor) -> torch.Tensor:
        """The first ``forward_model`` call (model_parallel.py:386-388; 1x: model.py:372-377) on the step kernels: the class tokens
        (+ P - 1 query tokens) of every sequence under ``attn_mask[:T0, :T0]`` -- blocks of P tokens with the causal mask inside the
        block for the first T0 - P tokens, then the last P tokens as one bidirectional block (exactly a decode block: every cached
        key plus the block).  The class embedding is fp32, so this call's residual stream is fp32 ("rt.in_first"; the decode steps'
        is bf16) and emb_norm runs on the raw rows.  K / V land in ``eng``'s static cache.  Returns norm(x) of the last P tokens
        [bsz, P, D] fp32."""
        P, n_cls, w = self.P, self.cls_token_num, self.w_
        bsz = ids.shape[0]
        T0 = n_cls + P - 1
        c = F.embedding(ids, w["cls_embedding.weight"]).view(bsz, n_cls, -1)          # a gather: fp32 rows of the table
        x = torch.cat([c, w["query_token"].repeat(bsz, 1, 1)], dim=1) if P > 1 else c
        R = eng.residual()[: bsz * P].view(bsz, P, -1)
        nc = T0 - P                                                                   # tokens ahead of the last block: causal

        def run(rows: torch.Tensor, past: int, causal: bool) -> None:
            R.zero_()
            R[:, : rows.shape[1]].copy_(rows)
            eng.reset([past] * min(bsz, 16))
            for k, v in (("rt.in_first", 1), ("rt.llm_causal", int(causal)), ("rt.no_advance", 1)):
                eng.set_int(k, v)
            try:
                eng.llm_step()
            finally:
                for k in ("rt.in_first", "rt.llm_causal", "rt.no_advance"):
                    eng.set_int(k, 0)
        for c0 in range(0, nc, P):
            run(x[:, c0: min(c0 + P, nc)], c0, True)       # a short last block: its pad rows' K / V are overwritten by the next block
        run(x[:, nc:], nc, False)
        eng.reset([T0] * min(bsz, 16))
        return eng.hidden().view(bsz, P, -1).clone()

    def _decode_step(self, eng: Engine, tokens: torch.Tensor) -> torch.Tensor:
        """proj_in + forward_model for one 16-token block on the engine; returns norm(x) [bsz, P, D] (bf16 values)."""
        bsz = tokens.shape[0]
        eng._tok.copy_(tokens.reshape(bsz * self.P, -1))
        eng.projector()
        eng.llm_step()
        return eng.hidden().view(bsz, self.P, -1)

    # ------------------------------------------------------------------ the whole AR step on one engine (HIP graphs)
    def _combined_engine(self, n: int, branches: int) -> Engine:
        """Head + proj_in + transformer for ``n`` samples x ``branches`` CFG branches (sequences ordered [cond.., uncond..] like the
        reference's ``torch.cat([cond, cond_null])``, model_parallel.py:377-380): phase 1 = proj_in + forward_model of the last
        tokens -> the head's condition (norm(x) + pos_for_diff, fused), phase 0 = DiffHead.sample with the step's guidance scale
        read from a device table."""
        key = (n, branches)
        if key not in self._comb:
            eng = Engine(self.head_w, self.proj_w, self.tr_w, num_images=n, branches=branches, device=self.device,
                         max_tokens=self.h * self.w, max_kv=self.total_tokens, parallel_num=self.P, extra_ints={"proj.rows_all": 1},
                         tune=getattr(self, "tune", None))
            eng.pos[: self.h * self.w].copy_(self.w_["pos_for_diff.weight"])
            eng.cfg_table = torch.ones(self.h * self.w // self.P + 1, dtype=torch.float32, device=self.device)
            eng.set_ptr("head.cfg_table", eng.cfg_table)
            self._comb[key] = eng
        return self._comb[key]

    def _graph_steps(self, eng: Engine, caches, T0: int, last: torch.Tensor, sample_steps: int, cfgs: list, noise, force_tokens,
                     preds: list, toks: list) -> None:
        """AR steps 1 .. seq_len - 1 on the combined engine.  ``last``: the tokens of step 0 for all sequences [bsz, P, C];
        ``cfgs[i]``: guidance scale of AR step i.  RNG: the draws of every remaining step, in the reference's order, up front."""
        P, seq_len, mult = self.P, len(cfgs), eng.branches
        n = eng.B
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        st = self._stream
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            self._load_cache(eng, caches, T0)
            eng.set_schedule(sample_steps, cfgs[1], seq_len, time_shift=self.time_shift)
            eng.cfg_table[:seq_len].copy_(torch.tensor(cfgs, dtype=torch.float32))
            if noise is None:
                shp = (n, P, self.head_w.C)
                for s_ in range(1, seq_len):
                    x = torch.randn(shp, device=self.device)
                    eng.noise[s_, 0] = x.view(n * P, -1)
                    for k in range(sample_steps):
                        eng.noise[s_, k + 1] = torch.randn_like(x).view(n * P, -1)
            else:
                for s_ in range(1, seq_len):
                    eng.noise[s_].copy_(noise[s_].to(self.device).view(sample_steps + 1, n * P, -1))
            tok_cur = eng.view("head.tok_cur", torch.float32, (mult * n * P, self.head_w.C))
            tok_cur.copy_(last.reshape(mult * n * P, -1))
            if self.use_graph:
                eng.capture(1)                                   # (stream capture records the launches, it does not run them)
                eng.capture(0)
            for i in range(1, seq_len):
                if self.use_graph:
                    eng.launch(1)
                    eng.launch(0)
                else:
                    eng.projector(); eng.llm_step(); eng.head_sample()
                pred = torch.cat([eng.pred().clone()] * mult, dim=0)
                preds.append(pred)
                tok = torch.sign(pred)
                toks.append(tok)
                if force_tokens is not None:
                    tok_cur.copy_(force_tokens[:, i * P:(i + 1) * P].to(self.device, torch.float32).reshape(mult * n * P, -1))
        torch.cuda.current_stream().wait_stream(st)

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
        eng.set_schedule(steps, cfg, 1, time_shift=self.time_shift)
        if noise is None:
            eng.draw_noise(1)                                # randn + N x randn_like: the reference's RNG order
        else:
            eng.load_noise(noise.view(1, steps + 1, B, self.P, -1))
        eng.reset([0] * min(rows, 16))                        # the head reads only the step counter
        eng.set_cond(z)
        eng.head_sample()
        x = eng.pred().clone()
        return torch.cat([x] * mult, dim=0)

    @staticmethod
    def _cfg_at(cfg_scale: float, cfg_schedule: str, i: int, seq_len: int) -> float:
        """Guidance scale of AR step i (model_parallel.py:356-365)."""
        if cfg_scale <= 1.0:
            return 1.0
        if cfg_schedule == "constant":
            return cfg_scale
        if cfg_schedule == "linear":
            return 1.0 + (cfg_scale - 1.0) * i / seq_len
        raise NotImplementedError(f"unknown cfg_schedule {cfg_schedule}")

    def _graph_ok(self, cfg_scale: float, cfg_schedule: str, seq_len: int) -> bool:
        """The combined-engine path needs every step after the first to have the same CFG arity (true for both schedules: the
        linear ramp is > 1 from step 1 on) and the sequence count to fit the engine."""
        if not self.combined_engine:
            return False
        return all((self._cfg_at(cfg_scale, cfg_schedule, j, seq_len) > 1.0) == (cfg_scale > 1.0) for j in range(1, seq_len))

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
        torch_first = not (self.native_transformer and self.native_first_step)   # the torch cross-check path keeps its own K / V
        caches = [(torch.zeros(bsz, self.n_head, self.total_tokens, hd, device=dev),
                   torch.zeros(bsz, self.n_head, self.total_tokens, hd, device=dev)) for _ in range(self.n_layer)] if torch_first else None
        seq_len = self.h * self.w // P
        w = self.w_
        toks, preds, last = [], [], None
        eng_t = self._tr_engine(bsz) if self.native_transformer else None
        for i in range(seq_len):
            if i == 0 and eng_t is not None and self.native_first_step:
                T0 = n_cls + P - 1
                x = self._first_step_native(eng_t, ids)
                caches = self._cache_views(eng_t, T0)             # (for the combined engine of the graph path)
            elif i == 0:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    T0 = n_cls + P - 1
                    c = F.embedding(ids, w["cls_embedding.weight"]).view(bsz, n_cls, -1)
                    x = torch.cat([c, w["query_token"].repeat(bsz, 1, 1)], dim=1) if P > 1 else c   # 1x: model.py:372-375
                    x = self._forward_model(x, self.attn_mask[:, :, :T0, :T0], 0, T0, caches)[:, -P:, :]
                if eng_t is not None:
                    self._load_cache(eng_t, caches, T0)
            elif eng_t is not None:
                x = self._decode_step(eng_t, last)
            else:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    s0 = P * (i - 1) + n_cls + P - 1
                    x = self._forward_model(self._proj_in(last), self.attn_mask[:, :, s0:s0 + P, :s0 + P], s0, s0 + P, caches)
            z = x.float() + w["pos_for_diff.weight"][i * P:(i + 1) * P, :]
            ci = self._cfg_at(cfg_scale, cfg_schedule, i, seq_len)
            pred = self._head_sample(z.float(), ci, sample_steps, None if noise is None else noise[i].to(dev))
            preds.append(pred)
            tok = torch.sign(pred)
            toks.append(tok)
            last = tok if force_tokens is None else force_tokens[:, i * P:(i + 1) * P].to(dev, tok.dtype)
            if i == 0 and eng_t is not None and seq_len > 1 and self._graph_ok(cfg_scale, cfg_schedule, seq_len):
                # every later step: one engine, two graph replays per step (projector + transformer | head sampling)
                mult = 2 if cfg_scale > 1.0 else 1
                cfgs = [self._cfg_at(cfg_scale, cfg_schedule, j, seq_len) for j in range(seq_len)]
                self._graph_steps(self._combined_engine(bsz // mult, mult), caches, T0, last, sample_steps, cfgs, noise, force_tokens,
                                  preds, toks)
                break
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
