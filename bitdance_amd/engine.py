"""Host side of the native AR step: weight packing, context/workspace management, graph replay.

This module is plumbing around libbitdance_hip.so (device memory via torch tensors, streams, ctypes);
all arithmetic of the hot path runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import torch

from ._lib import BitDanceHipError, check, lib

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _bf16(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=BF16).contiguous()


def pack_linear(ws: list[torch.Tensor], device) -> torch.Tensor:
    """Stack nn.Linear weights [n_i, K] along N and re-pack them into MFMA-operand order (bd_gemm.hip)."""
    K = ws[0].shape[1]
    n = sum(w.shape[0] for w in ws)
    if K % 64 or any(w.shape[0] % 32 for w in ws):
        raise BitDanceHipError(f"pack_linear: unsupported shape N={[w.shape[0] for w in ws]} K={K}")
    out = torch.empty(n * K, dtype=BF16, device=device)
    row = 0
    for w in ws:
        src = _bf16(w, device)
        check(lib().bd_pack_weight(out.data_ptr(), src.data_ptr(), src.shape[0], K, row, n, _stream()), "bd_pack_weight")
        row += src.shape[0]
    torch.cuda.current_stream().synchronize()
    return out


def pack_swiglu(gate: torch.Tensor, up: torch.Tensor, device) -> torch.Tensor:
    F_, K = gate.shape
    if K % 64 or F_ % 32:
        raise BitDanceHipError(f"pack_swiglu: unsupported shape F={F_} K={K}")
    g, u = _bf16(gate, device), _bf16(up, device)
    out = torch.empty(2 * F_ * K, dtype=BF16, device=device)
    check(lib().bd_pack_weight_swiglu(out.data_ptr(), g.data_ptr(), u.data_ptr(), F_, K, _stream()), "bd_pack_weight_swiglu")
    torch.cuda.current_stream().synchronize()
    return out


FP8_MAX = 448.0                                      # largest finite OCP e4m3 value


def quantize_rows_fp8(w: torch.Tensor):
    """Per-output-channel symmetric quantisation to OCP e4m3: scale[n] = max|W[n, :]| / 448, q = fp8(W / scale).  Returns
    (uint8 view of the fp8 bytes [N, K], fp32 scales [N]).  The scale is computed from the bf16 weight values in fp32."""
    wf = w.detach().to(torch.bfloat16).float()
    s = (wf.abs().amax(dim=1) / FP8_MAX).clamp_min(1e-12)
    q = (wf / s[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), s.contiguous()


def pack_linear_fp8(ws: list[torch.Tensor], device, k64: bool = False):
    """pack_linear for the fp8 weight path: (packed e4m3 bytes, fp32 scales [N]) (csrc/bd_gemm8.hip).  ``k64``: the operand order
    of the fp8 x fp8 matrix pipe (weights of the GEMMs that also take fp8 activations, ``weights="fp8a"``)."""
    K = ws[0].shape[1]
    n = sum(w.shape[0] for w in ws)
    if K % 64 or any(w.shape[0] % 32 for w in ws):
        raise BitDanceHipError(f"pack_linear_fp8: unsupported shape N={[w.shape[0] for w in ws]} K={K}")
    out = torch.empty(n * K, dtype=torch.uint8, device=device)
    scales = []
    row = 0
    for w in ws:
        q, s_ = quantize_rows_fp8(w.to(device))
        check((lib().bd_pack_weight8k if k64 else lib().bd_pack_weight8)(out.data_ptr(), q.data_ptr(), q.shape[0], K, row, n, _stream()),
              "bd_pack_weight8")
        scales.append(s_)
        row += q.shape[0]
    torch.cuda.current_stream().synchronize()
    return out, torch.cat(scales).contiguous()


def pack_swiglu_fp8(gate: torch.Tensor, up: torch.Tensor, device, k64: bool = False):
    F_, K = gate.shape
    if K % 64 or F_ % 32:
        raise BitDanceHipError(f"pack_swiglu_fp8: unsupported shape F={F_} K={K}")
    qg, sg = quantize_rows_fp8(gate.to(device))
    qu, su = quantize_rows_fp8(up.to(device))
    out = torch.empty(2 * F_ * K, dtype=torch.uint8, device=device)
    check((lib().bd_pack_weight8k_swiglu if k64 else lib().bd_pack_weight8_swiglu)(out.data_ptr(), qg.data_ptr(), qu.data_ptr(), F_, K, _stream()),
          "bd_pack_weight8_swiglu")
    torch.cuda.current_stream().synchronize()
    scales = torch.stack([sg.view(-1, 16), su.view(-1, 16)], dim=1).reshape(-1).contiguous()      # packed row order
    return out, scales


def _put_linear(p: dict, key: str, ws: list, device, fp8: int, act8: bool = False) -> None:
    """``fp8``: 0 bf16, 1 fp8 weights, 2 fp8 weights + (for the GEMMs marked ``act8``: the ones a row kernel feeds) fp8 activations."""
    if fp8:
        p[key], p[key + "_s"] = pack_linear_fp8(ws, device, k64=(fp8 == 2 and act8))
    else:
        p[key] = pack_linear(ws, device)


def _put_swiglu(p: dict, key: str, gate, up, device, fp8: int, act8: bool = False) -> None:
    if fp8:
        p[key], p[key + "_s"] = pack_swiglu_fp8(gate, up, device, k64=(fp8 == 2 and act8))
    else:
        p[key] = pack_swiglu(gate, up, device)


def pack_swiglu_bias(bg: torch.Tensor, bu: torch.Tensor, device) -> torch.Tensor:
    return torch.stack([_bf16(bg, device).view(-1, 16), _bf16(bu, device).view(-1, 16)], dim=1).reshape(-1).contiguous()


def _fp8_flag(weights: str) -> int:
    """0 bf16; 1 "fp8": e4m3 weights, bf16 activations; 2 "fp8a": e4m3 weights everywhere + e4m3 activations (per-row scales) on the
    fp8 matrix pipe for the GEMMs fed by a row kernel (head adaLN / qkv / w1, LLM q/k/v and gate/up)."""
    modes = {"bf16": 0, "fp8": 1, "fp8a": 2}
    if weights not in modes:
        raise BitDanceHipError(f"weights must be one of {sorted(modes)}, not {weights!r}")
    return modes[weights]


def row_blocks(m: int) -> int:
    return 1 if m <= 32 else (2 if m <= 64 else 4 * ((m + 127) // 128))


# ---------------------------------------------------------------------------------------------------
@dataclass
class HeadWeights:
    """vision_head.safetensors (keys ``net.*``) packed for the native head."""
    D: int
    C: int
    Dz: int
    H: int
    nblocks: int
    nada: int
    head_dim: int = 128                             # 128: T2I heads (flow_head_parallel_x.py:227); 64: imagenet (diff_head_parallel.py:207)
    final_sigmoid: bool = True                      # 2*sigmoid(out)-1 (flow_head_parallel_x.py:342); imagenet head: identity
    mlp: bool = False                               # MlpEncoder of the 1x ImageNet models (imagenet_gen/src/diff_head.py:165-253)
    ptrs: dict = field(default_factory=dict)        # name -> tensor (kept alive)
    tp_size: int = 1
    wdtype: int = 0                                 # 1: fp8-e4m3 streamed weights
    time_w0: torch.Tensor = None
    time_b0: torch.Tensor = None
    time_w2: torch.Tensor = None
    time_b2: torch.Tensor = None

    @staticmethod
    def from_state_dict(sd: dict, device, head_dim: int = 128, final_sigmoid: bool = True, tp_rank: int = 0,
                        tp_size: int = 1, weights: str = "bf16") -> "HeadWeights":
        """``tp_size`` > 1: pack this rank's slices (tp.shard_head_state); D / H below stay the FULL widths.
        ``weights`` = "fp8": the streamed Linears (cond_embed, adaLN, wqkv, wo, w1, w2) are stored e4m3 + per-channel scales."""
        fp8 = _fp8_flag(weights)
        if tp_size > 1:
            from .tp import shard_head_state
            sd = shard_head_state(sd, tp_rank, tp_size, head_dim)
        g = lambda k: sd[k]
        D, C = g("net.input_proj.weight").shape
        Dz = g("net.cond_embed.weight").shape[1]
        nb = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("net.res_blocks."))
        na = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("net.ada_ln_blocks."))
        if "net.res_blocks.0.w1.weight" not in sd:
            raise BitDanceHipError("native head requires the SwiGLU variant (use_swiglu=True)")
        H = g("net.res_blocks.0.w2.weight").shape[1]             # this rank's SwiGLU features
        # MLP head: ResBlock = norm -> modulate -> w1 (SwiGLU) -> w2 -> gated residual, adaLN blocks of 3 chunks, no attention
        mlp = "net.res_blocks.0.attn.wqkv.weight" not in sd and "net.res_blocks.0.norm.weight" in sd
        if mlp and tp_size > 1:
            raise BitDanceHipError("the MLP head has no tensor-parallel form")
        if not mlp and (head_dim not in (64, 128) or D % head_dim):
            raise BitDanceHipError(f"native head: head_dim {head_dim} unsupported for D={D}")
        hw = HeadWeights(D=D, C=C, Dz=Dz, H=H * tp_size, nblocks=nb, nada=na, head_dim=head_dim, final_sigmoid=final_sigmoid,
                         mlp=mlp)
        hw.tp_size = tp_size
        hw.wdtype = int(fp8)
        p = hw.ptrs
        _put_linear(p, "head.cond_w", [g("net.cond_embed.weight")], device, fp8)
        p["head.cond_b"] = _bf16(g("net.cond_embed.bias"), device)
        p["head.in_w"] = _bf16(g("net.input_proj.weight"), device)
        p["head.in_b"] = _bf16(g("net.input_proj.bias"), device)
        ada_w = [g(f"net.ada_ln_blocks.{j}.weight") for j in range(na)] + [g("net.final_layer.ada_ln_modulation.weight")]
        ada_b = [g(f"net.ada_ln_blocks.{j}.bias") for j in range(na)] + [g("net.final_layer.ada_ln_modulation.bias")]
        _put_linear(p, "head.ada_w", ada_w, device, fp8, act8=True)
        p["head.ada_b"] = torch.cat([_bf16(b, device) for b in ada_b]).contiguous()
        nada_cols = p["head.ada_b"].numel()
        if tp_size > 1 and not mlp and nada_cols % tp_size == 0 and (nada_cols // tp_size) % 256 == 0:
            # tensor parallel: ALSO this rank's contiguous slice of the stacked projection's output columns (SURVEY 8e "column-split
            # ada_ln_blocks"): the engine runs the grouped projection on it and all-gathers the modulation tensor when the communicator
            # offers a gather region (Engine: "tp.ada_split"); the whole matrix above stays for the modes that keep it replicated
            nl = nada_cols // tp_size
            full = torch.cat([_bf16(w_, device) for w_ in ada_w])
            _put_linear(p, "head.ada_w_l", [full[tp_rank * nl:(tp_rank + 1) * nl].contiguous()], device, fp8, act8=True)
            p["head.ada_b_l"] = p["head.ada_b"][tp_rank * nl:(tp_rank + 1) * nl].contiguous()
            del full
        for i in range(nb):
            s, d = f"net.res_blocks.{i}.", f"head.blk{i}."
            for n, src in ((("2", "norm"),) if mlp else (("1", "norm1"), ("2", "norm2"))):
                p[d + f"ln{n}_w"] = g(s + src + ".weight").detach().to(device, torch.float32).contiguous()
                p[d + f"ln{n}_b"] = g(s + src + ".bias").detach().to(device, torch.float32).contiguous()
            if not mlp:
                _put_linear(p, d + "wqkv", [g(s + "attn.wqkv.weight")], device, fp8, act8=True)
                p[d + "bqkv"] = _bf16(g(s + "attn.wqkv.bias"), device)
                _put_linear(p, d + "wo", [g(s + "attn.wo.weight")], device, fp8)
                p[d + "bo"] = _bf16(g(s + "attn.wo.bias"), device)
            w1, b1 = g(s + "w1.weight"), g(s + "w1.bias")
            _put_swiglu(p, d + "w1", w1[:H], w1[H:], device, fp8, act8=True)
            p[d + "b1"] = pack_swiglu_bias(b1[:H], b1[H:], device)
            _put_linear(p, d + "w2", [g(s + "w2.weight")], device, fp8)
            p[d + "b2"] = _bf16(g(s + "w2.bias"), device)
        p["head.lin_w"] = _bf16(g("net.final_layer.linear.weight"), device)
        p["head.lin_b"] = _bf16(g("net.final_layer.linear.bias"), device)
        hw.time_w0 = _bf16(g("net.time_embed.mlp.0.weight"), device)
        hw.time_b0 = _bf16(g("net.time_embed.mlp.0.bias"), device)
        hw.time_w2 = _bf16(g("net.time_embed.mlp.2.weight"), device)
        hw.time_b2 = _bf16(g("net.time_embed.mlp.2.bias"), device)
        return hw

    def ints(self) -> dict:
        return {"head.D": self.D, "head.C": self.C, "head.Dz": self.Dz, "head.H": self.H,
                "head.nblocks": self.nblocks, "head.nada": self.nada, "head.dh": self.head_dim,
                "head.sigmoid": int(self.final_sigmoid), "head.variant": int(self.mlp)}

    def time_table(self, ts: torch.Tensor) -> torch.Tensor:
        """time_embed(t_i) for every eval of the schedule, bf16 [N+1, D]  (flow_head_parallel_x.py:12-27,140-143).
        Data independent, so it is computed once per schedule (torch ops, bf16 Linear flow) instead of per eval."""
        half = self.time_w0.shape[1] // 2
        dev = ts.device
        t = 1000.0 * ts.float()
        freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32, device=dev) / half)
        args = t[:, None] * freqs[None]
        f = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        h = torch.nn.functional.linear(f.to(BF16), self.time_w0, self.time_b0)
        h = torch.nn.functional.silu(h)
        return torch.nn.functional.linear(h, self.time_w2, self.time_b2).contiguous()


@dataclass
class ProjWeights:
    """projector.safetensors: fc1 [D,C], fc2 [D,D] (modeling/utils.py:9-20)."""
    D: int
    C: int
    ptrs: dict = field(default_factory=dict)
    wdtype: int = 0

    @staticmethod
    def from_state_dict(sd: dict, device, weights: str = "bf16") -> "ProjWeights":
        D, C_ = sd["fc1.weight"].shape
        pw = ProjWeights(D=D, C=C_)
        pw.wdtype = int(_fp8_flag(weights))
        pw.ptrs["proj.w1"] = _bf16(sd["fc1.weight"], device)
        pw.ptrs["proj.b1"] = _bf16(sd["fc1.bias"], device)
        _put_linear(pw.ptrs, "proj.w2", [sd["fc2.weight"]], device, pw.wdtype)
        pw.ptrs["proj.b2"] = _bf16(sd["fc2.bias"], device)
        return pw

    def ints(self) -> dict:
        return {"proj.D": self.D, "proj.C": self.C}


@dataclass
class LlmWeights:
    """HF Qwen3 checkpoint (``model.layers.*``) packed for the native step kernels, which run both the decode step and the
    once-per-image prefill (llm.prefill_native).  With ``keep_for_prefill=True`` the original-layout bf16 tensors are kept
    as well, only for the torch cross-check path (llm.prefill_block, ``native_prefill=False``)."""
    cfg: dict
    ptrs: dict = field(default_factory=dict)
    sd: dict = field(default_factory=dict)          # original-layout bf16 tensors on device (prefill)
    tp_size: int = 1
    tp_rank: int = 0
    wdtype: int = 0

    @staticmethod
    def from_state_dict(sd: dict, cfg: dict, device, keep_for_prefill: bool = True, tp_rank: int = 0,
                        tp_size: int = 1, weights: str = "bf16") -> "LlmWeights":
        """``tp_size`` > 1: pack (and keep, for the prefill) this rank's slices (tp.shard_llm_state); ``cfg`` stays the full model's."""
        if tp_size > 1:
            from .tp import shard_llm_state
            sd = shard_llm_state(sd, cfg, tp_rank, tp_size)
        lw = LlmWeights(cfg=dict(cfg))
        lw.tp_size, lw.tp_rank = tp_size, tp_rank
        fp8 = _fp8_flag(weights)
        lw.wdtype = int(fp8)
        p = lw.ptrs
        if cfg["head_dim"] != 128:
            raise BitDanceHipError("native LLM path requires head_dim == 128 (Qwen3)")
        for i in range(cfg["num_hidden_layers"]):
            s, d = f"model.layers.{i}.", f"llm.l{i}."
            q, k, v = (sd[s + f"self_attn.{n}_proj.weight"] for n in "qkv")
            _put_linear(p, d + "wqkv", [q, k, v], device, fp8, act8=True)
            _put_linear(p, d + "wo", [sd[s + "self_attn.o_proj.weight"]], device, fp8)
            _put_swiglu(p, d + "wgu", sd[s + "mlp.gate_proj.weight"], sd[s + "mlp.up_proj.weight"], device, fp8, act8=True)
            _put_linear(p, d + "wdown", [sd[s + "mlp.down_proj.weight"]], device, fp8)
            p[d + "in_norm"] = _bf16(sd[s + "input_layernorm.weight"], device)
            p[d + "post_norm"] = _bf16(sd[s + "post_attention_layernorm.weight"], device)
            p[d + "q_norm"] = _bf16(sd[s + "self_attn.q_norm.weight"], device)
            p[d + "k_norm"] = _bf16(sd[s + "self_attn.k_norm.weight"], device)
        p["llm.final_norm"] = _bf16(sd["model.norm.weight"], device)
        if keep_for_prefill:
            lw.sd = {k: _bf16(v, device) for k, v in sd.items() if k != "lm_head.weight"}
        else:
            lw.sd = {"model.embed_tokens.weight": _bf16(sd["model.embed_tokens.weight"], device)}
        return lw

    def ints(self, Lmax: int, splits: int) -> dict:
        c = self.cfg
        return {"llm.D": c["hidden_size"], "llm.L": c["num_hidden_layers"], "llm.nh": c["num_attention_heads"],
                "llm.nkv": c["num_key_value_heads"], "llm.F": c["intermediate_size"], "llm.head_dim": c["head_dim"],
                "llm.Lmax": Lmax, "llm.splits": splits}

    def rope_tables(self, max_pos: int, device):
        """fp32 cos/sin [max_pos, 128] as Qwen3RotaryEmbedding computes them (HF:124-137)."""
        hd = self.cfg["head_dim"]
        inv = 1.0 / (self.cfg["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float, device=device) / hd))
        fr = torch.arange(max_pos, device=device).float()[:, None] * inv[None, :]
        emb = torch.cat((fr, fr), dim=-1)
        return emb.cos().contiguous(), emb.sin().contiguous()


# ---------------------------------------------------------------------------------------------------
def sampler_scalars(n_steps: int, device, last_step: float = 0.05, time_shift: float = 1.0):
    """The data-independent scalars of DiffHead.sample, computed with the same torch ops on the same device as
    the reference's 0-dim tensors (sampling_x.py:3-4,33-41,62-95).  Returns (fp32 [N+1, 6] on CPU, ts [N+1])."""
    t_all = torch.linspace(0, 1 - last_step, n_steps + 1, device=device, dtype=torch.float32)
    inv = 1 / time_shift
    t_all = inv / (inv + (1 / t_all - 1) ** 1.0)
    dt = t_all[1:] - t_all[:-1]
    t = torch.tensor(0.0, device=device, dtype=torch.float32)
    rows, ts = [], []
    for i in range(n_steps):
        sigma = 1 - t
        var = sigma ** 2 - (t / 1) * -1 * sigma
        den = (1 - t).clamp_min(0.05)
        ns = (2.0 * (1.0 - t) * dt[i]) ** 0.5
        rows.append(torch.stack([t, dt[i], den, var, 1 - t, ns]))
        ts.append(t.clone())
        t = t + dt[i]
    tf = torch.full((), 1 - last_step, device=device, dtype=torch.float32)
    zero = torch.zeros((), device=device)
    rows.append(torch.stack([tf, torch.full((), last_step, device=device, dtype=torch.float32),
                             (1 - tf).clamp_min(0.05), zero, 1 - tf, zero]))
    ts.append(tf)
    return torch.stack(rows).float().cpu().contiguous(), torch.stack(ts)


class Engine:
    """One native context for a fixed (num_images, CFG on/off): head + projector + LLM step, workspaces,
    and the two hipGraphs of an AR step."""

    def __init__(self, head: HeadWeights | None, proj: ProjWeights | None, llm: LlmWeights | None, *,
                 num_images: int, branches: int, device, max_tokens: int = 64, max_kv: int = 256,
                 attn_splits: int = 8, tune: dict | None = None, parallel_num: int = 64, comm=None,
                 extra_ints: dict | None = None):
        """``comm``: a tp.TPComm -- this engine is then rank comm.rank of a tensor-parallel group and ``head`` / ``llm`` must
        have been packed with the same (tp_rank, tp_size)."""
        self.l = lib()
        self.comm = comm if (comm is not None and comm.size > 1) else None
        for w in (head, llm):
            if w is not None and getattr(w, "tp_size", 1) != (self.comm.size if self.comm else 1):
                raise BitDanceHipError("weights were packed for a different tensor-parallel size than the engine's communicator")
        self.device = torch.device(device)
        self.head, self.proj, self.llm = head, proj, llm
        if parallel_num not in (1, 4, 16, 64):
            raise BitDanceHipError("parallel_num must be 64 / 16 (T2I 64x / 16x, ImageNet 16x), 4 or 1 (ImageNet 4x / 1x; the full-causal T2I loop)")
        self.B, self.branches, self.P = num_images, branches, parallel_num
        self.BP = self.B * self.P
        self.M = self.branches * self.BP
        self.max_tokens = max_tokens
        self.Lmax = ((max_kv + 63) // 64) * 64
        self._keep: dict[str, torch.Tensor] = {}
        self._sched_key = None
        self._cfg = None
        self._captured: set = set()
        wd = {getattr(w, "wdtype", 0) for w in (head, proj, llm) if w is not None}
        if len(wd) > 1:
            raise BitDanceHipError("head / projector / LLM weights must all be bf16 or all be fp8")
        self.wdtype = wd.pop() if wd else 0
        self.ctx = self.l.bd_ctx_create()
        ints = {"B": self.B, "branches": self.branches, "P": self.P, "wdtype": self.wdtype}
        if head is not None:
            ints.update(head.ints())
            ints["head.T"] = max_tokens
        if proj is not None:
            ints.update(proj.ints())
        if llm is not None:
            ints.update(llm.ints(self.Lmax, attn_splits))
        for k, v in (tune or {}).items():
            ints["tune." + k] = v
        ints.update(extra_ints or {})
        # tensor parallel on the hand-written exchange: column-split adaLN projection + push all-gather when this rank holds its
        # slice and the communicator's gather region takes one group's modulation tensor (G evaluations x Mpad rows x all columns)
        self.ada_split = False
        # Default from tp = 4 up: a rank receives (tp - 1) / tp of the group's tensor over tp - 1 links, i.e. 73 MB / tp per link and
        # group -- at tp = 2 that is 36 MB over ONE link (more time than streaming the 0.73 GB locally saves), at tp = 4 / 8 18 / 9 MB.
        # "tp.ada_split" = 1 / 0 in ``extra_ints`` forces it either way (the in-process tests run it at tp = 2).
        if self.comm is not None and head is not None and "head.ada_w_l" in head.ptrs and self.comm.backend in ("ipc", "none") \
                and ints.get("tp.ada_split", 1 if self.comm.size >= 4 else 0):
            mp = 32 if self.M <= 32 else (64 if self.M <= 64 else (self.M + 127) // 128 * 128)
            G = ints.get("tune.ada_group", 512 // mp if mp <= 128 else (1024 // mp if (mp <= 512 and 1024 % mp == 0) else 1))
            need = 2 * G * mp * head.ptrs["head.ada_b"].numel() * 2          # two slots (double-buffered by group parity)
            self.ada_split = G >= 2 and self.comm.gather_bytes >= need and (self.wdtype in (0, 2))
        ints["tp.ada_split"] = int(self.ada_split)
        # sequence-parallel row kernels (csrc/bd_sp.hip): every rank owns rows / tp rows of the head's residual stream and no stand-alone
        # exchange kernel is left in an evaluation.  Default wherever the form applies: the hand-written exchange with an operand landing
        # buffer, 128 rows (one image, 64-token patches), the cond / uncond rows of a patch position on one rank, bf16 activations.
        # "tp.seq" = 0 in ``extra_ints`` keeps the all-reduce form.  Ranks that share ONE physical GPU (in-process tests, several processes
        # on a single-GPU box) wait in a one-workgroup kernel in front of the consuming GEMM ("tune.sp_wait" = 0): a chip full of polling
        # GEMM workgroups would starve the peer rank's row kernel they are waiting for (observed: two 14B ranks on one GPU time out).
        self.seq_parallel = False
        if self.comm is not None and head is not None and not head.mlp and self.comm.backend in ("ipc", "none") and self.comm.hbuf_bytes > 0:
            # (a node whose self-test only passed WITH system-scope fences around the flags keeps the all-reduce form: the sequence-parallel
            # kernels implement the fence-less hand-off only)
            ok = (self.M == 128 and self.branches == 2 and (self.BP // 8) % self.comm.size == 0 and self.wdtype in (0, 1)
                  and self.comm.hbuf_bytes >= 128 * head.D * 2 and not getattr(self.comm, "fences", 0))
            # measured on one rank in loop-back (profiles/r05_tp_rank_critical_path.log, us per evaluation, all-reduce form vs this):
            # tp 2 976 vs 910, tp 4 755 vs 743, tp 8 670 vs 702 -- on by default up to 4 ranks; at 8 the rank's 16 rows make every row
            # kernel a 16-workgroup latency chain either way and the two extra destinations per push cost more than the saved launch
            # ... and only behind the hand-off's own self-test when the ranks sit on DIFFERENT devices (TPComm.from_process_group sets
            # comm.sp_ok; ranks inside one process / loop-back share one L2 domain and leave it None): the landing buffer is cacheable
            # memory written by the peers, which one-GPU tests cannot vouch for.  An explicit "tp.seq" = 1 still needs
            # sp_ok not to be False.
            sp_ok = getattr(self.comm, "sp_ok", None)
            trusted = sp_ok is True or (sp_ok is None and (getattr(self.comm, "in_process_peers", False) or getattr(self.comm, "loopback", False)))
            ok = ok and sp_ok is not False
            self.seq_parallel = bool(ints.get("tp.seq", 1 if (self.comm.size <= 4 and trusted) else 0)) and ok
            if ints.get("tp.seq", 0) and not ok:
                raise BitDanceHipError("tp.seq: sequence-parallel row kernels need 128 rows (one image with CFG, parallel_num 64), whole 8-row "
                                       "groups of patch positions per rank, bf16 activations and a communicator with an operand landing buffer")
        ints["tp.seq"] = int(self.seq_parallel)
        # ... and the Qwen3 decode step on the same hand-off (csrc/bd_sp.hip rms_sp_kernel; "tp.llm_seq"): the same conditions and the
        # same default (the prefill, eager and once per image, keeps the all-reduce form inside the same context)
        self.llm_seq_parallel = False
        if self.comm is not None and llm is not None and self.comm.backend in ("ipc", "none") and self.comm.hbuf_bytes > 0:
            sp_ok = getattr(self.comm, "sp_ok", None)
            trusted = sp_ok is True or (sp_ok is None and (getattr(self.comm, "in_process_peers", False) or getattr(self.comm, "loopback", False)))
            okl = (self.M == 128 and (self.M // 8) % self.comm.size == 0 and self.wdtype in (0, 1) and llm.cfg["head_dim"] == 128
                   and self.comm.hbuf_bytes >= 128 * llm.cfg["hidden_size"] * 6 and not getattr(self.comm, "fences", 0) and sp_ok is not False)
            self.llm_seq_parallel = bool(ints.get("tp.llm_seq", 1 if (self.comm.size <= 4 and trusted) else 0)) and okl
            if ints.get("tp.llm_seq", 0) and not okl:
                raise BitDanceHipError("tp.llm_seq: the sequence-parallel Qwen3 step needs 128 rows in whole 8-row groups per rank, bf16 activations "
                                       "and a communicator with an operand landing buffer of rows x hidden x 6 bytes whose hand-off self-test passed")
        ints["tp.llm_seq"] = int(self.llm_seq_parallel)
        if (self.seq_parallel or self.llm_seq_parallel) and "tune.sp_wait" not in ints and getattr(self.comm, "shares_gpu", False):
            ints["tune.sp_wait"] = 0
        for k, v in ints.items():
            check(self.l.bd_ctx_set_int(self.ctx, k.encode(), int(v)))
        if llm is not None:
            check(self.l.bd_ctx_set_float(self.ctx, b"llm.eps", float(llm.cfg["rms_norm_eps"])))
        for w in (head, proj, llm):
            if w is not None:
                for k, t in w.ptrs.items():
                    self.set_ptr(k, t)
        if self.comm is not None:
            check(self.l.bd_ctx_set_comm(self.ctx, self.comm.h), "bd_ctx_set_comm")
        check(self.l.bd_ctx_finalize(self.ctx), "bd_ctx_finalize")
        self.ws: dict[str, torch.Tensor] = {}
        for i in range(self.l.bd_ctx_ws_count(self.ctx)):
            name = self.l.bd_ctx_ws_name(self.ctx, i).decode()
            nbytes = self.l.bd_ctx_ws_bytes(self.ctx, i)
            if name == "head.ada_bf" and self.ada_split:
                # the gathered modulation tensor lives in the communicator's exported region: the peers push their columns into it
                if int(nbytes) > self.comm.gather_bytes:
                    raise BitDanceHipError("the communicator's gather region is smaller than one group's adaLN modulation tensor")
                check(self.l.bd_ctx_set_ptr(self.ctx, name.encode(), self.comm.gather_ptr), "bd_ctx_set_ptr")
                continue
            t = torch.zeros(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)
            self.ws[name] = t
            self.set_ptr(name, t)
        if head is not None:
            self.tok_all = torch.zeros(self.B, max_tokens, head.C, dtype=torch.float32, device=self.device)
            self.set_ptr("head.tok_all", self.tok_all)
        if llm is not None:
            self.cos, self.sin = llm.rope_tables(self.Lmax, self.device)
            self.set_ptr("llm.cos", self.cos)
            self.set_ptr("llm.sin", self.sin)
        if proj is not None or llm is not None:
            D = proj.D if proj is not None else llm.cfg["hidden_size"]
            self.pos = torch.zeros(max_tokens + self.P, D, dtype=torch.float32, device=self.device)
            self.set_ptr("pos", self.pos)
        self.noise = None
        check(self.l.bd_ctx_bind(self.ctx), "bd_ctx_bind")

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.l.bd_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------------------------
    def set_ptr(self, key: str, t: torch.Tensor) -> None:
        self._keep[key] = t
        check(self.l.bd_ctx_set_ptr(self.ctx, key.encode(), t.data_ptr()))

    def set_int(self, key: str, v: int) -> None:
        check(self.l.bd_ctx_set_int(self.ctx, key.encode(), int(v)))

    def view(self, name: str, dtype, shape) -> torch.Tensor:
        n = math.prod(shape)
        return self.ws[name].view(dtype)[:n].view(*shape)

    @property
    def Mpad(self) -> int:
        return row_blocks(self.M) * 32

    # -- schedule / noise ---------------------------------------------------------------------------
    def set_schedule(self, n_steps: int, cfg: float, ar_steps: int, time_shift: float = 1.0) -> None:
        """Sampler scalars of DiffHead.sample (sampling_x.py:62-64; ``time_shift`` = the head config's, 1.0 in the
        released configs), the time-embedding table and the noise buffer."""
        key = (n_steps, ar_steps, float(time_shift))
        if self._sched_key == key:
            # same schedule, another guidance scale (the ImageNet linear ramp changes it every AR step): a host-side scalar --
            # no table / noise / workspace is rebuilt; graphs captured with the old value are dropped
            if float(cfg) != self._cfg:
                check(self.l.bd_head_set_cfg(self.ctx, float(cfg)), "bd_head_set_cfg")
                self._cfg = float(cfg)
                self._captured.clear()
            return
        self._cfg = float(cfg)
        sc, ts = sampler_scalars(n_steps, self.device, time_shift=float(time_shift))
        self._sc = sc
        check(self.l.bd_head_set_schedule(self.ctx, n_steps, sc.data_ptr(), float(cfg)), "bd_head_set_schedule")
        self.temb = self.head.time_table(ts)
        self.set_ptr("head.temb", self.temb)
        self.noise = torch.zeros(ar_steps, n_steps + 1, self.BP, self.head.C, dtype=torch.float32, device=self.device)
        self.set_ptr("head.noise", self.noise)
        # y_i = silu(time_embed(t_i) + cond_embed(c)) of every evaluation, produced once per AR step next to cond_embed
        # (capacity rounded up to whole groups of <= 64 evaluations: the adaLN projections of a group run as one GEMM whose
        # operand is the group's rows, bd_api.hip head_ada_group; zero-filled so the rows past the schedule are finite)
        cap = (n_steps + 1 + 63) // 64 * 64
        y_bytes = cap * self.Mpad * self.head.D * 2
        if y_bytes <= (512 << 20):
            self.y_all = torch.zeros(y_bytes // 2, dtype=BF16, device=self.device)
            self.set_ptr("head.y_all", self.y_all)
            if self.wdtype == 2:                             # fp8 activations: per-row scales of every evaluation's y
                self.y_scale_all = torch.zeros(cap * self.Mpad, dtype=torch.float32, device=self.device)
                self.set_ptr("head.y_scale_all", self.y_scale_all)
            self.set_int("head.y_evals", cap)
        else:
            self.set_int("head.y_evals", 0)
        self.n_steps = n_steps
        self._sched_key = key
        self._captured.clear()                       # pointers / scalars are baked into the graphs

    def draw_noise(self, ar_steps: int) -> None:
        """Draw every normal the reference would draw, with the same calls in the same order, from the global
        device generator (sampling_x.py:60 randn, :40 randn_like) -- identical values for an identical seed."""
        shp = (self.B, self.P, self.head.C)
        for s in range(ar_steps):
            x = torch.randn(shp, device=self.device)
            self.noise[s, 0] = x.view(self.BP, -1)
            for i in range(self.n_steps):
                self.noise[s, i + 1] = torch.randn_like(x).view(self.BP, -1)

    def load_noise(self, noise: torch.Tensor) -> None:
        """Injected noise [ar_steps, N+1, B, P, C] (tests / CPU-vs-GPU parity)."""
        self.noise[: noise.shape[0]].copy_(noise.reshape(noise.shape[0], noise.shape[1], self.BP, -1))

    def reset(self, kv_len: list[int]) -> None:
        arr = (C.c_int * len(kv_len))(*kv_len)
        check(self.l.bd_step_reset(self.ctx, arr, len(kv_len), _stream()), "bd_step_reset")

    # -- operators ------------------------------------------------------------------------------------
    def set_cond(self, z: torch.Tensor) -> None:
        """z [M, Dz] fp32 -> head.cond_frag (what cond_embed's autocast cast does, in operand layout)."""
        z = z.reshape(self.M, -1).to(torch.float32).contiguous()
        check(self.l.bd_rows_to_frag(self.ws["head.cond_frag"].data_ptr(), z.data_ptr(), 1, self.M, z.shape[1],
                                     row_blocks(self.M), _stream()), "bd_rows_to_frag")

    def head_sample(self) -> None:
        check(self.l.bd_head_sample(self.ctx, _stream()), "bd_head_sample")

    def head_cond(self) -> None:
        check(self.l.bd_head_cond(self.ctx, _stream()), "bd_head_cond")

    def head_eval(self, i: int) -> None:
        check(self.l.bd_head_eval(self.ctx, i, _stream()), "bd_head_eval")

    def projector(self) -> None:
        check(self.l.bd_projector(self.ctx, _stream()), "bd_projector")

    def llm_step(self) -> None:
        check(self.l.bd_llm_step(self.ctx, _stream()), "bd_llm_step")

    def capture(self, phase: int) -> None:
        if phase in self._captured:
            return
        torch.cuda.current_stream().synchronize()
        check(self.l.bd_graph_capture(self.ctx, phase, _stream()), "bd_graph_capture")
        self._captured.add(phase)

    def launch(self, phase: int) -> None:
        check(self.l.bd_graph_launch(self.ctx, phase, _stream()), "bd_graph_launch")

    # -- measurement ---------------------------------------------------------------------------------------
    def profile_gemms(self, fn) -> dict:
        """Run ``fn`` (eager launches) with every GEMM launch bracketed by HIP events on the launch stream;
        returns {gemm name: dict(count, ms, bytes)} (in-situ durations, weights not cache-resident)."""
        check(self.l.bd_prof_enable(self.ctx, 1))
        try:
            fn()
            torch.cuda.current_stream().synchronize()
            out: dict = {}
            name = C.create_string_buffer(64)
            ms, nb = C.c_float(), C.c_double()
            for i in range(self.l.bd_prof_count(self.ctx)):
                check(self.l.bd_prof_get(self.ctx, i, name, C.byref(ms), C.byref(nb)), "bd_prof_get")
                r = out.setdefault(name.value.decode(), dict(count=0, ms=0.0, bytes=0.0))
                r["count"] += 1
                r["ms"] += ms.value
                r["bytes"] += nb.value
            return out
        finally:
            check(self.l.bd_prof_enable(self.ctx, 0))

    def gemm_config(self, name: str) -> tuple[int, int]:
        s_, nw = C.c_int(), C.c_int()
        check(self.l.bd_gemm_config(self.ctx, name.encode(), C.byref(s_), C.byref(nw)), "bd_gemm_config")
        return s_.value, nw.value

    # convenient typed views of outputs
    def pred(self) -> torch.Tensor:
        return self.view("head.pred", torch.float32, (self.B, self.P, self.head.C))

    def tok_cur(self) -> torch.Tensor:
        return self.view("head.tok_cur", torch.float32, (self.B, self.P, self.head.C))

    def hidden(self) -> torch.Tensor:
        D = self.llm.cfg["hidden_size"]
        return self.view("llm.hidden", torch.float32, (self.Mpad, D))[: self.M]

    def residual(self) -> torch.Tensor:
        D = self.llm.cfg["hidden_size"]
        return self.view("llm.R", torch.float32, (self.Mpad, D))
