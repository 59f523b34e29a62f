"""True-dimension parity cases: the HIP path vs this oracle at the sizes the headline benchmark launches.

TEST INFRASTRUCTURE (see oracle/__init__.py): called by tests/test_gpu_true_dims.py (which asserts the bounds) and by
the ``cpu_baseline`` leg of bench.py (which times the oracle side as the bounded CPU sample and prints the errors next
to the throughput line).  Nothing under bitdance_amd/ imports this.

The tiny-model goldens pin the *algorithm*; what they cannot show is that the kernel instantiations picked at
BitDance-14B dimensions -- the 9-wave ragged adaLN tile, 640-thread row kernels, 40-head attention, GQA group 5 with 56 q/k/v
slots -- compute the same thing.  Each case here builds a model slice at true width with seeded random weights
(values bf16-representable, so the CPU oracle and the device see identical parameters), runs

  * the diffusion head: one ``TransEncoder.forward`` (flow_head_parallel_x.py:325-342) of ``depth`` blocks and ``nada``
    adaLN projections at width D on M = branches*B*P rows  -> x_hat [M, C]
  * the LLM: one ``Qwen3Model.forward`` decode call (HF modeling_qwen3.py:367-427; t2i_pipeline.py:261-268) of ``layers``
    layers at true width over P new tokens per sequence against ``past`` cached tokens -> last_hidden_state

on the device through the C ABI (engine.Engine) and on the CPU through oracle.diff_head / oracle.qwen3 under
``Policy("autocast")``, and returns the error statistics plus the CPU time of the oracle side.
"""
from __future__ import annotations

import math
import time

import torch

from . import diff_head, qwen3, tiny_models as tm
from .numerics import Policy

HEAD_14B = dict(ch_target=32, ch_cond=5120, ch_latent=5120)           # train/configs/bitdance_14b_64x.yaml:22-33
QWEN3_14B = dict(hidden_size=5120, num_attention_heads=40, num_key_value_heads=8, head_dim=128,
                 intermediate_size=17408, vocab_size=64, rms_norm_eps=1e-6, rope_theta=1000000.0)


def _timed(fn, repeats: int):
    """Run ``fn`` ``repeats`` times; returns (last result, [seconds per run]) -- SURVEY 8(d) asks for >= 3 repeats per CPU leg."""
    out, ts = None, []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        with torch.no_grad():
            out = fn()
        ts.append(time.perf_counter() - t0)
    return out, ts


def _tstats(ts: list) -> dict:
    s = sorted(ts)
    return {"t_cpu_s": s[0], "t_cpu_median_s": s[len(s) // 2], "t_cpu_runs_s": [round(t, 4) for t in ts]}


def device_seeded_state(shapes: dict, seed: int, device, gain: float = 1.0) -> dict:
    """tiny_models.seeded_state's distribution (matrices N(0, gain/sqrt(fan_in)), norm scales 1+0.1N, biases 0.1N, all
    rounded to bf16) drawn on the device: 0.8 B parameters take seconds instead of minutes.  Returns fp32 tensors on
    ``device`` holding bf16-representable values."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        x = torch.empty(shp, dtype=torch.float32, device=device).normal_(generator=g)
        if len(shp) >= 2:
            x = x * (gain / math.sqrt(math.prod(shp[1:])))
        elif name.endswith("bias"):
            x = x * 0.1
        else:
            x = 1.0 + 0.1 * x
        out[name] = x.to(torch.bfloat16).to(torch.float32)
    return out


def head_case(device="cuda", *, D=5120, Dz=None, C=32, P=64, B=1, branches=2, depth=2, nada=2, head_dim=128,
              sigmoid=True, n_steps=3, eval_index=1, seed=101, tune: dict | None = None, weights: str = "bf16",
              mlp: bool = False, repeats: int = 1) -> dict:
    """One head evaluation at width D: device x_hat vs oracle x_hat.  eval_index > 0 exercises a non-zero timestep
    embedding; the latent is a fixed random tensor written straight into the engine's state.  ``mlp``: the MLP head of the
    1x ImageNet models (imagenet_gen/src/diff_head.py:228-253) instead of the transformer head."""
    from bitdance_amd import engine as E
    cfgd = dict(ch_target=C, ch_cond=Dz or D, ch_latent=D, depth_latent=depth, depth_adanln=nada)
    sd_dev = device_seeded_state((tm.mlp_head_shapes if mlp else tm.head_shapes)(cfgd), seed, device)
    hw = E.HeadWeights.from_state_dict(sd_dev, device, head_dim=head_dim, final_sigmoid=sigmoid, weights=weights)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    eng = E.Engine(hw, None, None, num_images=B, branches=branches, device=device, max_tokens=P, parallel_num=P,
                   tune=tune)
    M = branches * B * P
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(branches * B, P, cfgd["ch_cond"], generator=g)
    x = torch.randn(B, P, C, generator=g)
    eng.set_schedule(n_steps, 3.0, 1)
    eng.load_noise(torch.zeros(1, n_steps + 1, B, P, C))
    eng.reset([0] * min(branches * B, 16))
    eng.set_int("rt.dump_xhat", 1)
    eng.set_cond(z.to(device))
    eng.view("head.xt", torch.float32, (B * P, C)).copy_(x.reshape(B * P, C).to(device))
    eng.head_cond()
    eng.head_eval(eval_index)
    torch.cuda.synchronize()
    xhat = eng.view("head.xhat", torch.float32, (eng.Mpad, C))[:M].cpu().view(branches * B, P, C)
    t_i = float(eng._sc[eval_index, 0])
    comb = torch.cat([x] * branches)
    def cpu():
        if mlp:
            return diff_head.mlp_net_forward(sd, comb, torch.full((branches * B,), t_i), z, Policy("autocast")).float()
        return diff_head.net_forward(sd, comb, torch.full((branches * B,), t_i), z, Policy({"fp8": "fp8w", "fp8a": "fp8wa"}.get(weights, "autocast")),
                                     final_sigmoid=sigmoid, head_dim=head_dim).float()
    ref, ts = _timed(cpu, repeats)
    err = (xhat - ref).abs()
    extra = {}
    if weights in ("fp8", "fp8a"):                          # how far the fp8 modes are from the bf16 reference flow
        with torch.no_grad():
            ref16 = diff_head.net_forward(sd, comb, torch.full((branches * B,), t_i), z, Policy("autocast"),
                                          final_sigmoid=sigmoid, head_dim=head_dim).float()
        e16 = (xhat - ref16).abs()
        extra = {"vs_bf16_max": e16.max().item(), "vs_bf16_mean": e16.mean().item()}
    cfgs = {n: eng.gemm_config("head." + n) for n in (("ada", "w1", "w2") if mlp else ("ada", "qkv", "wo", "w1", "w2"))}
    macs_per_row = (D * C + D * cfgd["ch_cond"] + (nada * 6 + 2) * D * D + depth * (3 * D * D + D * D + 3 * D * D + 1.5 * D * D) + D * C)
    return {"max_err": err.max().item(), "mean_err": err.mean().item(), "ref_abs_mean": ref.abs().mean().item(),
            "finite": bool(torch.isfinite(xhat).all()), **_tstats(ts), "rows": M, "macs_per_row": macs_per_row,
            "gemm_cfg": {k: {"splitk": s, "nwaves": c & 15} for k, (s, c) in cfgs.items()}, "t": t_i, **extra}


def head_sample_case(device="cuda", *, D=5120, C=32, P=64, B=1, depth=6, nada=2, head_dim=128, n_steps=4, cfg=1.25, seed=131,
                     tune: dict | None = None, weights: str = "bf16", fp32_floor: bool = False) -> dict:
    """``DiffHead.sample`` at true width: n_steps + 1 CHAINED evaluations of the ``depth``-block head with classifier-free
    guidance ``cfg`` (sampling_x.py:44-97 driving flow_head_parallel_x.py:325-342), device vs oracle on the same noise.  At a
    guidance scale near 1 the chain is contractive, so the bound is a statement about the implementations, not about chaos
    (tests/test_gpu_chain_parity.py makes the same point on the tiny model)."""
    from bitdance_amd import engine as E
    cfgd = dict(ch_target=C, ch_cond=D, ch_latent=D, depth_latent=depth, depth_adanln=nada)
    sd_dev = device_seeded_state(tm.head_shapes(cfgd), seed, device)
    hw = E.HeadWeights.from_state_dict(sd_dev, device, head_dim=head_dim, weights=weights)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    branches = 2 if cfg > 1.0 else 1
    eng = E.Engine(hw, None, None, num_images=B, branches=branches, device=device, max_tokens=P, parallel_num=P, tune=tune)
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(branches * B, P, D, generator=g)
    noise = torch.randn(n_steps + 1, B, P, C, generator=g)
    eng.set_schedule(n_steps, cfg, 1)
    eng.load_noise(noise.view(1, n_steps + 1, B, P, C))
    eng.reset([0] * min(branches * B, 16))
    eng.set_cond(z.to(device))
    eng.head_sample()
    torch.cuda.synchronize()
    pred = eng.pred().cpu()
    tok = eng.tok_cur().cpu()
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = diff_head.sample(sd, z, cfg, n_steps, list(noise), Policy({"fp8": "fp8w", "fp8a": "fp8wa"}.get(weights, "autocast")))[:B]
    t_cpu = time.perf_counter() - t0
    err = (pred - ref).abs()
    agree = (torch.sign(pred) == torch.sign(ref)).float().mean().item()
    out = {"max_err": err.max().item(), "mean_err": err.mean().item(), "ref_abs_mean": ref.abs().mean().item(),
           "finite": bool(torch.isfinite(pred).all()), "tokens_are_sign_of_pred": bool(torch.equal(tok, torch.sign(pred))),
           "token_agreement": agree, "evaluations": n_steps + 1, "t_cpu_s": t_cpu}
    if fp32_floor:
        # the noise floor the device is judged against: how far the REFERENCE'S OWN bf16-autocast arithmetic lands from exact (fp32)
        # arithmetic after the same chain on the same noise -- two bf16 implementations cannot be expected to agree better than that
        t0 = time.perf_counter()
        with torch.no_grad():
            ref32 = diff_head.sample(sd, z, cfg, n_steps, list(noise), Policy("fp32"))[:B]
        e_floor, e_dev32 = (ref - ref32).abs(), (pred - ref32).abs()
        out.update({"floor_max_err": e_floor.max().item(), "floor_mean_err": e_floor.mean().item(),
                    "floor_token_agreement": (torch.sign(ref) == torch.sign(ref32)).float().mean().item(),
                    "dev_vs_fp32_max_err": e_dev32.max().item(), "dev_vs_fp32_mean_err": e_dev32.mean().item(),
                    "dev_vs_fp32_token_agreement": (torch.sign(pred) == torch.sign(ref32)).float().mean().item(),
                    "t_cpu_fp32_s": time.perf_counter() - t0})
    return out


def ae_case(device="cuda", *, config: str = "AE_D16C32", px: int = 256, seed: int = 5, repeats: int = 1) -> dict:
    """The tokenizer's conv decoder (autoencoder.py:169-196) at its released channel counts on a ``px`` x ``px`` image: the
    native kernels (bitdance_amd/ae_native.py) vs oracle/autoencoder.py under the autocast policy, and the oracle's CPU time --
    SURVEY 8(d)'s ``t_ae256`` (a 1024-pixel decode is 16 such tiles of work: the network is fully convolutional)."""
    from bitdance_amd import synthetic as syn
    from bitdance_amd.ae_native import NativeDecoder
    from bitdance_amd.autoencoder import VQModel
    from . import autoencoder as oae
    cfg = getattr(syn, config)
    ae = VQModel(**cfg).eval()
    ae.load_state_dict(syn.random_ae_state(cfg, device), strict=True, assign=True)
    ae.to(device)
    nat = NativeDecoder(ae.decoder, device)
    sd = {k: v.detach().float().cpu() for k, v in ae.state_dict().items()}
    down = 2 ** (len(cfg["ddconfig"]["ch_mult"]) - 1)
    zc = cfg["ddconfig"]["z_channels"]
    z = torch.sign(torch.randn(1, zc, px // down, px // down, generator=torch.Generator().manual_seed(seed)))
    got = nat.decode(z.to(device)).float().cpu()
    ref, ts = _timed(lambda: oae.decoder_forward(Policy("autocast"), sd, cfg["ddconfig"], z).float(), repeats)
    err = (got - ref).abs()
    return {"max_err": err.max().item(), "mean_err": err.mean().item(), "ref_abs_mean": ref.abs().mean().item(),
            "finite": bool(torch.isfinite(got).all()), **_tstats(ts), "px": px}


def llm_case(device="cuda", *, layers=1, P=64, past=(1000, 1017), cfg: dict | None = None, seed=202,
             tune: dict | None = None, weights: str = "bf16", repeats: int = 1) -> dict:
    """One native decode step (P new tokens per sequence, ragged cache lengths) at Qwen3-14B width vs the oracle.
    The K/V cache is filled with seeded random post-RoPE keys / values on both sides."""
    from bitdance_amd import engine as E
    c = dict(cfg or QWEN3_14B, num_hidden_layers=layers)
    D, nh, nkv, hd = c["hidden_size"], c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"]
    sd_dev = {k: v.to(torch.bfloat16) for k, v in device_seeded_state(tm.llm_shapes(c), seed, device).items()}
    lw = E.LlmWeights.from_state_dict(sd_dev, c, device, keep_for_prefill=False, weights=weights)
    w = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    nseq = len(past)
    eng = E.Engine(None, None, lw, num_images=nseq, branches=1, device=device, max_tokens=P, max_kv=max(past) + 2 * P,
                   parallel_num=P, tune=tune)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(nseq, P, D, generator=g)
    kc = eng.ws["llm.k_cache"].view(torch.bfloat16).view(layers, nseq, nkv, eng.Lmax, hd)
    vc = eng.ws["llm.vt_cache"].view(torch.bfloat16).view(layers, nseq, nkv, hd, eng.Lmax)
    caches = []
    for b, L in enumerate(past):
        per = []
        for li in range(layers):
            k = torch.randn(1, nkv, L, hd, generator=g).to(torch.bfloat16)
            v = torch.randn(1, nkv, L, hd, generator=g).to(torch.bfloat16)
            kc[li, b, :, :L] = k[0].to(device)
            vc[li, b, :, :, :L] = v[0].transpose(1, 2).to(device)
            per.append([k, v])
        caches.append(per)
    eng.set_int("rt.emit_cond", 0)
    eng.reset(list(past))
    eng.residual()[: nseq * P].copy_(x.reshape(nseq * P, D).to(device))
    eng.llm_step()
    torch.cuda.synchronize()
    got = eng.hidden().cpu().view(nseq, P, D)
    pol = Policy({"fp8": "fp8w", "fp8a": "fp8wa"}.get(weights, "autocast"))
    def cpu():
        refs = []
        for b, L in enumerate(past):
            ones = torch.ones(1, 1, P, L + P, dtype=torch.bool)
            o, _ = qwen3.model_forward(w, c, x[b:b + 1], list(caches[b]), ones, pol)     # (the forward replaces the list's entries)
            refs.append(o.float())
        return torch.cat(refs)
    ref, ts = _timed(cpu, repeats)
    err = (got - ref).abs()
    cfgs = {n: eng.gemm_config("llm." + n) for n in ("qkv", "o", "gu", "down")}
    return {"max_err": err.max().item(), "mean_err": err.mean().item(), "ref_abs_mean": ref.abs().mean().item(),
            "finite": bool(torch.isfinite(got).all()), **_tstats(ts), "rows": nseq * P, "layers": layers,
            "gemm_cfg": {k: {"splitk": s, "nwaves": c_ & 15} for k, (s, c_) in cfgs.items()}}


def ar_step_case(device="cuda", *, D=5120, C=32, P=64, depth=6, nada=2, n_steps=8, cfg=1.25, past=(1000, 1017), seed=151) -> dict:
    """ONE autoregressive step across the head -> LLM -> head seam at true width (t2i_pipeline.py:241-270): the condition of a patch
    -> ``DiffHead.sample`` (n_steps + 1 chained evaluations of the ``depth``-block head, guidance ``cfg``) -> ``sign`` -> the
    2-layer projector (+ 2-D position embedding) -> ONE Qwen3-14B decoder layer + final norm over the 64 new tokens of the cond and
    the uncond sequence against ragged caches -> the NEXT patch's condition (hidden + position embedding), device (engine: head
    sample graph phase, projector, LLM step -- the two phases of an AR step) vs oracle on identical noise.  Reported: the sampled
    latent's error and token agreement; the next condition's error with the oracle fed the DEVICE's tokens (numerics of projector +
    layer alone: a flipped token is a legitimate +-2 in one projector input, not an arithmetic error) and fed its own tokens."""
    from bitdance_amd import engine as E
    from . import pipeline as opipe
    B, branches = 1, 2
    cfgd = dict(ch_target=C, ch_cond=D, ch_latent=D, depth_latent=depth, depth_adanln=nada)
    sd_h = device_seeded_state(tm.head_shapes(cfgd), seed, device)
    hw = E.HeadWeights.from_state_dict(sd_h, device)
    head_w = {k: v.cpu() for k, v in sd_h.items()}
    del sd_h
    sd_p = device_seeded_state(tm.proj_shapes(C, D), seed + 1, device)
    pw = E.ProjWeights.from_state_dict(sd_p, device)
    proj_w = {k: v.cpu() for k, v in sd_p.items()}
    del sd_p
    c = dict(QWEN3_14B, num_hidden_layers=1)
    nkv, hd = c["num_key_value_heads"], c["head_dim"]
    sd_l = {k: v.to(torch.bfloat16) for k, v in device_seeded_state(tm.llm_shapes(c), seed + 2, device).items()}
    lw = E.LlmWeights.from_state_dict(sd_l, c, device, keep_for_prefill=False)
    llm_w = {k: v.cpu() for k, v in sd_l.items()}
    del sd_l
    eng = E.Engine(hw, pw, lw, num_images=B, branches=branches, device=device, max_tokens=2 * P, max_kv=max(past) + 2 * P, parallel_num=P)
    g = torch.Generator().manual_seed(seed + 3)
    ps = int(P ** 0.5)
    pos = opipe.pos_embed_2d(opipe.sincos_1d(D // 2, 256), 2 * ps, 2 * ps, ps)[: 3 * P]       # patches 0 .. 2 of a 4-patch image
    eng.pos.copy_(pos.to(device))
    hid0 = torch.randn(branches * B, P, D, generator=g)                                        # the LLM's hidden state of the previous step
    cond = hid0 + pos[None, :P]
    noise = torch.randn(n_steps + 1, B, P, C, generator=g)
    kc = eng.ws["llm.k_cache"].view(torch.bfloat16).view(1, branches * B, nkv, eng.Lmax, hd)
    vc = eng.ws["llm.vt_cache"].view(torch.bfloat16).view(1, branches * B, nkv, hd, eng.Lmax)
    caches = []
    for b, L in enumerate(past):
        k = torch.randn(1, nkv, L, hd, generator=g).to(torch.bfloat16)
        v = torch.randn(1, nkv, L, hd, generator=g).to(torch.bfloat16)
        kc[0, b, :, :L] = k[0].to(device)
        vc[0, b, :, :, :L] = v[0].transpose(1, 2).to(device)
        caches.append([[k, v]])
    eng.set_schedule(n_steps, cfg, 1)
    eng.load_noise(noise.view(1, n_steps + 1, B, P, C))
    eng.reset(list(past))
    eng.set_cond(cond.to(device))
    eng.head_sample()
    eng.projector()
    eng.llm_step()
    torch.cuda.synchronize()
    pred = eng.pred().cpu()
    tok = eng.tok_cur().cpu()
    hidden = eng.hidden().cpu().view(branches * B, P, D)
    next_cond = hidden + pos[None, P:2 * P]
    pol = Policy("autocast")
    t0 = time.perf_counter()
    with torch.no_grad():
        ref_pred = diff_head.sample(head_w, cond, cfg, n_steps, list(noise), pol)[:B]
        ref_tok = torch.sign(ref_pred)

        def after(tokens):                                   # projector + position embedding + one decoder layer + final norm, per sequence
            x = opipe.projector(proj_w, torch.cat([tokens] * branches), pol) + pos[None, :P]
            outs = []
            for b, L in enumerate(past):
                ones = torch.ones(1, 1, P, L + P, dtype=torch.bool)
                o, _ = qwen3.model_forward(llm_w, c, x[b:b + 1], list(caches[b]), ones, pol)
                outs.append(o.float())
            return torch.cat(outs) + pos[None, P:2 * P]
        ref_next_forced = after(tok)
        ref_next_free = after(ref_tok)
    t_cpu = time.perf_counter() - t0
    e_pred = (pred - ref_pred).abs()
    e_f = (next_cond - ref_next_forced).abs()
    e_free = (next_cond - ref_next_free).abs()
    return {"pred_max_err": e_pred.max().item(), "pred_mean_err": e_pred.mean().item(),
            "token_agreement": (tok == ref_tok).float().mean().item(), "tokens_are_sign_of_pred": bool(torch.equal(tok, torch.sign(pred))),
            "next_cond_max_err": e_f.max().item(), "next_cond_mean_err": e_f.mean().item(), "next_cond_ref_abs_mean": ref_next_forced.abs().mean().item(),
            "next_cond_free_max_err": e_free.max().item(), "next_cond_free_mean_err": e_free.mean().item(),
            "finite": bool(torch.isfinite(next_cond).all()), "evaluations": n_steps + 1, "t_cpu_s": t_cpu}
