"""Parity of the HIP hot path (through the C ABI, libbitdance_hip.so) against the CPU oracle and the committed
reference golden vectors.  Needs a real MI355X:  pytest -m gpu

Tolerances (stated per test): integer/sign work and the fp32 sampler update are bit-exact given identical
inputs; bf16-flow operators are compared with the oracle's autocast policy at the bf16 noise level measured
between the oracle and the reference itself (tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import diff_head, pipeline as opipe, qwen3, sampler          # noqa: E402
from oracle import tiny_models as tm                                      # noqa: E402
from oracle.numerics import Policy                                        # noqa: E402

DEV = "cuda"


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].ndim > 0 else z[k]) for k in z.files}


@pytest.fixture(scope="module")
def eng_mod():
    from bitdance_amd import engine
    from bitdance_amd._lib import lib
    lib()                                            # fails loudly if the .so is missing
    return engine


# ----------------------------------------------------------------------------------------------- GEMM
def frag(eng_mod, x):
    """fp32 [M,K] -> fragment-major bf16 via the C ABI."""
    from bitdance_amd._lib import check, lib
    M, K = x.shape
    rb = eng_mod.row_blocks(M)
    out = torch.zeros(rb * 32 * K, dtype=torch.bfloat16, device=DEV)
    check(lib().bd_rows_to_frag(out.data_ptr(), x.contiguous().data_ptr(), 1, M, K, rb,
                                torch.cuda.current_stream().cuda_stream))
    return out, rb


@pytest.mark.parametrize("M,N,K,S,nw", [
    (128, 256, 256, 1, 4), (128, 256, 256, 4, 2), (128, 512, 384, 3, 8), (64, 256, 256, 2, 4),
    (32, 128, 192, 1, 2), (256, 256, 256, 2, 4), (256, 512, 384, 2, 8), (256, 5120, 5120, 6, 8), (512, 512, 384, 2, 8), (512, 1024, 5120, 1, 8), (128, 5120, 5120, 4, 4), (128, 15360, 5120, 2, 4),
    (128, 5120, 17408, 6, 4), (128, 7168, 5120, 3, 2),
    (128, 352, 256, 1, 5), (128, 640, 384, 2, 9), (128, 608, 256, 1, 9), (128, 608, 256, 1, 10 + 256),
    (128, 736, 384, 3, 10 + 256)])      # ragged last tiles (5- and 9-wave workgroups, 5 panels x 2 K-parts)
def test_gemm_partial(eng_mod, M, N, K, S, nw):
    """F.linear under bf16 autocast == sum of the split-K slabs (fp32 accumulation of bf16 products)."""
    from bitdance_amd._lib import check, lib
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    xf, rb = frag(eng_mod, x)
    wp = eng_mod.pack_linear([w], DEV)
    out = torch.full((S, rb * 32, N), float("nan"), device=DEV)
    check(lib().bd_gemm_partial(xf.data_ptr(), rb, wp.data_ptr(), N, K, S, nw, out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = out.sum(0)[:M]
    ref = x.to(torch.bfloat16).double() @ w.double().t()
    err = (got.double() - ref).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 + 1e-5, err              # fp32 accumulation-order noise only


@pytest.mark.parametrize("nw,kw,ring", [(4, 1, 2), (4, 1, 3), (4, 1, 4), (8, 1, 2), (8, 1, 3), (8, 1, 4), (2, 1, 2), (4, 2, 2), (4, 2, 3),
                                        (8, 2, 2), (2, 2, 2), (9, 1, 2), (10, 1, 2)])
def test_gemm_every_stage_count(eng_mod, nw, kw, ring):
    """Every K-stage count 1 .. 30 through every tile form / register-ring depth: the guarded tail of the K loop runs 1 .. U + R - 1
    phases.  Regression for a compiler hazard (bd_common.h BD_MFMA_DRAIN): when the LAST tail phase executed, its final MFMA was
    followed across a taken branch by the accumulator copies with too few wait states -- acc[3][15] (rows 27 / 31 of the last row
    block) stale at exactly 14 stages (ring 3, K-part tiles) or 15 / 27 (ring 4); the regression below runs every stage count."""
    from bitdance_amd._lib import check, lib
    M, N = 128, 32 * (nw // kw) * 2
    g = torch.Generator(device=DEV).manual_seed(100 * nw + 10 * kw + ring)
    st = torch.cuda.current_stream().cuda_stream
    for n in range(1, 31):
        K = 64 * kw * n
        x = torch.randn(M, K, device=DEV, generator=g)
        w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
        xf, rb = frag(eng_mod, x)
        wp = eng_mod.pack_linear([w], DEV)
        out = torch.full((1, rb * 32, N), float("nan"), device=DEV)
        check(lib().bd_gemm_partial(xf.data_ptr(), rb, wp.data_ptr(), N, K, 1, nw + 16 * ring + 256 * (kw - 1), out.data_ptr(), st))
        torch.cuda.synchronize()
        ref = x.to(torch.bfloat16).double() @ w.double().t()
        err = (out[0, :M].double() - ref).abs().max().item()
        assert err <= 2e-5 * K ** 0.5 + 1e-5, (n, err)


@pytest.mark.parametrize("M,F_,K,nw", [(128, 384, 256, 2), (128, 512, 256, 4), (64, 256, 256, 2), (128, 7680, 5120, 2),
                                       (256, 512, 256, 8), (512, 7680, 5120, 8), (128, 352, 256, 5), (128, 17408, 5120, 5),
                                       (128, 352, 256, 10 + 256), (128, 17408, 5120, 10 + 256)])
def test_gemm_swiglu(eng_mod, M, F_, K, nw):
    """Linear -> chunk -> silu(h1)*h2 with the reference's bf16 rounding points (flow_head:250-251)."""
    from bitdance_amd._lib import check, lib
    g = torch.Generator(device=DEV).manual_seed(F_ + K)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = (torch.randn(2 * F_, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = (torch.randn(2 * F_, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    xf, rb = frag(eng_mod, x)
    wp = eng_mod.pack_swiglu(w[:F_], w[F_:], DEV)
    bp = eng_mod.pack_swiglu_bias(b[:F_], b[F_:], DEV)
    act = torch.zeros(rb * 32 * F_, dtype=torch.bfloat16, device=DEV)
    check(lib().bd_gemm_swiglu(xf.data_ptr(), rb, wp.data_ptr(), bp.data_ptr(), 2 * F_, K, nw, act.data_ptr(),
                               torch.cuda.current_stream().cuda_stream))
    h = (x.to(torch.bfloat16).float() @ w.float().t() + b.float()).to(torch.bfloat16)
    ref = (torch.nn.functional.silu(h[:, :F_]) * h[:, F_:])
    # un-fragment: chunk (ks, rb) lane l holds rows rb*32+(l&31), k = ks*16+(l>>5)*8+j
    a = act.view(F_ // 16, rb, 2, 32, 8).permute(1, 3, 0, 2, 4).reshape(rb * 32, F_)[:M]
    d = (a.float() - ref.float()).abs()
    assert (d > 0).float().mean() <= 0.02 and d.max() <= 0.07, ((d > 0).float().mean(), d.max())   # <= 1 bf16 ulp flips


@pytest.mark.parametrize("ring,xcd", [(2, 1), (3, 0), (3, 1)])
@pytest.mark.parametrize("M,N,K,S", [(512, 1024, 5120, 1), (512, 2304, 384, 1), (512, 2304, 384, 2), (768, 2560, 448, 1)])
def test_gemm_wide_options(eng_mod, M, N, K, S, ring, xcd):
    """The 256-row kernel's measurement switches (weight ring depth, row tiles of a weight slice on one XCD) change the
    schedule and the block -> tile map, never the values: bit-identical to the default setting, and right."""
    from bitdance_amd._lib import check, lib
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    xf, rb = frag(eng_mod, x)
    wp = eng_mod.pack_linear([w], DEV)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        for r_, x_ in ((2, 0), (ring, xcd)):
            check(lib().bd_set_gemm_option(b"wide.ring", r_))
            check(lib().bd_set_gemm_option(b"wide.xcd", x_))
            out = torch.full((S, rb * 32, N), float("nan"), device=DEV)
            check(lib().bd_gemm_partial(xf.data_ptr(), rb, wp.data_ptr(), N, K, S, 8, out.data_ptr(), st))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        check(lib().bd_set_gemm_option(b"wide.ring", 2))
        check(lib().bd_set_gemm_option(b"wide.xcd", -1))
    assert torch.equal(outs[0], outs[1])
    ref = x.to(torch.bfloat16).double() @ w.double().t()
    err = (outs[1].sum(0)[:M].double() - ref).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 + 1e-5, err


@pytest.mark.parametrize("K,S", [(64, 1), (128, 1), (192, 1), (448, 1), (768, 1), (1024, 1), (5120, 1), (1024, 2), (2048, 3)])
def test_gemm_tile_kernel_both_fetch_forms(eng_mod, K, S):
    """The LDS-tiled 256 x 256 kernel (bd_gemm_tile.hip: from 1024 rows on N >= 4096) in both operand-fetch forms -- LDS-DMA
    (option "tile" = 3), register-staged global loads, three stages ahead in three register sets (= 2), and W straight into registers with
    only A through LDS (= 4, round 6) -- against the 256-row
    weight-streaming kernel (= 0): every accumulator sees its K steps in the same order through the same MFMA, so the fp32 slabs,
    the bf16(+bias) output and the fused SwiGLU operand are bit-identical; and right against fp64.  K covers 32-deep stage counts
    2 .. 160 with every residue mod 3 (the register sets rotate over whole triples; the last 1-3 stages run behind conditions)
    and split-K slices."""
    from bitdance_amd._lib import check, lib
    M, N = 1024, 4096
    g = torch.Generator(device=DEV).manual_seed(K + S)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = (torch.randn(N, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    xf, rb = frag(eng_mod, x)
    wp = eng_mod.pack_linear([w], DEV)
    ws = eng_mod.pack_swiglu(w[: N // 2], w[N // 2:], DEV)
    bs = eng_mod.pack_swiglu_bias(b[: N // 2], b[N // 2:], DEV)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for tile in (0, 3, 2, 4):
            check(lib().bd_set_gemm_option(b"tile", tile))
            slabs = torch.full((S, rb * 32, N), float("nan"), device=DEV)
            check(lib().bd_gemm_partial(xf.data_ptr(), rb, wp.data_ptr(), N, K, S, 8, slabs.data_ptr(), st))
            outs = [slabs]
            if S == 1:
                o16 = torch.zeros(rb * 32, N, dtype=torch.bfloat16, device=DEV)
                check(lib().bd_gemm_bf16(xf.data_ptr(), rb, wp.data_ptr(), b.data_ptr(), N, K, 1, 8, None, None, o16.data_ptr(), st))
                act = torch.zeros(rb * 32 * (N // 2), dtype=torch.bfloat16, device=DEV)
                check(lib().bd_gemm_swiglu(xf.data_ptr(), rb, ws.data_ptr(), bs.data_ptr(), N, K, 8, act.data_ptr(), st))
                outs += [o16, act]
            torch.cuda.synchronize()
            res[tile] = outs
    finally:
        check(lib().bd_set_gemm_option(b"tile", 1))
    for tile in (3, 2, 4):
        for a, c in zip(res[0], res[tile]):
            assert torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16),
                               c.view(torch.int32 if c.dtype == torch.float32 else torch.int16)), (tile, a.dtype)
    ref = x.to(torch.bfloat16).double() @ w.double().t()
    err = (res[2][0].sum(0)[:M].double() - ref).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 + 1e-5, err
    if S == 1:
        d = (res[2][1][:M].double() - (ref + b.double())).abs().max().item()
        assert d <= 0.05, d                                                  # one bf16 rounding of values of O(1)


# ----------------------------------------------------------------------------------------------- head
def tiny_head_engine(eng_mod, B=2, branches=2):
    sd = tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11)
    hw = eng_mod.HeadWeights.from_state_dict(sd, DEV)
    return sd, eng_mod.Engine(hw, None, None, num_images=B, branches=branches, device=DEV, max_tokens=64)


def test_head_eval_and_sampler_step(eng_mod, golden_dir):
    """One TransEncoder.forward (x_hat) vs the oracle's autocast flow, and the fp32 SDE update bit-exact
    given that x_hat (sampling_x.py:33-41 restated op for op)."""
    g = load(golden_dir, "head_amp")
    sd, eng = tiny_head_engine(eng_mod)
    n, cfg = 3, 2.5
    eng.set_schedule(n, cfg, 1)
    eng.load_noise(g["noise"].view(1, n + 1, 2, 64, 32))
    eng.reset([0, 0, 0, 0])
    eng.set_int("rt.dump_xhat", 1)
    eng.set_cond(g["z"].to(DEV))
    # eval 0 by hand: latent = first draw
    xt = eng.view("head.xt", torch.float32, (128, 32))
    xt.copy_(g["noise"][0].reshape(128, 32))
    x_before = xt.clone().cpu()
    eng.head_cond()
    eng.head_eval(0)
    torch.cuda.synchronize()
    xhat = eng.view("head.xhat", torch.float32, (eng.Mpad, 32))[:256].cpu().view(4, 64, 32)
    t0 = torch.zeros(4)
    comb = torch.cat([x_before.view(2, 64, 32)] * 2)
    ref = diff_head.net_forward(sd, comb, t0, g["z"], Policy("autocast")).float()
    err = (xhat - ref).abs()
    assert err.max() <= 5e-2 and err.mean() <= 6e-3, (err.max(), err.mean())
    # sampler step on the device's own x_hat, with the engine's scalar table: bit exact fp32, op for op
    t, dt, den, var, omt, ns = (eng._sc[0, j] for j in range(6))
    x = x_before.view(2, 64, 32)
    v = (xhat - comb) / den
    vc, vu = v.chunk(2)
    v = vu + cfg * (vc - vu)
    score = (t * v - x) / var
    drift = v + omt * score
    want = x + drift * dt + ns * g["noise"][1]
    assert torch.equal(xt.cpu().view(2, 64, 32), want)
    # and the table itself equals the oracle's restatement of the reference scalars (to the last ulp or one)
    ts, dts = sampler.step_table(n)
    assert abs(float(t) - float(ts[0])) <= 1e-7 and abs(float(dt) - float(dts[0])) <= 1e-7


@pytest.mark.parametrize("cfg,branches", [(2.5, 2), (1.0, 1)])
def test_head_sample_vs_oracle(eng_mod, golden_dir, cfg, branches):
    """DiffHead.sample end to end (N=3) vs oracle and vs the reference's own output (golden head_amp)."""
    g = load(golden_dir, "head_amp")
    sd, eng = tiny_head_engine(eng_mod, B=2, branches=branches)
    n = 3
    z = g["z"][: 2 * branches]
    eng.set_schedule(n, cfg, 1)
    eng.load_noise(g["noise"].view(1, n + 1, 2, 64, 32))
    eng.reset([0] * (2 * branches))
    eng.set_cond(z.to(DEV))
    eng.head_sample()
    torch.cuda.synchronize()
    pred = eng.pred().cpu()
    ref = diff_head.sample(sd, z, cfg, n, list(g["noise"]), Policy("autocast"))[:2]
    err = (pred - ref).abs()
    amp = 1.0 if cfg <= 1.0 else (2 * cfg - 1)                # CFG amplifies the bf16 noise of each eval
    assert err.mean() <= 1.5e-2 * amp and err.max() <= 0.12 * amp, (err.mean(), err.max())
    tok = eng.tok_cur().cpu()
    assert torch.equal(tok, torch.sign(pred))                  # binarisation: bit exact (sign(0)=0)
    if branches == 2:
        gerr = (pred - g["sample"][:2]).abs()
        assert gerr.mean() <= 6e-2 and gerr.max() <= 0.4       # vs the reference itself (same bound as the oracle's)


# ----------------------------------------------------------------------------------------------- LLM
def tiny_llm(eng_mod, B=2, branches=1):
    sd = {k: v.to(torch.bfloat16) for k, v in tm.seeded_state(tm.llm_shapes(tm.TINY_LLM), seed=22).items()}
    lw = eng_mod.LlmWeights.from_state_dict(sd, tm.TINY_LLM, DEV)
    eng = eng_mod.Engine(None, None, lw, num_images=B, branches=branches, device=DEV, max_tokens=64, max_kv=256)
    return sd, lw, eng


def test_llm_prefill_and_decode_step(eng_mod, golden_dir):
    """Prefill (hipBLASLt/SDPA path) + the native 64-token decode step vs the oracle and the reference (llm_amp)."""
    from bitdance_amd.llm import prefill_block
    g = load(golden_dir, "llm_amp")
    sd, lw, eng = tiny_llm(eng_mod)
    emb = torch.nn.functional.embedding(g["ids"].long().to(DEV), lw.sd["model.embed_tokens.weight"])
    h1 = prefill_block(eng, lw, emb, 0, 0, causal=True)
    h2 = prefill_block(eng, lw, g["blk"].to(DEV).to(torch.bfloat16), 0, 11, causal=False)
    for got, ref in ((h1, g["h1"]), (h2, g["h2"])):
        e = (got.float().cpu() - ref).abs()
        assert e.max() <= 0.12 and e.mean() <= 1e-2, (e.max(), e.mean())
    eng.set_int("rt.emit_cond", 0)
    eng.reset([75, 75])
    eng.residual()[:128].copy_(g["dec"].reshape(128, 256).to(DEV))
    eng.llm_step()
    torch.cuda.synchronize()
    h3 = eng.hidden().cpu().view(2, 64, 256)
    e = (h3 - g["h3"]).abs()
    assert e.max() <= 0.12 and e.mean() <= 1e-2, (e.max(), e.mean())      # vs the reference itself
    # vs the oracle fed with the same (device-produced) cache: tighter
    w = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    pol = Policy("autocast")
    embc = torch.nn.functional.embedding(g["ids"].long(), w["model.embed_tokens.weight"])
    _, cache = qwen3.model_forward(w, tm.TINY_LLM, embc, None, None, pol)
    ones = torch.ones(2, 1, 64, 75, dtype=torch.bool)
    _, cache = qwen3.model_forward(w, tm.TINY_LLM, g["blk"].to(torch.bfloat16), cache, ones, pol)
    ones = torch.ones(2, 1, 64, 139, dtype=torch.bool)
    o3, _ = qwen3.model_forward(w, tm.TINY_LLM, g["dec"], cache, ones, pol)
    e = (h3 - o3).abs()
    assert e.max() <= 0.1 and e.mean() <= 8e-3, (e.max(), e.mean())


def test_llm_second_step_uses_appended_kv(eng_mod, golden_dir):
    """Two consecutive native steps == the oracle's two decode calls (KV append + position advance)."""
    from bitdance_amd.llm import prefill_block
    g = load(golden_dir, "llm_amp")
    sd, lw, eng = tiny_llm(eng_mod)
    emb = torch.nn.functional.embedding(g["ids"].long().to(DEV), lw.sd["model.embed_tokens.weight"])
    prefill_block(eng, lw, emb, 0, 0, causal=True)
    eng.set_int("rt.emit_cond", 0)
    eng.reset([11, 11])
    x1, x2 = g["dec"], g["blk"] * 1.5
    eng.residual()[:128].copy_(x1.reshape(128, 256).to(DEV)); eng.llm_step()
    eng.residual()[:128].copy_(x2.reshape(128, 256).to(DEV)); eng.llm_step()
    torch.cuda.synchronize()
    got = eng.hidden().cpu().view(2, 64, 256)
    w = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    pol = Policy("autocast")
    _, cache = qwen3.model_forward(w, tm.TINY_LLM, torch.nn.functional.embedding(g["ids"].long(), w["model.embed_tokens.weight"]), None, None, pol)
    _, cache = qwen3.model_forward(w, tm.TINY_LLM, x1, cache, torch.ones(2, 1, 64, 75, dtype=torch.bool), pol)
    ref, _ = qwen3.model_forward(w, tm.TINY_LLM, x2, cache, torch.ones(2, 1, 64, 139, dtype=torch.bool), pol)
    e = (got - ref).abs()
    assert e.max() <= 0.12 and e.mean() <= 1e-2, (e.max(), e.mean())


# ----------------------------------------------------------------------------------------------- whole loop
def tiny_pipeline(head_cfg=None, native_prefill=False):
    from bitdance_amd.t2i_pipeline import BitDanceT2IPipeline
    from bitdance_amd.autoencoder import VQModel
    llm_sd = {k: v.to(torch.bfloat16) for k, v in tm.seeded_state(tm.llm_shapes(tm.TINY_LLM), seed=22).items()}
    ae_shapes = {k: tuple(v.shape) for k, v in VQModel(**tm.TINY_AE).state_dict().items()}
    return BitDanceT2IPipeline.from_components(
        tokenizer=tm.FakeTokenizer(), llm_cfg=tm.TINY_LLM, llm_sd=llm_sd, ae_config=tm.TINY_AE,
        ae_sd=tm.seeded_state(ae_shapes, seed=44, gain=1.4), head_config=dict(head_cfg or tm.TINY_HEAD),
        head_sd=tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11),
        proj_sd=tm.seeded_state(tm.proj_shapes(32, 256), seed=33), device=DEV, native_prefill=native_prefill)


def test_gen_image_teacher_forced_vs_reference(golden_dir):
    """The AR loop with the reference's injected noise, teacher-forced with the reference's tokens
    (a flipped near-zero latent changes all later tokens): pre-sign latents per AR step."""
    from bitdance_amd.llm import prefill_block
    g = load(golden_dir, "gen_amp")
    pipe = tiny_pipeline()
    n, cfg, steps = int(g["n_steps"]), float(g["cfg"]), 4
    noise = g["noise"].view(steps, n + 1, 1, 64, 32)
    cond_ids, uncond_ids = pipe._prompt_ids("a red fox", "<|", [256, 256], True)
    eng = pipe._engine(1, 2, 256, max(len(cond_ids), len(uncond_ids)) + 320)
    eng.set_schedule(n, cfg, steps)
    eng.load_noise(noise.to(DEV))
    pos = pipe.get_2d_embed(16, 16, ps=8)
    eng.pos[:256].copy_(pos)
    embed = pipe.llm_w.sd["model.embed_tokens.weight"]
    hid, kv = [], []
    for br, ids in enumerate([cond_ids, uncond_ids]):
        x = torch.nn.functional.embedding(torch.tensor(ids, device=DEV), embed)[None]
        T0 = x.shape[1] - 64
        prefill_block(eng, pipe.llm_w, x[:, :T0], br, 0, causal=True)
        hid.append(prefill_block(eng, pipe.llm_w, x[:, T0:], br, T0, causal=False))
        kv.append(x.shape[1])
    eng.set_cond((torch.cat(hid)[:, -64:] + pos[None, :64]).reshape(128, -1))
    eng.reset(kv)
    preds = []
    for s in range(steps):
        eng.head_sample()
        preds.append(eng.pred().clone())
        eng.tok_cur().copy_(g["tokens"][:, s * 64:(s + 1) * 64].to(DEV))        # teacher forcing
        if s + 1 < steps:
            eng.projector(); eng.llm_step()
    torch.cuda.synchronize()
    pred, ref = torch.stack(preds).cpu(), g["preds"][:, :1]
    err = (pred - ref).abs()
    assert err.mean() <= 0.2, err.mean()
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.97
    # per-step bound = 1.5x what the CPU oracle itself shows against the reference ([0.21, 0.14, 0.06, 0.06])
    for s_, bound in enumerate((0.32, 0.22, 0.12, 0.12)):
        assert err[s_].mean() <= bound, (s_, err[s_].mean())


@pytest.mark.parametrize("native_prefill", [True, False])
def test_interleaved_edit_plan_vs_reference(golden_dir, native_prefill):
    """MLLModel.forward_inference_block_causal (modeling/mllm.py:695-897) for an image-editing plan [user text, user image,
    model image] on the native engine against the reference's golden (interleaved_amp): encode_image (:899-930: conv encoder ->
    binary tokens in patch order -> native projector -> + 2-D pos embed) reproduces the reference's latents exactly and its
    embeddings to bf16 noise; the generated image's pre-sign latents, teacher-forced with the reference's tokens, stay within the
    text-to-image loop's bound; the plan validation mirrors what the reference can run; graph replay == eager."""
    from bitdance_amd.mllm import MLLModel
    g = load(golden_dir, "interleaved_amp")
    m = MLLModel(tiny_pipeline(native_prefill=native_prefill))
    img = g["image"]
    with torch.autocast("cuda", dtype=torch.bfloat16):                       # the golden ran under (emulated) autocast: bf16 convs
        emb, lat = m.encode_image([img.to(DEV)])
    # sign of the conv encoder's output in patch order: exact on one device (tests/test_host_cpu.py, CPU vs the reference); MIOpen
    # vs the CPU convolution flips the bits whose pre-sign value is ~0
    same = lat.cpu() == g["image_latents"]
    assert set(lat.unique().tolist()) <= {-1.0, 1.0} and same.float().mean().item() >= 0.97, same.float().mean()
    rows = same.all(-1)                                                      # tokens with identical bits: same projector input
    assert rows.float().mean().item() >= 0.3
    assert (emb.float().cpu() - g["image_embeds"])[rows].abs().max().item() <= 0.06
    text = "<|im_start|>user\nmake the fox red<|im_end|>\n<|im_start|>assistant\n"
    plan = [{"type": "text", "from": "user"}, {"type": "image", "from": "user"}, {"type": "image", "from": "model"}]
    n = int(g["n_steps"])
    kw = dict(max_length_vision=256, sample_steps=n, image_size=[256, 256], cfg_scale=float(g["cfg"]),
              noise=g["noise"].view(4, n + 1, 1, 64, 32), return_tokens=True)
    tr = {}
    out = m.forward_inference_block_causal(plan, [text], [img], force_tokens=g["tokens"], trace=tr, **kw)
    assert out["generated_text"] == [] and out["generated_image"][0].shape == (1, 256, 32)
    pred, ref = torch.stack(tr["pred"]).cpu(), g["preds"][:, :1]
    err = (pred - ref).abs()
    assert err.mean() <= 0.25, err.mean()                                      # the bound of the T2I / 16x loops (CFG amplifies x7)
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.95
    if native_prefill:
        m._p.use_graph = False
        t_eager = m.forward_inference_block_causal(plan, [text], [img], **kw)["generated_image"][0].cpu()
        m._p.use_graph = True
        t_graph = m.forward_inference_block_causal(plan, [text], [img], **kw)["generated_image"][0].cpu()
        assert torch.equal(t_eager, t_graph) and set(t_graph.unique().tolist()) <= {-1.0, 0.0, 1.0}
        image = m.forward_inference_block_causal(plan, [text], [img], **dict(kw, return_tokens=False))["generated_image"][0]
        assert image.shape == (1, 3, 256, 256) and torch.isfinite(image).all()
        with pytest.raises(NotImplementedError):                               # the reference's text branch does not run either
            m.forward_inference_block_causal([{"type": "text", "from": "user"}, {"type": "text", "from": "model"}], [text], [])
        with pytest.raises(ValueError):
            m.forward_inference_block_causal(plan, [text], [img], **dict(kw, max_length_vision=64))


def test_gen_image_graph_equals_eager_and_decodes(golden_dir):
    """hipGraph replay == eager launches bit for bit; output image has the reference's shape/range."""
    g = load(golden_dir, "gen_amp")
    pipe = tiny_pipeline()
    noise = g["noise"].view(4, 5, 1, 64, 32)
    kw = dict(cond_prompt="a red fox", uncond_prompt="<|", guidance_scale=4.0, num_sampling_steps=4, max_length=256,
              num_images=1, image_size=[256, 256], noise=noise)
    pipe.use_graph = False
    t_eager = pipe.gen_image(return_tokens=True, **kw).cpu()
    pipe.use_graph = True
    t_graph = pipe.gen_image(return_tokens=True, **kw).cpu()
    t_graph2 = pipe.gen_image(return_tokens=True, **kw).cpu()               # replay of the cached graphs
    assert torch.equal(t_eager, t_graph) and torch.equal(t_graph, t_graph2)
    assert set(t_graph.unique().tolist()) <= {-1.0, 0.0, 1.0}
    assert (t_graph[:, :64] == g["tokens"][:, :64]).float().mean() >= 0.85     # first patch vs the reference
    img = pipe.gen_image(**kw)
    assert img.shape == (1, 3, 256, 256) and torch.isfinite(img).all()
    with pytest.raises(ValueError):
        pipe.generate("x", height=300, width=300)


def test_seed_reproducible_and_rng_order():
    """Same seed -> same tokens; the loop consumes AR_steps*(N+1) normals in the reference's call order."""
    pipe = tiny_pipeline()
    kw = dict(cond_prompt="a", uncond_prompt="b", guidance_scale=3.0, num_sampling_steps=2, max_length=128,
              num_images=2, image_size=[256, 128], return_tokens=True)
    torch.manual_seed(5); a = pipe.gen_image(**kw).cpu()
    torch.manual_seed(5); b = pipe.gen_image(**kw).cpu()
    assert torch.equal(a, b)
    torch.manual_seed(5)
    want = []
    for s in range(2):
        x = torch.randn((2, 64, 32), device=DEV); want.append(x)
        for i in range(2):
            want.append(torch.randn_like(x))
    eng = next(iter(pipe._engines.values()))
    assert torch.equal(eng.noise.view(6, 2, 64, 32).cpu(), torch.stack(want).cpu())


def test_reference_loop_on_seams_equals_fused_graph():
    """The reference's gen_image loop (t2i_pipeline.py:199-268), driven through the drop-in operator seams
    (llm_model.model / vision_head.sample / embed_vision_mlp), produces bit-identical tokens to the fused
    hipGraph path for the same seed: same arithmetic per row, same RNG order."""
    pipe = tiny_pipeline()
    cfg, n, B, P = 3.0, 3, 1, 64
    kw = dict(cond_prompt="a red fox", uncond_prompt="<|", guidance_scale=cfg, num_sampling_steps=n, max_length=128,
              num_images=B, image_size=[256, 128], return_tokens=True)
    torch.manual_seed(11)
    fused = pipe.gen_image(**kw).cpu()
    # --- the reference loop, restated against the seam attributes
    torch.manual_seed(11)
    model = pipe.llm_model.model
    cond_ids, uncond_ids = pipe._prompt_ids("a red fox", "<|", [256, 128], True)
    pos = pipe.get_2d_embed(16, 8, ps=8).unsqueeze(0)
    hid, pkv = [], []
    for ids in (cond_ids, uncond_ids):
        x = model.embed_tokens(torch.tensor(ids, device=DEV))[None]
        o = model(inputs_embeds=x[:, :-P], use_cache=True)
        past = o.past_key_values[0][0].shape[2]
        ones = torch.ones(B, 1, P, P + past, dtype=torch.bool, device=DEV)
        o = model(inputs_embeds=x[:, -P:], past_key_values=o.past_key_values, use_cache=True, attention_mask=ones)
        hid.append(o.last_hidden_state[:, -P:]); pkv.append(o.past_key_values)
    out = []
    for step in range(2):
        sl = slice(step * P, (step + 1) * P)
        hf = torch.cat(hid, dim=0) + pos[:, sl]
        pred = pipe.vision_head.sample(hf, num_sampling_steps=n, cfg=cfg)
        tok = torch.sign(pred)
        out.append(tok[:B])
        emb = pipe.embed_vision_mlp(tok) + pos[:, sl]
        assert emb.dtype == torch.float32
        ones = torch.ones(2 * B, 1, P, P + pkv[0][0][0].shape[2], dtype=torch.bool, device=DEV)
        for br in range(2):
            o = model(inputs_embeds=emb[br * B:(br + 1) * B], past_key_values=pkv[br], use_cache=True,
                      attention_mask=ones[br * B:(br + 1) * B])
            pkv[br] = o.past_key_values
            hid[br] = o.last_hidden_state[:, -P:]
    seam = torch.cat(out, dim=1).cpu()
    assert torch.equal(seam, fused)


def test_16x_model_teacher_forced_and_graph(golden_dir):
    """The 16x models (parallel_num = 16: 16-token patches, the reference's <=32-token attention branch, M = 32 rows):
    teacher-forced pre-sign latents vs the reference (golden gen16_amp), graph == eager, decode shape."""
    from bitdance_amd.llm import prefill_block
    g = load(golden_dir, "gen16_amp")
    pipe = tiny_pipeline(tm.TINY_HEAD16)
    assert pipe.parallel_num == 16 and pipe.ps == 4
    n, cfg, steps, P = int(g["n_steps"]), float(g["cfg"]), 4, 16
    noise = g["noise"].view(steps, n + 1, 1, P, 32)
    cond_ids, uncond_ids = pipe._prompt_ids("a red fox", "<|", [128, 128], True)
    eng = pipe._engine(1, 2, 64, max(len(cond_ids), len(uncond_ids)) + 128)
    eng.set_schedule(n, cfg, steps)
    eng.load_noise(noise.to(DEV))
    pos = pipe.get_2d_embed(8, 8, ps=4)
    eng.pos[:64].copy_(pos)
    embed = pipe.llm_w.sd["model.embed_tokens.weight"]
    hid, kv = [], []
    for br, ids in enumerate([cond_ids, uncond_ids]):
        x = torch.nn.functional.embedding(torch.tensor(ids, device=DEV), embed)[None]
        T0 = x.shape[1] - P
        prefill_block(eng, pipe.llm_w, x[:, :T0], br, 0, causal=True)
        hid.append(prefill_block(eng, pipe.llm_w, x[:, T0:], br, T0, causal=False))
        kv.append(x.shape[1])
    eng.set_cond((torch.cat(hid)[:, -P:] + pos[None, :P]).reshape(2 * P, -1))
    eng.reset(kv)
    preds = []
    for s_ in range(steps):
        eng.head_sample()
        preds.append(eng.pred().clone())
        eng.tok_cur().copy_(g["tokens"][:, s_ * P:(s_ + 1) * P].to(DEV))
        if s_ + 1 < steps:
            eng.projector(); eng.llm_step()
    torch.cuda.synchronize()
    pred, ref = torch.stack(preds).cpu(), g["preds"][:, :1]
    err = (pred - ref).abs()
    assert err.mean() <= 0.3, err.mean()
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.95
    kw = dict(cond_prompt="a red fox", uncond_prompt="<|", guidance_scale=cfg, num_sampling_steps=n, max_length=64,
              num_images=1, image_size=[128, 128], noise=noise)
    pipe.use_graph = False
    te = pipe.gen_image(return_tokens=True, **kw).cpu()
    pipe.use_graph = True
    tg = pipe.gen_image(return_tokens=True, **kw).cpu()
    assert torch.equal(te, tg)
    assert (tg[:, :P] == g["tokens"][:, :P]).float().mean() >= 0.8
    img = pipe.gen_image(**kw)
    assert img.shape == (1, 3, 128, 128) and torch.isfinite(img).all()


def test_gfq_index_math_bit_exact():
    """GFQ sign-quantise -> little-endian index and back (gfq.py:152-160,217-239): integer work, bit exact vs the oracle,
    exhaustive over all 256 byte patterns plus random multi-codebook tokens with exact zeros."""
    from bitdance_amd._lib import check, lib
    from oracle import gfq
    st = torch.cuda.current_stream().cuda_stream
    allbits = gfq.codes_from_indices(np.arange(256), 8)                        # [256, 8] +-1
    g = torch.Generator().manual_seed(3)
    z = torch.randn(1000, 32, generator=g)
    z[::7, ::5] = 0.0                                                          # zeros quantise to -1 (h > 0 test)
    z = torch.cat([torch.from_numpy(np.tile(allbits, (1, 4))).float(), z])
    zd = z.to(DEV).contiguous()
    idx = torch.empty(z.shape[0], 4, dtype=torch.int32, device=DEV)
    check(lib().bd_gfq_indices(zd.data_ptr(), idx.data_ptr(), z.shape[0], 4, 8, st))
    q_ref, idx_ref = gfq.quantize_to_indices(z.numpy(), 4)
    assert np.array_equal(idx.cpu().numpy(), idx_ref.astype(np.int32))
    codes = torch.empty_like(zd)
    check(lib().bd_gfq_codes(idx.data_ptr(), codes.data_ptr(), z.shape[0], 4, 8, st))
    assert np.array_equal(codes.cpu().numpy(), q_ref)                          # round trip == sign quantisation


@pytest.mark.parametrize("M,N,K,S,nw", [(128, 256, 256, 4, 4), (128, 5120, 5120, 6, 4), (128, 15360, 5120, 2, 4),
                                         (128, 15360, 5120, 3, 8), (256, 5120, 7680, 9, 8), (32, 512, 384, 3, 2)])
def test_gemm_in_launch_splitk_reduction(eng_mod, M, N, K, S, nw):
    """Split-K slices reduced inside the launch (last-arriver epilogue, agent-scope release/acquire): the bf16 output
    equals bf16(fp64 reference + bias) up to accumulation-order flips, on every one of several back-to-back launches
    (the tile counters must re-arm themselves), and under concurrent load from a streaming kernel."""
    from bitdance_amd._lib import check, lib
    g = torch.Generator(device=DEV).manual_seed(N + K + S)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = (torch.randn(N, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    xf, rb = frag(eng_mod, x)
    wp = eng_mod.pack_linear([w], DEV)
    scratch = torch.empty(S * rb * 32 * N, device=DEV)
    cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
    ref = (x.to(torch.bfloat16).double() @ w.double().t() + b.double())[:M]
    st = torch.cuda.current_stream().cuda_stream
    big = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    for it in range(6):
        out = torch.full((rb * 32, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        if it % 2:                                           # uneven background load on the memory system
            check(lib().bd_probe_read(big.data_ptr(), big.numel(), 97, sink.data_ptr(), st))
        check(lib().bd_gemm_bf16(xf.data_ptr(), rb, wp.data_ptr(), b.data_ptr(), N, K, S, nw, scratch.data_ptr(),
                                 cnt.data_ptr(), out.data_ptr(), st))
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0                     # counters re-armed
        want = ref.to(torch.bfloat16)                        # correctly rounded result
        d = (out[:M].double() - want.double()).abs()
        assert bool((d <= 2.0 ** -7 * ref.abs().clamp_min(2.0 ** -6)).all()), (it, float(d.max()))   # <= 1 bf16 ulp
        assert float((out[:M] != want).double().mean()) <= 0.02                                  # rounding-boundary flips
        if it == 0:
            first = out.clone()
        assert torch.equal(out[:M], first[:M])               # fixed summation order: bit-identical run to run


@pytest.mark.parametrize("M,N,K,S", [(512, 15360, 5120, 2), (512, 15360, 5120, 1), (256, 512, 256, 1), (512, 2304, 384, 2), (1024, 5120, 7680, 2)])
def test_gemm_256_row_kernel_wide_epilogues(eng_mod, M, N, K, S):
    """The 256-row kernel (gemm_wide_kernel) with its 16 B-per-lane epilogues (round 6: bf16 rows and SwiGLU operand chunks leave through a
    per-wave LDS patch), one K slice and two slices reduced inside the launch: the bf16(+bias) output equals the correctly rounded fp64
    reference up to accumulation-order flips and is bit-identical run to run; the fused SwiGLU equals silu(bf16(h1)) * bf16(h2) at the
    reference's rounding points (flow_head_parallel_x.py:250-251) -- the same bounds as test_gemm_swiglu -- and, at one slice, the
    128-row kernel's bits."""
    from bitdance_amd._lib import check, lib
    tile_was = 1
    check(lib().bd_set_gemm_option(b"tile", 0))                 # the 256-row kernel also where the tiled kernel would take over
    check(lib().bd_set_gemm_option(b"half", 0))                 # ... and where the 256 x 128-tile kernel would
    try:
        g = torch.Generator(device=DEV).manual_seed(M + N + K + S)
        x = torch.randn(M, K, device=DEV, generator=g)
        w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
        b = (torch.randn(N, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
        xf, rb = frag(eng_mod, x)
        st = torch.cuda.current_stream().cuda_stream
        scratch = torch.empty(max(S, 2) * rb * 32 * N, device=DEV)
        cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
        # ---- bf16 + bias
        wp = eng_mod.pack_linear([w], DEV)
        ref = (x.to(torch.bfloat16).double() @ w.double().t() + b.double())[:M]
        want = ref.to(torch.bfloat16)
        first = None
        for it in range(3):
            out = torch.full((rb * 32, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            check(lib().bd_gemm_bf16(xf.data_ptr(), rb, wp.data_ptr(), b.data_ptr(), N, K, S, 8, scratch.data_ptr(), cnt.data_ptr(), out.data_ptr(), st))
            torch.cuda.synchronize()
            assert int(cnt.abs().sum()) == 0
            d = (out[:M].double() - want.double()).abs()
            assert bool((d <= 2.0 ** -7 * ref.abs().clamp_min(2.0 ** -6)).all()), (it, float(d.max()))
            assert float((out[:M] != want).double().mean()) <= 0.02
            first = out.clone() if first is None else first
            assert torch.equal(out, first)
        # ---- fused SwiGLU (N = 2 F packed gate / up)
        F_ = N // 2
        wsp = eng_mod.pack_swiglu(w[:F_], w[F_:], DEV)
        bsp = eng_mod.pack_swiglu_bias(b[:F_], b[F_:], DEV)
        h = (x.to(torch.bfloat16).float() @ w.float().t() + b.float()).to(torch.bfloat16)
        sref = torch.nn.functional.silu(h[:, :F_]) * h[:, F_:]
        acts = []
        for it in range(2):
            act = torch.full((rb * 32 * F_,), float("nan"), dtype=torch.bfloat16, device=DEV)
            if S == 1:
                check(lib().bd_gemm_swiglu(xf.data_ptr(), rb, wsp.data_ptr(), bsp.data_ptr(), N, K, 8, act.data_ptr(), st))
            else:
                check(lib().bd_gemm_swiglu_splitk(xf.data_ptr(), rb, wsp.data_ptr(), bsp.data_ptr(), N, K, S, 8, scratch.data_ptr(), cnt.data_ptr(),
                                                  act.data_ptr(), st))
            torch.cuda.synchronize()
            a = act.view(F_ // 16, rb, 2, 32, 8).permute(1, 3, 0, 2, 4).reshape(rb * 32, F_)[:M]
            d = (a.float() - sref.float()).abs()
            assert torch.isfinite(a).all() and (d > 0).float().mean() <= 0.02 and d.max() <= 0.07, ((d > 0).float().mean(), d.max())
            acts.append(act)
        assert torch.equal(acts[0], acts[1])
        if S == 1 and M % 128 == 0:                              # one slice: every K sum in the 128-row kernel's order -> its bits
            act4 = torch.zeros(rb * 32 * F_, dtype=torch.bfloat16, device=DEV)
            for r0 in range(0, M, 128):
                xs, rbs = frag(eng_mod, x[r0:r0 + 128])
                a4 = torch.zeros(rbs * 32 * F_, dtype=torch.bfloat16, device=DEV)
                check(lib().bd_gemm_swiglu(xs.data_ptr(), rbs, wsp.data_ptr(), bsp.data_ptr(), N, K, 4, a4.data_ptr(), st))
                torch.cuda.synchronize()
                got = acts[0].view(F_ // 16, rb, 2, 32, 8).permute(1, 3, 0, 2, 4).reshape(rb * 32, F_)[r0:r0 + 128]
                assert torch.equal(got, a4.view(F_ // 16, rbs, 2, 32, 8).permute(1, 3, 0, 2, 4).reshape(rbs * 32, F_)[:128])
            del act4
    finally:
        check(lib().bd_set_gemm_option(b"tile", tile_was))
        check(lib().bd_set_gemm_option(b"half", 1))


HALF_FORM_DEFAULT = 1


@pytest.mark.parametrize("M,N,K", [(512, 15360, 5120), (512, 256, 128), (768, 384, 192), (500, 1152, 704), (512, 5120, 7680)])
def test_gemm_512_row_kernel_half_tiles(eng_mod, M, N, K):
    """The 512-row kernel (bd_gemm_half.hip: 256 x 128 tiles, ONE K slice, the two wave groups of a workgroup take the even / odd 32-deep
    K sub-stages and add their accumulators through LDS).  bf16(+bias): equals the correctly rounded fp64 reference up to accumulation-order
    flips, bit-identical run to run, and bit-identical to the 256-row kernel's TWO-slab sum order where that is the same sum -- it is not
    (even / odd sub-stages against first / second half of K), so the comparison with that kernel is by the rounding bound only.  Fused
    SwiGLU: silu(bf16(h1)) * bf16(h2) at the reference's rounding points (flow_head_parallel_x.py:250-251), test_gemm_swiglu's bounds."""
    from bitdance_amd._lib import check, lib
    check(lib().bd_set_gemm_option(b"half", 2))                 # any N (the default routes 8192 <= N <= 16384 only)
    try:
        g = torch.Generator(device=DEV).manual_seed(M + N + K)
        x = torch.randn(M, K, device=DEV, generator=g)
        w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
        b = (torch.randn(N, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
        xf, rb = frag(eng_mod, x)
        assert rb % 8 == 0 and rb >= 16
        st = torch.cuda.current_stream().cuda_stream
        scratch = torch.empty(2 * rb * 32 * N, device=DEV)
        cnt = torch.zeros(16384, dtype=torch.int32, device=DEV)
        wp = eng_mod.pack_linear([w], DEV)
        ref = (x.to(torch.bfloat16).double() @ w.double().t() + b.double())[:M]
        want = ref.to(torch.bfloat16)
        first = None
        for it in range(4):                                     # both operand paths of the K loop ("half.form" 0 / 1): the same sums, bit for bit
            check(lib().bd_set_gemm_option(b"half.form", it & 1))
            out = torch.full((rb * 32, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            check(lib().bd_gemm_bf16(xf.data_ptr(), rb, wp.data_ptr(), b.data_ptr(), N, K, 1, 8, scratch.data_ptr(), cnt.data_ptr(), out.data_ptr(), st))
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()            # every element of the padded tile rows written
            d = (out[:M].double() - want.double()).abs()
            assert bool((d <= 2.0 ** -7 * ref.abs().clamp_min(2.0 ** -6)).all()), (it, float(d.max()))
            assert float((out[:M] != want).double().mean()) <= 0.02
            first = out.clone() if first is None else first
            assert torch.equal(out, first)
        # the 256-row kernel on the same operands (one slice): same values up to the order of the fp32 sums
        check(lib().bd_set_gemm_option(b"half", 0))
        check(lib().bd_set_gemm_option(b"tile", 0))
        out_w = torch.full((rb * 32, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        if N % 256 == 0:
            check(lib().bd_gemm_bf16(xf.data_ptr(), rb, wp.data_ptr(), b.data_ptr(), N, K, 1, 8, scratch.data_ptr(), cnt.data_ptr(), out_w.data_ptr(), st))
            torch.cuda.synchronize()
            assert float((out_w[:M] != first[:M]).double().mean()) <= 0.02
        check(lib().bd_set_gemm_option(b"tile", 1))
        check(lib().bd_set_gemm_option(b"half", 2))
        # ---- K slices as fp32 slabs (the N = 5120 Linears at 512 rows: 3 slabs): the slab sum equals the fp64 product of the bf16 operands
        for S in (2, 3):
            if (K // 64) // S < 2:
                continue
            slabs = torch.full((S, rb * 32, N), float("nan"), device=DEV)
            check(lib().bd_gemm_partial(xf.data_ptr(), rb, wp.data_ptr(), N, K, S, 8, slabs.data_ptr(), st))
            torch.cuda.synchronize()
            assert torch.isfinite(slabs).all()
            tot = slabs.double().sum(0)[:M]
            refp = x.to(torch.bfloat16).double() @ w.double().t()
            assert float((tot - refp).abs().max()) <= 1e-4 * max(1.0, float(refp.abs().max())), (S, float((tot - refp).abs().max()))
            slabs2 = torch.empty_like(slabs)
            check(lib().bd_gemm_partial(xf.data_ptr(), rb, wp.data_ptr(), N, K, S, 8, slabs2.data_ptr(), st))
            torch.cuda.synchronize()
            assert torch.equal(slabs, slabs2)
        # ---- fused SwiGLU (N = 2 F packed gate / up)
        F_ = N // 2
        wsp = eng_mod.pack_swiglu(w[:F_], w[F_:], DEV)
        bsp = eng_mod.pack_swiglu_bias(b[:F_], b[F_:], DEV)
        h = (x.to(torch.bfloat16).float() @ w.float().t() + b.float()).to(torch.bfloat16)
        sref = torch.nn.functional.silu(h[:, :F_]) * h[:, F_:]
        acts = []
        for it in range(2):
            act = torch.full((rb * 32 * F_,), float("nan"), dtype=torch.bfloat16, device=DEV)
            check(lib().bd_gemm_swiglu(xf.data_ptr(), rb, wsp.data_ptr(), bsp.data_ptr(), N, K, 8, act.data_ptr(), st))
            torch.cuda.synchronize()
            assert torch.isfinite(act.float()).all()
            a = act.view(F_ // 16, rb, 2, 32, 8).permute(1, 3, 0, 2, 4).reshape(rb * 32, F_)[:M]
            d = (a.float() - sref.float()).abs()
            assert (d > 0).float().mean() <= 0.02 and d.max() <= 0.07, ((d > 0).float().mean(), d.max())
            acts.append(act)
        assert torch.equal(acts[0], acts[1])
    finally:
        check(lib().bd_set_gemm_option(b"half", 1))
        check(lib().bd_set_gemm_option(b"half.form", HALF_FORM_DEFAULT))
        check(lib().bd_set_gemm_option(b"tile", 1))


# ----------------------------------------------------------------------------------------------- imagenet (I1-I3)
def test_imagenet_head_eval_dh64_vs_oracle(eng_mod):
    """diff_head_parallel.TransEncoder (head_dim 64, explicit-softmax attention, no final sigmoid) on the HIP head:
    one evaluation against the oracle's autocast policy.  Tolerance: bf16 noise of a 4-block head on outputs of O(1)."""
    c = tm.TINY_IN
    sd = tm.seeded_state(tm.imagenet_shapes(c), seed=29)
    hsd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    hw = eng_mod.HeadWeights.from_state_dict(hsd, DEV, head_dim=64, final_sigmoid=False)
    B, P, C, D = 2, 16, c["latent_dim"], c["dim"]
    eng = eng_mod.Engine(hw, None, None, num_images=B, branches=2, device=DEV, max_tokens=P, parallel_num=P)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2 * B, P, D, generator=g)
    noise = torch.randn(1, 4, B, P, C, generator=g)
    eng.set_schedule(3, 2.0, 1)
    eng.load_noise(noise.to(DEV))
    eng.reset([0] * (2 * B))
    eng.set_int("rt.dump_xhat", 1)
    eng.set_cond(z.to(DEV))
    x0 = noise[0, 0]
    eng.view("head.xt", torch.float32, (B * P, C)).copy_(x0.reshape(B * P, C))     # eval 0: latent = first draw
    eng.head_cond()
    eng.head_eval(0)
    torch.cuda.synchronize()
    xhat = eng.view("head.xhat", torch.float32, (eng.Mpad, C))[: 2 * B * P].cpu().view(2 * B, P, C)
    ref = diff_head.net_forward(hsd, torch.cat([x0, x0]), torch.zeros(2 * B), z, Policy("autocast"),
                                final_sigmoid=False, head_dim=64).float()
    d = (xhat - ref).abs()
    assert d.max() <= 0.08 * ref.abs().max() + 0.02 and d.mean() <= 0.01 * ref.abs().mean() + 2e-3, (d.max(), d.mean())


def test_ln_mod_wave_per_row_matches_workgroup_per_row(eng_mod):
    """From 1024 rows up the head's ``ln_mod`` runs one WAVE per row, eight rows per workgroup (bd_rows.hip ln_mod_rows_kernel; the
    ImageNet batch) instead of one workgroup per row: same arithmetic per element, the LayerNorm sums in a different order.  One
    evaluation of the tiny ImageNet head (D = 256: the one-pass form) at 1024 rows with ``tune.ln_rows`` 1 / 0: the predictions agree
    to bf16 noise of the statistics (a last-bit difference in mean / rstd moves an output by <= 1 bf16 ulp before the next
    Linear), and the first sequences match the oracle within the bound of the small-batch test."""
    c = tm.TINY_IN
    sd = tm.seeded_state(tm.imagenet_shapes(c), seed=29)
    hsd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    hw = eng_mod.HeadWeights.from_state_dict(hsd, DEV, head_dim=64, final_sigmoid=False)
    B, P, C, D = 32, 16, c["latent_dim"], c["dim"]                          # 2 * 32 * 16 = 1024 rows
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2 * B, P, D, generator=g)
    noise = torch.randn(1, 4, B, P, C, generator=g)
    x0 = noise[0, 0]
    outs = []
    for rows_mode in (1, 0):
        eng = eng_mod.Engine(hw, None, None, num_images=B, branches=2, device=DEV, max_tokens=P, parallel_num=P,
                             tune={"ln_rows": rows_mode})
        eng.set_schedule(3, 2.0, 1)
        eng.load_noise(noise.to(DEV))
        eng.reset([0] * (2 * B))
        eng.set_int("rt.dump_xhat", 1)
        eng.set_cond(z.to(DEV))
        eng.view("head.xt", torch.float32, (B * P, C)).copy_(x0.reshape(B * P, C))
        eng.head_cond()
        eng.head_eval(0)
        torch.cuda.synchronize()
        outs.append(eng.view("head.xhat", torch.float32, (eng.Mpad, C))[: 2 * B * P].cpu().view(2 * B, P, C).clone())
    d = (outs[0] - outs[1]).abs()
    assert d.max() <= 0.03 * outs[1].abs().max() and d.mean() <= 2e-3 * outs[1].abs().mean() + 1e-4, (d.max(), d.mean())
    sel = [0, 1, B, B + 1]                                                  # two cond + the matching uncond sequences
    ref = diff_head.net_forward(hsd, torch.cat([x0[:2], x0[:2]]), torch.zeros(4), z[sel], Policy("autocast"),
                                final_sigmoid=False, head_dim=64).float()
    e = (outs[0][sel] - ref).abs()
    assert e.max() <= 0.08 * ref.abs().max() + 0.02 and e.mean() <= 0.01 * ref.abs().mean() + 2e-3, (e.max(), e.mean())


def test_imagenet_sample_teacher_forced_vs_reference(golden_dir):
    """BitDance.sample (model_parallel.py:371-419) with the reference's noise and tokens fed back: per-AR-step pre-sign
    latents against the reference's own (golden imagenet_amp).  Bound per step: the oracle-vs-reference bf16 noise
    (tests/test_oracle_golden.py) x 1.5, scaled by the CFG amplification (2 cfg_i - 1) of the linear ramp."""
    from bitdance_amd.imagenet import BitDance
    g = load(golden_dir, "imagenet_amp")
    c = tm.TINY_IN
    m = BitDance(tm.seeded_state(tm.imagenet_shapes(c), seed=29), device=DEV, **c)
    N, P = int(g["n_steps"]), c["parallel_num"]
    steps = (c["resolution"] // 16) ** 2 // P
    noise = [g["noise0"]] + [g["noise1"][k * (N + 1):(k + 1) * (N + 1)] for k in range(steps - 1)]
    ref_tok = torch.sign(g["preds"])
    lat, tokens, preds = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]), noise=noise, force_tokens=ref_tok,
                                  return_tokens=True)
    preds, lat = preds.cpu(), lat.cpu()
    assert torch.equal(lat, g["latent"])                      # teacher-forced tokens, un-patchified: index work exact
    for i in range(steps):
        sl = slice(i * P, (i + 1) * P)
        cfg_i = 1.0 + (float(g["cfg"]) - 1.0) * i / steps
        ref = g["preds"][:, sl]
        d = (preds[:, sl] - ref).abs()
        assert d.mean().item() <= 0.07 * max(1.0, 2 * cfg_i - 1) * ref.abs().mean().item(), (i, d.mean())
        firm = ref.abs() > 0.5
        assert (torch.sign(preds[:, sl])[firm] == torch.sign(ref)[firm]).float().mean().item() >= 0.96, i
    # the HIP transformer decode steps against the same steps as torch ops (the reference's own arithmetic): identical
    # rounding points, so the two differ by accumulation order only
    m.native_transformer = False
    _, _, preds_t = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]), noise=noise, force_tokens=ref_tok, return_tokens=True)
    m.native_transformer = True
    dt = (preds - preds_t.cpu()).abs()
    assert dt.mean().item() <= 0.09 * g["preds"].abs().mean().item(), dt.mean()      # bf16 noise x CFG amplification, as above
    # free-running: same seed -> same latent; RNG consumption = AR_steps * (1 + N) draws in the reference's order
    torch.manual_seed(7)
    a = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]))
    torch.manual_seed(7)
    b = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]))
    assert torch.equal(a, b) and a.shape == (2, c["latent_dim"], 8, 8) and set(a.unique().tolist()) <= {-1.0, 0.0, 1.0}


@pytest.mark.parametrize("name", ["16x", "4x", "1x"])
def test_imagenet_first_call_native_vs_torch(name):
    """The FIRST forward_model call (class + query tokens under attn_mask[:T0, :T0], model_parallel.py:386-388; 1x: model.py:372-377)
    on the step kernels -- causal blocks of P class tokens, then the last P tokens as one bidirectional block, fp32 residual stream --
    against the same call as torch ops under the device's autocast (the reference's own arithmetic): norm(x) of the last P tokens
    and every layer's K / V for all T0 positions.  Identical rounding points, so what remains is accumulation order."""
    import torch.nn.functional as F
    from bitdance_amd.imagenet import BitDance
    c = dict({"16x": tm.TINY_IN, "4x": tm.TINY_IN_4X, "1x": tm.TINY_IN_1X}[name])
    m = BitDance(tm.seeded_state(tm.imagenet_shapes(c), seed=29 if name == "16x" else 31), device=DEV, **c)
    P, n_cls = m.P, m.cls_token_num
    ids = torch.tensor([3, 7, c["num_classes"], c["num_classes"]], device=DEV)
    bsz, T0 = ids.shape[0], n_cls + P - 1
    eng = m._tr_engine(bsz)
    x_nat = m._first_step_native(eng, ids).float().cpu()
    nat = [(k.float().cpu(), v.float().cpu()) for k, v in m._cache_views(eng, T0)]
    hd = m.dim // m.n_head
    caches = [(torch.zeros(bsz, m.n_head, m.total_tokens, hd, device=DEV), torch.zeros(bsz, m.n_head, m.total_tokens, hd, device=DEV))
              for _ in range(m.n_layer)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        cc = F.embedding(ids, m.w_["cls_embedding.weight"]).view(bsz, n_cls, -1)
        x = torch.cat([cc, m.w_["query_token"].repeat(bsz, 1, 1)], dim=1) if P > 1 else cc
        x_ref = m._forward_model(x, m.attn_mask[:, :, :T0, :T0], 0, T0, caches)[:, -P:, :]
    assert x_ref.dtype == torch.float32                          # the first call's stream is fp32 (fp32 class embedding)
    x_ref = x_ref.cpu()
    d = (x_nat - x_ref).abs()
    scale = x_ref.abs().mean().item()
    print(f"[imagenet first call {name}] hidden max {d.max().item():.4f} mean {d.mean().item():.5f} (ref mean |x| {scale:.3f})")
    assert torch.isfinite(x_nat).all() and d.mean().item() <= 0.01 * scale + 1e-3 and d.max().item() <= 0.12 * max(1.0, x_ref.abs().max().item())
    for l, ((kn, vn), (kr, vr)) in enumerate(zip(nat, caches)):
        dk = (kn - kr[:, :, :T0].cpu()).abs()
        dv = (vn - vr[:, :, :T0].cpu()).abs()
        assert dk.mean().item() <= 0.01 * kr.abs().mean().item() + 1e-3 and dv.mean().item() <= 0.01 * vr.abs().mean().item() + 1e-3, (l, dk.mean(), dv.mean())


@pytest.mark.parametrize("name", ["1x", "4x"])
def test_imagenet_variants_teacher_forced_vs_reference(golden_dir, name):
    """The other released ImageNet variants on the native engine: BitDance-*-1x (imagenet_gen/src/model.py:
    one token per AR step, causal transformer, MLP head diff_head.py:228-253 -> head.variant 1, P = 1 decode steps) and the 4x
    parallel variant (model_parallel.py, parallel_num 4: 4-token head attention and decode blocks).  Reference noise and tokens
    fed back; per-AR-step pre-sign latents against the reference's own (goldens imagenet1x_amp / imagenet4x_amp), bound = the
    oracle-vs-reference bf16 noise x 1.5 scaled by the CFG amplification; HIP transformer vs the same steps as torch ops."""
    from bitdance_amd.imagenet import BitDance
    g = load(golden_dir, f"imagenet{name}_amp")
    c = dict(tm.TINY_IN_1X if name == "1x" else tm.TINY_IN_4X)
    m = BitDance(tm.seeded_state(tm.imagenet_shapes(c), seed=31), device=DEV, **c)
    assert m.head_w.mlp == (name == "1x")
    N, P = int(g["n_steps"]), c["parallel_num"]
    steps = (c["resolution"] // 16) ** 2 // P
    noise = [g["noise0"]] + [g["noise1"][k * (N + 1):(k + 1) * (N + 1)] for k in range(steps - 1)]
    ref_tok = torch.sign(g["preds"])
    lat, tokens, preds = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]), noise=noise, force_tokens=ref_tok,
                                  return_tokens=True)
    preds, lat = preds.cpu(), lat.cpu()
    assert torch.equal(lat, g["latent"])
    for i in range(steps):
        sl = slice(i * P, (i + 1) * P)
        cfg_i = 1.0 + (float(g["cfg"]) - 1.0) * i / steps
        ref = g["preds"][:, sl]
        d = (preds[:, sl] - ref).abs()
        assert d.mean().item() <= 0.075 * max(1.0, 2 * cfg_i - 1) * ref.abs().mean().item() + 3e-3, (i, d.mean())
    firm = g["preds"].abs() > 0.5
    assert (torch.sign(preds)[firm] == ref_tok[firm]).float().mean().item() >= 0.96
    m.native_transformer = False
    _, _, preds_t = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]), noise=noise, force_tokens=ref_tok, return_tokens=True)
    m.native_transformer = True
    dt = (preds - preds_t.cpu()).abs()
    assert dt.mean().item() <= 0.09 * g["preds"].abs().mean().item(), dt.mean()
    torch.manual_seed(7)
    a = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]))
    torch.manual_seed(7)
    b = m.sample(g["ids"], N, cfg_scale=float(g["cfg"]))
    assert torch.equal(a, b) and a.shape == (2, c["latent_dim"], 4, 4) and set(a.unique().tolist()) <= {-1.0, 0.0, 1.0}


def test_imagenet_mlp_head_eval_vs_oracle(eng_mod):
    """One evaluation of the MLP head (imagenet_gen/src/diff_head.py:228-253: input_proj, adaLN blocks of 3 chunks, ResBlocks
    = LayerNorm-modulate -> SwiGLU -> gated residual, final layer, no squash) on the HIP engine, rows = sequences (P = 1),
    against the oracle's autocast policy."""
    c = tm.TINY_IN_1X
    sd = tm.seeded_state(tm.imagenet_shapes(c), seed=31)
    hsd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    hw = eng_mod.HeadWeights.from_state_dict(hsd, DEV, head_dim=64, final_sigmoid=False)
    assert hw.mlp
    B, C, D = 24, c["latent_dim"], c["dim"]
    eng = eng_mod.Engine(hw, None, None, num_images=B, branches=2, device=DEV, max_tokens=1, parallel_num=1)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2 * B, 1, D, generator=g)
    noise = torch.randn(1, 4, B, 1, C, generator=g)
    eng.set_schedule(3, 2.0, 1)
    eng.load_noise(noise.to(DEV))
    eng.reset([0] * 16)
    eng.set_int("rt.dump_xhat", 1)
    eng.set_cond(z.to(DEV))
    x0 = noise[0, 0]
    eng.view("head.xt", torch.float32, (B, C)).copy_(x0.reshape(B, C))
    eng.head_cond()
    eng.head_eval(0)
    torch.cuda.synchronize()
    xhat = eng.view("head.xhat", torch.float32, (eng.Mpad, C))[: 2 * B].cpu()
    ref = diff_head.mlp_net_forward(hsd, torch.cat([x0, x0]).view(2 * B, C), torch.zeros(2 * B), z.view(2 * B, D),
                                    Policy("autocast")).float()
    d = (xhat - ref).abs()
    assert d.max() <= 0.08 * ref.abs().max() + 0.02 and d.mean() <= 0.01 * ref.abs().mean() + 2e-3, (d.max(), d.mean())
    # the whole sampler on the MLP head: CFG-mixed sample vs the oracle's, same injected noise
    eng.head_sample()
    torch.cuda.synchronize()
    got = eng.pred().cpu().view(B, C)
    from oracle import sampler
    fwd = lambda xx, tt, cc: diff_head.mlp_net_forward(hsd, xx, tt, cc, Policy("autocast"))
    want = sampler.euler_maruyama(C, fwd, z.view(2 * B, D), 2.0, 3, list(noise[0].reshape(4, B, C)))[:B]
    ds = (got - want).abs()
    assert ds.mean().item() <= 0.05 * want.abs().mean().item() + 5e-3, (ds.mean(), want.abs().mean())


def test_imagenet_transformer_decode_step_vs_oracle():
    """One proj_in + forward_model block (model_parallel.py:342-350, layers_parallel.py:120-168,229-241) on the HIP engine
    against the oracle's autocast policy, K/V cache pre-filled by the oracle's first step.  Output = norm(x): bf16 values
    of O(1); tolerance = a few bf16 ulps of accumulated GEMM-order noise over 2 layers."""
    from bitdance_amd.imagenet import BitDance
    from oracle import imagenet as oim
    c = dict(tm.TINY_IN)
    sd = tm.seeded_state(tm.imagenet_shapes(c), seed=29)
    m = BitDance(sd, device=DEV, **c)
    pol = Policy("autocast")
    bsz, P, ncls = 4, c["parallel_num"], c["cls_token_num"]
    hw = c["resolution"] // 16
    total = hw * hw + ncls
    hd = c["dim"] // c["n_head"]
    caches = [(torch.zeros(bsz, c["n_head"], total, hd), torch.zeros(bsz, c["n_head"], total, hd)) for _ in range(c["n_layer"])]
    fc, mask = oim.rope_table(c), oim.block_causal_mask(hw * hw + ncls - 1, ncls - 1, P)[None, None]
    ids = torch.tensor([1, 4, 10, 10])
    x0 = torch.cat([torch.nn.functional.embedding(ids, sd["cls_embedding.weight"]).view(bsz, ncls, -1),
                    sd["query_token"].repeat(bsz, 1, 1)], dim=1)
    T0 = ncls + P - 1
    oim.forward_model(sd, c, x0, mask[:, :, :T0, :T0], fc[:T0], caches, 0, T0, pol)
    g = torch.Generator().manual_seed(3)
    tok = torch.sign(torch.randn(bsz, P, c["latent_dim"], generator=g))
    ref = oim.forward_model(sd, c, oim.proj_in(sd, tok, pol), mask[:, :, T0:T0 + P, :T0 + P], fc[T0:T0 + P],
                            [(k.clone(), v.clone()) for k, v in caches], T0, T0 + P, pol).float()
    eng = m._tr_engine(bsz)
    m._load_cache(eng, [(k.to(DEV), v.to(DEV)) for k, v in caches], T0)
    got = m._decode_step(eng, tok.to(DEV)).float().cpu()
    d = (got - ref).abs()
    assert d.max().item() <= 0.06 * ref.abs().max().item() + 0.02 and d.mean().item() <= 0.01 * ref.abs().mean().item() + 1e-3, \
        (d.max(), d.mean(), ref.abs().mean())


def test_imagenet_bitdance_b_dims_run():
    """BitDance-B-16x at its real dimensions (model_parallel.py:456-465: dim 768, 24 layers, 12 heads of 64, FFN 2048;
    head 6 blocks / 2 adaLN at 768; 256 px -> 16 AR steps; 64 class tokens), random weights, 2 sampling steps: shape /
    divisibility coverage of every kernel on the non-power-of-two widths, HIP transformer == torch transformer within
    bf16 noise on the first decode steps, deterministic."""
    from bitdance_amd.imagenet import BitDance
    c = dict(dim=768, n_layer=24, n_head=12, diff_layers=6, diff_dim=768, diff_adanln_layers=2, latent_dim=32, down_size=16,
             patch_size=1, resolution=256, cls_token_num=64, num_classes=1000, parallel_num=16, time_shift=1.0)
    sd = {k: v for k, v in tm.seeded_state(tm.imagenet_shapes(c), seed=5).items()}
    m = BitDance(sd, device=DEV, **c)
    ids = torch.tensor([1, 207, 980, 33])
    torch.manual_seed(11)
    lat, tok, pred = m.sample(ids, 2, cfg_scale=4.0, return_tokens=True)
    torch.manual_seed(11)
    lat2, tok2, pred2 = m.sample(ids, 2, cfg_scale=4.0, return_tokens=True)
    assert lat.shape == (4, 32, 16, 16) and torch.isfinite(pred).all() and torch.equal(pred, pred2)
    m.native_transformer = False
    torch.manual_seed(11)
    _, _, pred_t = m.sample(ids, 2, cfg_scale=4.0, force_tokens=tok, return_tokens=True)
    m.native_transformer = True
    torch.manual_seed(11)
    _, _, pred_n = m.sample(ids, 2, cfg_scale=4.0, force_tokens=tok, return_tokens=True)
    P = 16
    for i in (0, 1, 2):                                        # (step 0: the native first call vs the torch one, since round 4)
        d = (pred_n[:, i * P:(i + 1) * P] - pred_t[:, i * P:(i + 1) * P]).abs().mean().item()
        ref = pred_t[:, i * P:(i + 1) * P].abs().mean().item()
        assert d <= 0.08 * ref + 1e-6, (i, d, ref)
    m.native_first_step = False                                # ... with the first call on torch ops in both: step 0 identical
    torch.manual_seed(11)
    _, _, pred_f = m.sample(ids, 2, cfg_scale=4.0, force_tokens=tok, return_tokens=True)
    m.native_first_step = True
    assert torch.equal(pred_f[:, :P], pred_t[:, :P])


@pytest.mark.parametrize("variant", ["b1x", "h1x", "b4x"])
def test_imagenet_released_variants_real_dims_run(variant):
    """The other released checkpoints at their REAL dimensions (imagenet_gen/src/model.py:394-430: B-1x width 768 / 24 layers,
    H-1x width 1280 / 40 layers / 12-block MLP head with 3 adaLN projections; model_parallel.py B-4x), random weights, 2 sampling
    steps, 4 classes with CFG: shape / divisibility coverage of every kernel on those widths at P = 1 and P = 4; HIP transformer
    == torch transformer within bf16 noise on the first decode steps; deterministic."""
    from bitdance_amd import synthetic as syn
    from bitdance_amd.imagenet import BitDance
    c = dict(syn.IMAGENET_MODELS[variant])
    m = BitDance(syn.random_imagenet_state(c, DEV), device=DEV, **c)
    P = c["parallel_num"]
    assert m.head_w.mlp == (P == 1)
    ids = torch.tensor([1, 207, 980, 33])
    torch.manual_seed(11)
    lat, tok, pred = m.sample(ids, 2, cfg_scale=4.0, return_tokens=True)
    torch.manual_seed(11)
    lat2, tok2, pred2 = m.sample(ids, 2, cfg_scale=4.0, return_tokens=True)
    assert lat.shape == (4, 32, 16, 16) and torch.isfinite(pred).all() and torch.equal(pred, pred2)
    assert set(lat.unique().tolist()) <= {-1.0, 0.0, 1.0}
    if variant == "h1x":                                       # 40 layers as torch ops: covered at B width
        return
    # decode steps of the HIP transformer against the same steps as torch ops (the reference's arithmetic), compared where
    # they differ -- norm(x), before the head's sampler amplifies bf16 noise (measured in round 3: rel. error 0.010-0.017 at every variant)
    import torch.nn.functional as F
    n_cls, bsz, hd = m.cls_token_num, 8, m.dim // m.n_head
    cls_ids = torch.arange(bsz, device=DEV)
    caches = [(torch.zeros(bsz, m.n_head, m.total_tokens, hd, device=DEV), torch.zeros(bsz, m.n_head, m.total_tokens, hd, device=DEV))
              for _ in range(m.n_layer)]
    T0 = n_cls + P - 1
    with torch.autocast("cuda", dtype=torch.bfloat16):
        x = F.embedding(cls_ids, m.w_["cls_embedding.weight"]).view(bsz, n_cls, -1)
        if P > 1:
            x = torch.cat([x, m.w_["query_token"].repeat(bsz, 1, 1)], dim=1)
        m._forward_model(x, m.attn_mask[:, :, :T0, :T0], 0, T0, caches)
    eng = m._tr_engine(bsz)
    m._load_cache(eng, caches, T0)
    g = torch.Generator(device=DEV).manual_seed(3)
    for i in range(1, 4):
        tk = torch.sign(torch.randn(bsz, P, c["latent_dim"], device=DEV, generator=g))
        s0 = P * (i - 1) + n_cls + P - 1
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref = m._forward_model(m._proj_in(tk), m.attn_mask[:, :, s0:s0 + P, :s0 + P], s0, s0 + P, caches).float()
        d = (m._decode_step(eng, tk).float() - ref).abs()
        assert d.mean().item() <= 0.03 * ref.abs().mean().item() and d.max().item() <= 0.25, (variant, i, d.mean(), d.max())


@pytest.mark.parametrize("name,schedule", [("const", "constant"), ("nocfg", "linear")])
def test_imagenet_other_cfg_branches_vs_reference(golden_dir, name, schedule):
    """Constant CFG (mixed from the first step, two-branch head context throughout) and cfg_scale <= 1 (single branch)
    against the reference's golden latents, teacher-forced; same per-step bound as the linear-ramp test."""
    from bitdance_amd.imagenet import BitDance
    g = load(golden_dir, f"imagenet_{name}_amp")
    c = tm.TINY_IN
    m = BitDance(tm.seeded_state(tm.imagenet_shapes(c), seed=29), device=DEV, **c)
    N, P, cfg = int(g["n_steps"]), c["parallel_num"], float(g["cfg"])
    steps = (c["resolution"] // 16) ** 2 // P
    noise = [g["noise"][k * (N + 1):(k + 1) * (N + 1)] for k in range(steps)]
    ref_tok = torch.sign(g["preds"])
    lat, tokens, preds = m.sample(g["ids"], N, cfg_scale=cfg, cfg_schedule=schedule, noise=noise, force_tokens=ref_tok,
                                  return_tokens=True)
    preds = preds.cpu()
    assert torch.equal(lat.cpu(), g["latent"])
    amp = max(1.0, 2 * cfg - 1) if cfg > 1.0 else 1.0
    for i in range(steps):
        sl = slice(i * P, (i + 1) * P)
        ref = g["preds"][:, sl]
        d = (preds[:, sl] - ref).abs()
        assert d.mean().item() <= 0.07 * amp * ref.abs().mean().item(), (i, d.mean())
        firm = ref.abs() > 0.5
        assert (torch.sign(preds[:, sl])[firm] == torch.sign(ref)[firm]).float().mean().item() >= 0.96, i


@pytest.mark.parametrize("name,n_img", [("genb2", 2), ("gennocfg", 1)])
def test_gen_image_batch2_and_nocfg_vs_reference(golden_dir, name, n_img):
    """num_images = 2 (M = 256 rows: the 256-row GEMM passes inside the loop) and guidance_scale <= 1 (single branch)
    against the reference's golden latents, teacher-forced with its tokens; per-step bounds as in the 1-image test
    (no CFG amplification for the single-branch case)."""
    from bitdance_amd.llm import prefill_block
    g = load(golden_dir, name + "_amp")
    pipe = tiny_pipeline()
    n, cfg, steps, P = int(g["n_steps"]), float(g["cfg"]), 2, 64
    branches = 2 if cfg > 1.0 else 1
    noise = g["noise"].view(steps, n + 1, n_img, P, 32)
    cond_ids, uncond_ids = pipe._prompt_ids("a red fox", "<|", [256, 128], branches == 2)
    eng = pipe._engine(n_img, branches, 128, len(cond_ids) + 192)
    eng.set_schedule(n, cfg, steps)
    eng.load_noise(noise.to(DEV))
    pos = pipe.get_2d_embed(16, 8, ps=8)
    eng.pos[:128].copy_(pos)
    embed = pipe.llm_w.sd["model.embed_tokens.weight"]
    hid, kv = [], []
    for br, ids in enumerate([cond_ids, uncond_ids][:branches]):
        x = torch.nn.functional.embedding(torch.tensor(ids, device=DEV), embed)[None].repeat(n_img, 1, 1)
        T0 = x.shape[1] - P
        prefill_block(eng, pipe.llm_w, x[:, :T0], br * n_img, 0, causal=True)
        hid.append(prefill_block(eng, pipe.llm_w, x[:, T0:], br * n_img, T0, causal=False))
        kv += [x.shape[1]] * n_img
    eng.set_cond((torch.cat(hid)[:, -P:] + pos[None, :P]).reshape(eng.M, -1))
    eng.reset(kv)
    preds = []
    for s in range(steps):
        eng.head_sample()
        preds.append(eng.pred().clone())
        eng.tok_cur().copy_(g["tokens"][:, s * P:(s + 1) * P].to(DEV))          # teacher forcing
        if s + 1 < steps:
            eng.projector(); eng.llm_step()
    torch.cuda.synchronize()
    pred, ref = torch.stack(preds).cpu(), g["preds"][:, :n_img]
    err = (pred - ref).abs()
    # 1.5x what the CPU oracle itself shows against the reference: [0.21, 0.11] with CFG 4, [0.023, 0.011] without
    for s_, bound in enumerate((0.32, 0.18) if branches == 2 else (0.04, 0.03)):
        assert err[s_].mean() <= bound, (s_, err[s_].mean())
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.97


@pytest.mark.parametrize("variant", ["16x", "4x", "1x"])
def test_imagenet_combined_engine_graph_equals_per_step_path(variant):
    """BitDance.sample runs AR steps 1.. on one engine (head + proj_in + transformer) as two graph replays per step, the guidance
    scale of the linear ramp read from a device table (model_parallel.py:352-419).  Same kernels, same launch configurations as the
    per-step path over separate engines: identical latents, eager and replayed, for the 16x / 4x / 1x models."""
    from bitdance_amd.imagenet import BitDance
    c = dict({"16x": tm.TINY_IN, "4x": tm.TINY_IN_4X, "1x": tm.TINY_IN_1X}[variant])
    m = BitDance(tm.seeded_state(tm.imagenet_shapes(c), seed=41), device=DEV, **c)
    ids = torch.tensor([3, 7, 1])
    outs = {}
    for mode in ("separate", "eager", "graph"):
        m.combined_engine = mode != "separate"
        m.use_graph = mode == "graph"
        torch.manual_seed(5)
        _, tok, pred = m.sample(ids, 3, cfg_scale=3.0, return_tokens=True)
        outs[mode] = (tok.clone(), pred.clone())
    assert torch.equal(outs["eager"][1], outs["graph"][1]) and torch.equal(outs["eager"][0], outs["graph"][0])
    assert torch.equal(outs["separate"][1], outs["eager"][1])
    torch.manual_seed(5)
    _, tok2, pred2 = m.sample(ids, 3, cfg_scale=3.0, return_tokens=True)        # second call: graphs reused
    assert torch.equal(pred2, outs["graph"][1])


def test_imagenet_more_than_16_sequences():
    """Batches beyond the 16 per-sequence length slots of the step state (the eval batch is 384 classes): every imagenet
    sequence has the same length, so slot 0 serves all of them.  40 sequences: HIP transformer == torch transformer within
    bf16 noise, per-sample results independent of the batch they are in."""
    from bitdance_amd.imagenet import BitDance
    c = tm.TINY_IN
    m = BitDance(tm.seeded_state(tm.imagenet_shapes(c), seed=29), device=DEV, **c)
    ids = torch.arange(20) % 10
    g = torch.Generator().manual_seed(9)
    N, P, steps = 2, c["parallel_num"], 4
    noise = [torch.randn(N + 1, 40 if i == 0 else 20, P, c["latent_dim"], generator=g) for i in range(steps)]
    lat, tok, pred = m.sample(ids, N, cfg_scale=2.0, noise=noise, return_tokens=True)
    m.native_transformer = False
    _, _, pred_t = m.sample(ids, N, cfg_scale=2.0, noise=noise, force_tokens=tok, return_tokens=True)
    m.native_transformer = True
    _, _, pred_n = m.sample(ids, N, cfg_scale=2.0, noise=noise, force_tokens=tok, return_tokens=True)
    d = (pred_n - pred_t).abs().mean().item()
    assert d <= 0.06 * pred_t.abs().mean().item(), d
    # sample 3 alone (same noise rows) == sample 3 inside the batch of 20
    sub = [torch.cat([nz[:, 3:4], nz[:, 23:24]], dim=1) if nz.shape[1] == 40 else nz[:, 3:4] for nz in noise]
    tok_sub = torch.cat([tok[3:4], tok[23:24]])
    _, _, pred_1 = m.sample(ids[3:4], N, cfg_scale=2.0, noise=sub, force_tokens=tok_sub, return_tokens=True)
    torch.testing.assert_close(pred_1[0], pred_n[3], atol=2e-2, rtol=2e-2)


def test_c_abi_error_behaviour(eng_mod):
    """Every entry point returns 0 or a negative code with text in bd_last_error(); Python raises BitDanceHipError --
    never a silent fallback.  Unsupported shapes, empty split-K slices, out-of-schedule evals, unbound contexts."""
    from bitdance_amd._lib import BitDanceHipError, check, lib
    l = lib()
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros(1 << 20, dtype=torch.float32, device=DEV)
    assert l.bd_gemm_partial(buf.data_ptr(), 4, buf.data_ptr(), 256, 100, 1, 4, buf.data_ptr(), st) != 0      # K % 64
    assert l.bd_gemm_partial(buf.data_ptr(), 4, buf.data_ptr(), 200, 128, 1, 4, buf.data_ptr(), st) != 0      # N % (32 nw)
    assert l.bd_gemm_partial(buf.data_ptr(), 4, buf.data_ptr(), 256, 128, 3, 4, buf.data_ptr(), st) != 0      # empty K slice
    assert l.bd_last_error()
    with pytest.raises(BitDanceHipError):
        check(l.bd_pack_weight(buf.data_ptr(), buf.data_ptr(), 48, 64, 0, 48, st), "bd_pack_weight")          # rows % 32
    sd, eng = tiny_head_engine(eng_mod)
    with pytest.raises(BitDanceHipError):
        eng.head_sample()                                                                                      # no schedule set
    eng.set_schedule(3, 2.0, 1)
    with pytest.raises(BitDanceHipError):
        eng.head_eval(7)                                                                                       # outside the schedule
    ctx = l.bd_ctx_create()
    assert l.bd_ctx_bind(ctx) != 0 and l.bd_head_sample(ctx, st) != 0                                         # not finalized / bound
    l.bd_ctx_destroy(ctx)
    with pytest.raises(BitDanceHipError):
        eng_mod.Engine(None, None, None, num_images=1, branches=2, device=DEV, parallel_num=32)               # P must be 16 or 64
    # the library is still healthy afterwards
    eng.load_noise(torch.zeros(1, 4, 2, 64, 32))
    eng.reset([0, 0, 0, 0])
    eng.set_cond(torch.zeros(256, 256, device=DEV))
    eng.head_sample()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.pred()).all()


# ----------------------------------------------------------------------------------------------- drop-in constructors
def test_pipeline_from_model_dir_and_mllm_surface(tmp_path, golden_dir):
    """The REAL constructor (t2i_pipeline.py:45-75): a released-layout directory on disk (stub tokenizer, json configs,
    safetensors incl. the sharded-index form) -> BitDanceT2IPipeline(model_path) -> generate(); and the MLLModel-shaped
    surface (mllm.py:258-272) over the same native loop returns the same tokens as the pipeline."""
    from tests.model_dir import write_model_dir
    from bitdance_amd.mllm import MLLModel
    from bitdance_amd.t2i_pipeline import BitDanceT2IPipeline
    write_model_dir(str(tmp_path), sharded=True)
    pipe = BitDanceT2IPipeline(str(tmp_path), device=DEV)
    assert pipe.parallel_num == 64 and pipe.ps == 8 and pipe.vae_patch_size == 16 and pipe.hidden_size == 256
    imgs = pipe.generate("a red fox", height=512, width=512, num_sampling_steps=2, guidance_scale=3.0, num_images=1, seed=5)
    assert len(imgs) == 1 and imgs[0].size == (512, 512)           # 512 x 512 is in the reference's IMAGE_SIZE_LIST
    with pytest.raises(ValueError):
        pipe.generate("x", height=250, width=256)
    # same weights as the in-memory tiny pipeline -> same tokens for the same injected noise
    ref = tiny_pipeline()
    n, steps = 3, 2
    noise = torch.randn(steps, n + 1, 1, 64, 32, generator=torch.Generator().manual_seed(7))
    kw = dict(guidance_scale=3.0, num_sampling_steps=n, max_length=128, num_images=1, image_size=[256, 128])
    # the stub tokenizer and tm.FakeTokenizer map prompts to different ids: compare through identical id lists
    ids = lambda p_, prompt: p_._prompt_ids(prompt, "<|", [256, 128], True)
    assert len(ids(pipe, "fox")[0]) == len(ids(ref, "fox")[0])
    m = MLLModel(pipe)
    t1 = m.gen_image_block_causal("a red fox", "<|im_start|>", noise=noise, return_tokens=True, **kw)
    t2 = pipe.gen_image("a red fox", "<|im_start|>", noise=noise, return_tokens=True, **kw)
    assert torch.equal(t1, t2) and set(t1.unique().tolist()) <= {-1.0, 0.0, 1.0}
    assert m.config.head.vision_pred["parallel_num"] == 64 and m.vision_latent_dim == 32
    img = m.gen_image("a red fox", "<|im_start|>", **kw)
    assert img.shape == (1, 3, 256, 128) and torch.isfinite(img).all()
    with pytest.raises(NotImplementedError):
        m.gen_image_full_causal("x")


@pytest.mark.parametrize("native_prefill", [True, False])
def test_full_causal_loop_vs_reference(golden_dir, native_prefill):
    """MLLModel.gen_image_full_causal (modeling/mllm.py:274-384: parallel_num == 1, one token per AR step, causal prefill, no
    query tokens) on the native loop at P = 1 against the reference's own run (golden full_causal_amp, 16 AR steps of one token,
    CFG 4): teacher-forced with the reference's tokens the pre-sign latents stay within the loops' bf16 bound and the firm signs
    agree; hipGraph replay == eager launches; the dispatch of gen_image (mllm.py:268-272) reaches it; decode runs."""
    from bitdance_amd.mllm import MLLModel
    g = load(golden_dir, "full_causal_amp")
    pipe = tiny_pipeline(dict(tm.TINY_HEAD, parallel_num=1), native_prefill=native_prefill)
    m = MLLModel(pipe)
    assert m.parallel_num == 1 and m.ps == 1
    n, cfg = int(g["n_steps"]), float(g["cfg"])
    noise = g["noise"].view(16, n + 1, 1, 1, 32)
    cond_ids, uncond_ids = pipe._prompt_ids("a red fox", "<|", [64, 64], True)
    assert len(cond_ids) == len(tm.FakeTokenizer().encode("a red fox")) + 3          # <|vision_start|>, res_h, res_w: no query tokens
    embed = pipe.llm_w.sd["model.embed_tokens.weight"]
    ctx = [torch.nn.functional.embedding(torch.tensor(ids, device=DEV), embed) for ids in (cond_ids, uncond_ids)]
    tr = {}
    pipe.gen_image_from_context(ctx[0], ctx[1], guidance_scale=cfg, num_sampling_steps=n, num_images=1, image_size=[64, 64],
                                noise=noise, return_tokens=True, force_tokens=g["tokens"], trace=tr)
    pred, ref = torch.stack(tr["pred"]).cpu(), g["preds"][:, :1]
    err = (pred - ref).abs()
    assert err.mean() <= 0.25, err.mean()
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.95
    kw = dict(guidance_scale=cfg, num_sampling_steps=n, max_length=16, num_images=1, image_size=[64, 64])
    pipe.use_graph = False
    t_eager = m.gen_image_full_causal("a red fox", "<|", noise=noise, return_tokens=True, **kw).cpu()
    pipe.use_graph = True
    t_graph = m.gen_image_full_causal("a red fox", "<|", noise=noise, return_tokens=True, **kw).cpu()
    assert torch.equal(t_eager, t_graph) and t_graph.shape == (1, 16, 32) and set(t_graph.unique().tolist()) <= {-1.0, 0.0, 1.0}
    assert (t_graph == g["tokens"]).float().mean() >= 0.7                             # free-running: flips propagate
    img = m.gen_image("a red fox", "<|", **kw)                                        # parallel_num == 1 -> gen_image_full_causal
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    with pytest.raises(NotImplementedError):
        MLLModel(tiny_pipeline()).gen_image_full_causal("x")                          # a parallel_num = 64 model: the block-causal loop


def test_native_prefill_vs_reference_and_oracle(eng_mod, golden_dir):
    """The prompt passes on the step kernels (causal block + bf16 hidden-state flow) against the reference's own outputs
    (golden llm_amp: h1 = causal call over 11 tokens, h2 = all-visible call over the next 64) with the bounds of the torch
    prefill, then a decode step on top of the natively filled cache (h3); ragged prompts (a 75-token and a 70-token
    sequence in one engine) against the oracle."""
    from bitdance_amd.llm import native_block, prefill_native
    g = load(golden_dir, "llm_amp")
    sd, lw, eng = tiny_llm(eng_mod)
    emb = torch.nn.functional.embedding(g["ids"].long().to(DEV), lw.sd["model.embed_tokens.weight"])
    h1 = native_block(eng, emb, 0, causal=True)
    h2 = native_block(eng, g["blk"].to(DEV).to(torch.bfloat16), 11, causal=False)
    for got, ref in ((h1, g["h1"]), (h2, g["h2"])):
        e = (got.float().cpu() - ref).abs()
        assert e.max() <= 0.12 and e.mean() <= 1e-2, (e.max(), e.mean())
    eng.set_int("rt.emit_cond", 0)
    eng.reset([75, 75])
    eng.residual()[:128].copy_(g["dec"].reshape(128, 256).to(DEV))
    eng.llm_step()
    torch.cuda.synchronize()
    e = (eng.hidden().cpu().view(2, 64, 256) - g["h3"]).abs()
    assert e.max() <= 0.12 and e.mean() <= 1e-2, (e.max(), e.mean())
    # ragged: sequence 0 has 139 prompt tokens (75 causal + 64), sequence 1 has 134 (70 + 64): blocks of 64 with padding
    w = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    gen = torch.Generator().manual_seed(3)
    xs = [(torch.randn(t, 256, generator=gen) * 0.5).to(torch.bfloat16) for t in (139, 134)]
    hid, kv = prefill_native(eng, [x.to(DEV) for x in xs])
    assert kv == [139, 134]
    pol = Policy("autocast")
    for b, x in enumerate(xs):
        T0 = x.shape[0] - 64
        _, cache = qwen3.model_forward(w, tm.TINY_LLM, x[None, :T0], None, None, pol)
        ref, _ = qwen3.model_forward(w, tm.TINY_LLM, x[None, T0:], cache, torch.ones(1, 1, 64, x.shape[0], dtype=torch.bool), pol)
        e = (hid[b].float().cpu() - ref[0].float()).abs()
        assert e.max() <= 0.1 and e.mean() <= 8e-3, (b, e.max(), e.mean())


def test_pipeline_native_prefill_matches_torch_prefill():
    """Whole loop with the native prefill (the default) vs the torch prefill: first-patch tokens agree (the two prefills differ
    by bf16 summation order only), no second LLM weight copy is kept."""
    p_nat, p_torch = tiny_pipeline(native_prefill=True), tiny_pipeline(native_prefill=False)
    assert set(p_nat.llm_w.sd) == {"model.embed_tokens.weight"} and len(p_torch.llm_w.sd) > 10
    n, steps = 3, 2
    noise = torch.randn(steps, n + 1, 1, 64, 32, generator=torch.Generator().manual_seed(7))
    kw = dict(guidance_scale=3.0, num_sampling_steps=n, max_length=128, num_images=1, image_size=[256, 128], noise=noise)
    a = p_nat.gen_image("a red fox", "<|", return_tokens=True, **kw)
    b = p_torch.gen_image("a red fox", "<|", return_tokens=True, **kw)
    assert (a[:, :64] == b[:, :64]).float().mean().item() >= 0.95
    assert torch.isfinite(p_nat.gen_image("a red fox", "<|", **kw)).all()


@pytest.mark.parametrize("B,branches", [(1, 2), (2, 2), (2, 1)])
def test_head_sample_chain_equals_standalone_evaluations(eng_mod, B, branches):
    """head_sample runs the evaluations as a chain: y_i of the whole schedule is produced once next to cond_embed, and the final
    kernel of evaluation i writes x0 = input_proj(x_t) of evaluation i+1 (no prologue launch).  Same arithmetic per element as
    the standalone evaluations (bd_head_eval: own prologue each): bit-identical latents after every evaluation."""
    sd = tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11)
    hw = eng_mod.HeadWeights.from_state_dict(sd, DEV)
    n = 3
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(1, n + 1, B, 64, 32, generator=g)
    z = torch.randn(B * branches, 64, 256, generator=g)

    def mk(chain, y_all=True):
        eng = eng_mod.Engine(hw, None, None, num_images=B, branches=branches, device=DEV, max_tokens=64)
        eng.set_schedule(n, 2.5 if branches == 2 else 1.0, 1)
        if not y_all:
            eng.set_int("head.y_evals", 0)
        eng.load_noise(noise)
        eng.reset([0] * (B * branches))
        eng.set_cond(z.to(DEV))
        eng.set_int("rt.chain", chain)
        return eng
    a, b, c = mk(1), mk(0), mk(0, y_all=False)
    for e in (a, b, c):
        e.view("head.xt", torch.float32, (B * 64, 32)).copy_(e.noise[0, 0])
        e.head_cond()
    for i in range(n + 1):
        for e in (a, b, c):
            e.head_eval(i)
        torch.cuda.synchronize()
        xa, xb, xc = (e.view("head.xt", torch.float32, (B * 64, 32)) for e in (a, b, c))
        assert torch.equal(xa, xb) and torch.equal(xb, xc), i
    s = mk(0)
    s.head_sample()
    torch.cuda.synchronize()
    assert torch.equal(s.pred(), b.pred())


@pytest.mark.parametrize("P,B,groups,weights", [(64, 1, (1, 2, 4), "bf16"), (16, 1, (1, 16), "bf16"), (16, 2, (1, 8), "bf16"),
                                                 (64, 1, (1, 2, 4), "fp8a"), (16, 1, (1, 16), "fp8a"), (16, 2, (1, 8), "fp8a")])
def test_grouped_adaln_projection_bit_identical(eng_mod, P, B, groups, weights):
    """The adaLN projection depends on (t_i, cond) only, so head_sample computes it for G evaluations per GEMM launch
    (tune.ada_group; default 512 rows per launch).  Every row's K sum runs in the same order through the same MFMA as in the
    per-evaluation launch: the sampled latents are bit-identical for every G, eager and as a replayed graph.  fp8a: the group's
    e4m3 operand and per-row scales come from head_y_all_kernel, the 256-row form of the fp8 x fp8 kernel multiplies them."""
    cfg = dict(tm.TINY_HEAD, parallel_num=P)
    sd = tm.seeded_state(tm.head_shapes(cfg), seed=13)
    hw = eng_mod.HeadWeights.from_state_dict(sd, DEV, weights=weights)
    n = 6
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(1, n + 1, B, P, 32, generator=g)
    z = torch.randn(B * 2, P, 256, generator=g)
    ref = None
    for G in groups:
        eng = eng_mod.Engine(hw, None, None, num_images=B, branches=2, device=DEV, max_tokens=P, parallel_num=P, tune={"ada_group": G})
        eng.set_schedule(n, 3.0, 1)
        eng.load_noise(noise)
        eng.reset([0] * (2 * B))
        eng.set_cond(z.to(DEV))
        eng.head_sample()
        torch.cuda.synchronize()
        pred = eng.pred().clone()
        if ref is None:
            ref = pred
        assert torch.equal(pred, ref), G
        with torch.cuda.stream(torch.cuda.Stream()):
            eng.capture(0)
            eng.reset([0] * (2 * B))
            eng.launch(0)
            torch.cuda.synchronize()
            assert torch.equal(eng.pred(), ref), G


def test_bench_contract_on_tiny_workload():
    """bench.py's one-JSON-line contract (driver-facing): run the tiny workload end to end in a subprocess and check the fields
    the driver and the judge read, incl. roofline and cpu_baseline.parity."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    if rf["bound"] == "hbm":                                   # physically streamed bytes; the algorithmic figure beside them
        assert rf["hbm_bound_launches"]["frac"] > 0 and rf["hbm_bound_launches"]["launches"] <= rf["launches"]
        assert rf["algorithmic"]["GBs"] >= rf["achieved"] - 0.5 and rf["frac"] <= 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["parity"]["within_bounds"] is True


def test_decode_image_in_chunks_equals_one_batch():
    """decode_image decodes batches larger than ``decode_chunk`` in chunks (round 6: 32 images at 1024 px ran out of memory in the native
    decoder's work buffers): convolutions and per-sample GroupNorm make the images independent, so the chunked result is the
    whole-batch result bit for bit (autoencoder.py:172-196, t2i_pipeline.py:274-283)."""
    pipe = tiny_pipeline()
    g = torch.Generator().manual_seed(9)
    lat = torch.sign(torch.randn(5, 16 * 8, 32, generator=g)).to(DEV)
    with torch.amp.autocast("cuda", enabled=True, dtype=torch.bfloat16):
        pipe.decode_chunk = 8
        whole = pipe.decode_image(lat, image_size=[16, 8], ps=pipe.ps)
        pipe.decode_chunk = 2
        parts = pipe.decode_image(lat, image_size=[16, 8], ps=pipe.ps)
    native = pipe.ae._native_state.get("dec", {}).get("obj") is not None
    d = (whole.float() - parts.float()).abs()
    print(f"[chunked decode] native decoder: {native}; max |whole - chunked| {d.max().item():.3g} on |x| max {whole.abs().max().item():.3g}")
    assert whole.shape[0] == 5
    if native:
        assert torch.equal(whole, parts)                    # the native kernels' arithmetic does not depend on the batch
    else:
        assert d.max() <= 2e-2 * max(1.0, whole.abs().max().item())    # MIOpen may pick another algorithm per batch size
