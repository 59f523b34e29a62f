"""The oracle (CPU restatement) against the reference's own outputs (tests/golden, made by
oracle/gen_golden.py from the unmodified reference).  CPU only; no /root/reference needed."""
import os

import numpy as np
import pytest
import torch

from oracle import diff_head, gfq, pipeline, qwen3, sampler
from oracle import tiny_models as tm
from oracle.numerics import Policy


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" and z[k].ndim > 0 else z[k]) for k in z.files}


@pytest.mark.parametrize("tag", ["cfg", "nocfg"])
def test_sampler_bit_exact(golden_dir, tag):
    g = load(golden_dir, "sampler_" + tag)
    A, Cm = g["A"], g["Cm"]
    toy = lambda x, t, c: torch.tanh(x @ A + c @ Cm + t.view(-1, 1, 1))
    out = sampler.euler_maruyama(4, toy, g["c"], float(g["cfg"]), int(g["n_steps"]), list(g["noise"]))
    assert int(g["calls"]) == int(g["n_steps"]) + 1            # RNG draws: 1 + N (SURVEY 8c vii)
    assert torch.equal(out, g["out"])                           # fp32, op-for-op: bit exact


def head_weights():
    return tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11)


def test_head_forward_fp32(golden_dir):
    g = load(golden_dir, "head_fp32")
    y = diff_head.net_forward(head_weights(), g["x"], g["t"], g["c"], Policy("fp32"))
    torch.testing.assert_close(y, g["y"], atol=2e-5, rtol=1e-4)


def test_head_sample_fp32(golden_dir):
    g = load(golden_dir, "head_fp32")
    s = diff_head.sample(head_weights(), g["z"], float(g["cfg"]), int(g["n_steps"]), list(g["noise"]), Policy("fp32"))
    torch.testing.assert_close(s, g["sample"], atol=1e-4, rtol=1e-4)


def test_head_forward_amp(golden_dir):
    """bf16-autocast flow: identical rounding points => only accumulation-order noise (<= 1-2 bf16 ulp)."""
    g = load(golden_dir, "head_amp")
    tr = {}
    y = diff_head.net_forward(head_weights(), g["x"], g["t"], g["c"], Policy("autocast"), trace=tr).float()
    # after ONE block only accumulation-order noise may differ: a misplaced rounding point would flip ~half
    x1 = tr["x1"].float()
    assert (x1 != g["x1"]).float().mean() <= 0.2 and (x1 - g["x1"]).abs().mean() <= 1.5e-3
    err = (y - g["y"]).abs()                     # 4 blocks of chained bf16 roundings: a few bf16 ulp
    assert err.max() <= 5e-2 and err.mean() <= 6e-3, (err.max(), err.mean())


def test_head_sample_amp(golden_dir):
    g = load(golden_dir, "head_amp")
    s = diff_head.sample(head_weights(), g["z"], float(g["cfg"]), int(g["n_steps"]), list(g["noise"]), Policy("autocast"))
    err = (s - g["sample"]).abs()
    # CFG (x2.5) and the 1/(1-t) velocity scaling amplify the per-eval bf16 noise of the 4 chained evals
    assert err.max() <= 0.4 and err.mean() <= 6e-2, (err.max(), err.mean())


def llm_weights(dtype=torch.float32):
    return {k: v.to(dtype) for k, v in tm.seeded_state(tm.llm_shapes(tm.TINY_LLM), seed=22).items()}


def run_llm(g, pol, dtype):
    w = llm_weights(dtype)
    emb = torch.nn.functional.embedding(g["ids"].long(), w["model.embed_tokens.weight"])
    h1, cache = qwen3.model_forward(w, tm.TINY_LLM, emb, None, None, pol)
    past = cache[0][0].shape[2]
    ones = torch.ones(2, 1, 64, 64 + past + 5, dtype=torch.bool)
    h2, cache = qwen3.model_forward(w, tm.TINY_LLM, g["blk"].to(dtype), cache, ones, pol)
    ones = torch.ones(2, 1, 64, 64 + cache[0][0].shape[2], dtype=torch.bool)
    h3, cache = qwen3.model_forward(w, tm.TINY_LLM, g["dec"], cache, ones, pol)
    return h1, h2, h3, cache[0][0]


def test_llm_fp32(golden_dir):
    g = load(golden_dir, "llm_fp32")
    h1, h2, h3, k0 = run_llm(g, Policy("fp32"), torch.float32)
    for a, b in ((h1, g["h1"]), (h2, g["h2"]), (h3, g["h3"]), (k0, g["k0"])):
        torch.testing.assert_close(a, b, atol=3e-5, rtol=1e-4)


def test_llm_amp(golden_dir):
    g = load(golden_dir, "llm_amp")
    h1, h2, h3, k0 = run_llm(g, Policy("autocast"), torch.bfloat16)
    assert h1.dtype == torch.bfloat16 and h2.dtype == torch.bfloat16 and h3.dtype == torch.float32
    assert k0.dtype == torch.float32            # K promoted by the decode-time cat
    for a, b in ((h1, g["h1"]), (h2, g["h2"]), (h3, g["h3"])):
        err = (a.float() - b).abs()
        assert err.max() <= 0.12 and err.mean() <= 1e-2, (err.max(), err.mean())


def run_gen(g, pol, dtype, force=None, trace=None, P=64, hw=16, w=None, num_images=1):
    lw = llm_weights(dtype)
    tok = tm.FakeTokenizer()
    w = hw if w is None else w
    return pipeline.gen_tokens(
        lw, tm.TINY_LLM, head_weights(), tm.seeded_state(tm.proj_shapes(32, 256), seed=33),
        lw["model.embed_tokens.weight"], tok.encode("a red fox"), tok.encode("<|"),
        [tm.VISION_START, tm.RES_BASE + hw, tm.RES_BASE + w], [tm.QUERY_BASE + i for i in range(1, P)],
        h=hw, w=w, parallel_num=P, guidance_scale=float(g["cfg"]), num_sampling_steps=int(g["n_steps"]),
        num_images=num_images, noise=list(g["noise"]), pol=Policy(pol), force_tokens=force, trace=trace)


def test_gen_tokens_fp32(golden_dir):
    """Whole AR loop, fp32: every binary token equals the reference's (bit-exact index work)."""
    g = load(golden_dir, "gen_fp32")
    assert int(g["calls"]) == 4 * (int(g["n_steps"]) + 1)       # AR_steps * (1 + N) draws, SURVEY 8c(vii)
    tr = {}
    out = run_gen(g, "fp32", torch.float32, trace=tr)
    assert torch.equal(out, g["tokens"])                        # [B, h*w, C], patch-raster order
    torch.testing.assert_close(torch.stack(tr["pred"]), g["preds"][:, :1], atol=2e-4, rtol=1e-3)


def test_gen_tokens_amp_teacher_forced(golden_dir):
    """bf16 flow: a flipped near-zero latent changes every later token (SURVEY 7 'Hard parts'), so the
    loop is teacher-forced with the reference's tokens and the pre-sign latents are compared per step."""
    g = load(golden_dir, "gen_amp")
    tr = {}
    out = run_gen(g, "autocast", torch.bfloat16, force=g["tokens"], trace=tr)
    pred, ref = torch.stack(tr["pred"]), g["preds"][:, :1]
    err = (pred - ref).abs()
    # CFG mixes x_hat_u + cfg (x_hat_c - x_hat_u): bf16 noise of the evals is amplified ~(2 cfg - 1) = 7x
    assert err.mean() <= 0.2, err.mean()
    firm = ref.abs() > 0.5                                      # tokens that are not coin flips
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.97
    assert (out == g["tokens"]).float().mean() >= 0.85


def test_gen_tokens_16x_fp32(golden_dir):
    """16x models (parallel_num = 16, the <=32-token attention branch flow_head:203-208): tokens exact in fp32."""
    g = load(golden_dir, "gen16_fp32")
    tr = {}
    out = run_gen(g, "fp32", torch.float32, trace=tr, P=16, hw=8)
    assert torch.equal(out, g["tokens"])
    torch.testing.assert_close(torch.stack(tr["pred"]), g["preds"][:, :1], atol=2e-4, rtol=1e-3)


def test_gen_tokens_16x_amp_teacher_forced(golden_dir):
    g = load(golden_dir, "gen16_amp")
    tr = {}
    run_gen(g, "autocast", torch.bfloat16, force=g["tokens"], trace=tr, P=16, hw=8)
    pred, ref = torch.stack(tr["pred"]), g["preds"][:, :1]
    assert (pred - ref).abs().mean() <= 0.25
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.95


def test_posembed(golden_dir):
    g = load(golden_dir, "posembed")
    table = pipeline.sincos_1d(128, 256)
    assert torch.equal(table, g["table"])
    assert torch.equal(pipeline.pos_embed_2d(table, 4, 6, 2), g["e_4_6_2"])
    assert torch.equal(pipeline.pos_embed_2d(table, 16, 16, 8), g["e_16_16_8"])


def test_gfq_index_math(golden_dir):
    g = load(golden_dir, "gfq")
    idx = g["idx"].numpy()
    bits = gfq.indices_to_bits(idx, 8)
    assert np.array_equal(bits, g["bits"].numpy().astype(bool))
    assert np.array_equal(gfq.bits_to_indices(bits), g["back"].numpy())
    assert np.array_equal(gfq.codes_from_indices(idx, 8), g["codebook"].numpy())
    # GFQ.forward (gfq.py:196-291) on a random latent with exact +0 / -0 entries (both quantise to -1): codes and the
    # four per-codebook index streams, bit exact
    z = g["fwd_z"].numpy()                                       # [b, 32, h, w]
    zt = np.transpose(z, (0, 2, 3, 1)).reshape(z.shape[0], -1, 32)
    quant, ind = gfq.quantize_to_indices(zt, 4)
    want_q = np.transpose(g["fwd_quant"].numpy(), (0, 2, 3, 1)).reshape(z.shape[0], -1, 32)
    assert np.array_equal(quant, want_q)
    assert np.array_equal(np.transpose(ind, (2, 0, 1)).reshape(4, -1), g["fwd_indices"].numpy())
    assert quant[0, 0].tolist() == [-1.0] * 32                  # the all-zero token


# ------------------------------------------------------------- class-conditional ImageNet model
def _imagenet_run(g, pol, force=None):
    from oracle import imagenet
    w = tm.seeded_state(tm.imagenet_shapes(tm.TINY_IN), seed=29)
    noise = list(g["noise0"]) + list(g["noise1"])
    cfg = dict(tm.TINY_IN)
    return imagenet.sample(w, cfg, g["ids"], int(g["n_steps"]), float(g["cfg"]), noise, pol, force_tokens=force)


def test_imagenet_tables(golden_dir):
    """2-D RoPE table in patch-raster order and the block-causal mask are exact (layers_parallel.py:255-270,
    model_parallel.py:90-101,197-215)."""
    from oracle import imagenet
    g = load(golden_dir, "imagenet_fp32")
    c = tm.TINY_IN
    assert torch.equal(imagenet.rope_table(c), g["rope"])
    hw = c["resolution"] // 16
    assert torch.equal(imagenet.block_causal_mask(hw * hw + c["cls_token_num"] - 1, c["cls_token_num"] - 1,
                                                  c["parallel_num"]), g["mask"])
    assert int(g["calls"]) == (hw * hw // c["parallel_num"]) * (int(g["n_steps"]) + 1)      # RNG draws per AR step: 1 + N


def test_imagenet_sample_fp32(golden_dir):
    """BitDance.sample end to end in fp32: every token identical, pre-sign latents to 1e-4 (CFG ramp, the un-mixed
    first step, KV-cached block-causal transformer, head with head_dim 64 and no final sigmoid)."""
    g = load(golden_dir, "imagenet_fp32")
    lat, tokens, preds = _imagenet_run(g, Policy("fp32"))
    torch.testing.assert_close(preds, g["preds"], atol=2e-4, rtol=1e-3)
    assert torch.equal(lat, g["latent"])
    assert torch.equal(tokens[:2], torch.sign(g["preds"])[:2])


def test_imagenet_sample_amp_teacher_forced(golden_dir):
    """Emulated CUDA bf16 autocast, reference tokens fed back (teacher forcing): per-step latents at bf16-noise level."""
    g = load(golden_dir, "imagenet_amp")
    ref_tok = torch.sign(g["preds"])
    _, tokens, preds = _imagenet_run(g, Policy("autocast"), force=ref_tok)
    P, steps = tm.TINY_IN["parallel_num"], preds.shape[1] // tm.TINY_IN["parallel_num"]
    for i in range(steps):                                     # bf16 noise of the evals, amplified ~(2 cfg_i - 1) by the CFG mix
        sl = slice(i * P, (i + 1) * P)
        cfg_i = 1.0 + (float(g["cfg"]) - 1.0) * i / steps
        ref = g["preds"][:, sl]
        d = (preds[:, sl] - ref).abs()
        assert d.mean().item() <= 0.045 * max(1.0, 2 * cfg_i - 1) * ref.abs().mean().item(), (i, d.mean())
        firm = ref.abs() > 0.5
        assert (torch.sign(preds[:, sl])[firm] == torch.sign(ref)[firm]).float().mean().item() >= 0.97, i


@pytest.mark.parametrize("name,schedule", [("const", "constant"), ("nocfg", "linear")])
def test_imagenet_other_cfg_branches_fp32(golden_dir, name, schedule):
    """head_sample's other branches (model_parallel.py:356-365): constant CFG, and cfg_scale <= 1 (one branch, no
    null-class rows): tokens identical to the reference in fp32."""
    from oracle import imagenet
    g = load(golden_dir, f"imagenet_{name}_fp32")
    w = tm.seeded_state(tm.imagenet_shapes(tm.TINY_IN), seed=29)
    lat, tokens, preds = imagenet.sample(w, dict(tm.TINY_IN), g["ids"], int(g["n_steps"]), float(g["cfg"]),
                                         list(g["noise"]), Policy("fp32"), cfg_schedule=schedule)
    assert int(g["calls"]) == 4 * (int(g["n_steps"]) + 1)
    torch.testing.assert_close(preds, g["preds"], atol=2e-4, rtol=1e-3)
    assert torch.equal(lat, g["latent"])


# ------------------------------------------------------------- the other released ImageNet variants
_IN_VARIANTS = {"1x": tm.TINY_IN_1X, "4x": tm.TINY_IN_4X}


def _imagenet_variant_run(name, g, pol, force=None):
    from oracle import imagenet
    c = dict(_IN_VARIANTS[name])
    w = tm.seeded_state(tm.imagenet_shapes(c), seed=31)
    noise = list(g["noise0"]) + list(g["noise1"])
    return imagenet.sample(w, c, g["ids"], int(g["n_steps"]), float(g["cfg"]), noise, pol, force_tokens=force)


@pytest.mark.parametrize("name", ["1x", "4x"])
def test_imagenet_variants_fp32(golden_dir, name):
    """BitDance-*-1x (imagenet_gen/src/model.py:352-391: one token per step, causal transformer, MLP head diff_head.py:228-253)
    and the 4x parallel variant (model_parallel.py with parallel_num 4) end to end in fp32: RoPE table exact, RNG draws =
    AR steps x (1 + N), every token identical to the reference's, pre-sign latents to 1e-4."""
    from oracle import imagenet
    g = load(golden_dir, f"imagenet{name}_fp32")
    c = _IN_VARIANTS[name]
    assert torch.equal(imagenet.rope_table(c), g["rope"])
    hw = c["resolution"] // 16
    assert int(g["calls"]) == (hw * hw // c["parallel_num"]) * (int(g["n_steps"]) + 1)
    lat, tokens, preds = _imagenet_variant_run(name, g, Policy("fp32"))
    torch.testing.assert_close(preds, g["preds"], atol=2e-4, rtol=1e-3)
    assert torch.equal(lat, g["latent"])
    assert torch.equal(tokens[:2], torch.sign(g["preds"])[:2])


@pytest.mark.parametrize("name", ["1x", "4x"])
def test_imagenet_variants_amp_teacher_forced(golden_dir, name):
    """The same under the emulated CUDA bf16 autocast with the reference's tokens fed back: per-step latents at bf16-noise
    level (amplified by the CFG mix), firm signs agree."""
    g = load(golden_dir, f"imagenet{name}_amp")
    c = _IN_VARIANTS[name]
    ref_tok = torch.sign(g["preds"])
    _, tokens, preds = _imagenet_variant_run(name, g, Policy("autocast"), force=ref_tok)
    P = c["parallel_num"]
    steps = preds.shape[1] // P
    for i in range(steps):
        sl = slice(i * P, (i + 1) * P)
        cfg_i = 1.0 + (float(g["cfg"]) - 1.0) * i / steps
        ref = g["preds"][:, sl]
        d = (preds[:, sl] - ref).abs()
        assert d.mean().item() <= 0.05 * max(1.0, 2 * cfg_i - 1) * ref.abs().mean().item() + 2e-3, (i, d.mean())
    firm = g["preds"].abs() > 0.5
    assert (torch.sign(preds)[firm] == ref_tok[firm]).float().mean().item() >= 0.97


@pytest.mark.parametrize("name,n_img", [("genb2", 2), ("gennocfg", 1)])
def test_gen_tokens_batch2_and_nocfg_fp32(golden_dir, name, n_img):
    """gen_image with num_images = 2 (rows [cond x2 | uncond x2], per-image noise) and with guidance_scale <= 1 (single
    branch: no uncond prefill, no CFG mix), 256x128: every token equals the reference's in fp32."""
    g = load(golden_dir, name + "_fp32")
    assert int(g["calls"]) == 2 * (int(g["n_steps"]) + 1)       # 2 AR steps x (1 + N) draws, independent of num_images
    tr = {}
    out = run_gen(g, "fp32", torch.float32, trace=tr, hw=16, w=8, num_images=n_img)
    assert torch.equal(out, g["tokens"])
    torch.testing.assert_close(torch.stack(tr["pred"]), g["preds"][:, :n_img], atol=2e-4, rtol=1e-3)


# ------------------------------------------------------------- interleaved text + image context
_EDIT_TEXT = "<|im_start|>user\nmake the fox red<|im_end|>\n<|im_start|>assistant\n"
_EDIT_PLAN = [{"type": "text", "from": "user"}, {"type": "image", "from": "user"}, {"type": "image", "from": "model"}]


def _interleaved_ctx(g, pol, dtype):
    from oracle import pipeline as op
    proj = tm.seeded_state(tm.proj_shapes(32, 256), seed=33)
    llm = {k: v.to(dtype) for k, v in tm.seeded_state(tm.llm_shapes(tm.TINY_LLM), seed=22).items()}
    emb_img = op.encode_image(proj, g["image_latents"], (8, 8), 256, 8, pol)
    tok = tm.FakeTokenizer()
    c, u = op.interleaved_context(llm["model.embed_tokens.weight"], _EDIT_PLAN, [_EDIT_TEXT], [emb_img], tok.encode,
                                  start_of_image=tm.VISION_START, end_of_image=tm.VISION_END,
                                  res_ids=(tm.RES_BASE + 16, tm.RES_BASE + 16), query_ids=[tm.QUERY_BASE + i for i in range(1, 64)],
                                  cfg_on=True)
    return llm, proj, emb_img, c, u


def test_interleaved_context_and_encode_image(golden_dir):
    """mllm.py:899-930 (encode_image after the tokenizer) and the context assembly of forward_inference_block_causal
    (:719-745,865-895) for an editing plan [user text, user image, generated image]: the image embeddings equal the reference's,
    and the four prefill calls the reference made (cond causal / cond last block / uncond causal / uncond last block) have exactly
    the lengths of the assembled contexts -- the unconditional one is the text without its first user block."""
    from oracle import pipeline as op
    g = load(golden_dir, "interleaved_fp32")
    llm, proj, emb_img, c, u = _interleaved_ctx(g, Policy("fp32"), torch.float32)
    torch.testing.assert_close(emb_img, g["image_embeds"], atol=1e-5, rtol=1e-5)
    assert [c.shape[0] - 64, 64, u.shape[0] - 64, 64] == g["prefill_lens"].tolist()
    assert op.remove_first_user_block(_EDIT_TEXT) == "<|im_start|>assistant\n" and op.remove_first_user_block("abc") == "abc"
    q = torch.arange(2 * 16 * 8, dtype=torch.float32).view(2, 16, 8)
    t = op.image_latents_to_tokens(q, 8)                                  # 'c (h p1) (w p2) -> (h w p1 p2) c'
    assert t.shape == (128, 2) and t[0].tolist() == [0.0, 128.0] and t[1].tolist() == [1.0, 129.0] and t[8].tolist() == [8.0, 136.0] \
        and t[64].tolist() == [64.0, 192.0]
    ga = load(golden_dir, "interleaved_amp")
    _, _, emb_a, _, _ = _interleaved_ctx(ga, Policy("autocast"), torch.bfloat16)
    assert emb_a.dtype == torch.bfloat16 and (emb_a.float() - ga["image_embeds"]).abs().max().item() <= 0.04


def test_interleaved_image_generation_fp32(golden_dir):
    """The image-generating part of MLLModel.forward_inference_block_causal (mllm.py:745-864) on that context in fp32: every
    token of the generated image equals the reference's, RNG draws = AR steps x (1 + N)."""
    from oracle import pipeline as op
    g = load(golden_dir, "interleaved_fp32")
    llm, proj, _, c, u = _interleaved_ctx(g, Policy("fp32"), torch.float32)
    head = tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11)
    assert int(g["calls"]) == 4 * (int(g["n_steps"]) + 1)
    tr = {}
    out = op.gen_tokens_from_context(llm, tm.TINY_LLM, head, proj, c, u, h=16, w=16, parallel_num=64, guidance_scale=float(g["cfg"]),
                                     num_sampling_steps=int(g["n_steps"]), num_images=1, noise=list(g["noise"]), pol=Policy("fp32"),
                                     trace=tr)
    assert torch.equal(out, g["tokens"])
    torch.testing.assert_close(torch.stack(tr["pred"]), g["preds"][:, :1], atol=2e-4, rtol=1e-3)


def test_interleaved_image_generation_amp_teacher_forced(golden_dir):
    """The same under the emulated bf16 autocast with the reference's tokens fed back: per-step latents within the bound of the
    text-to-image loop (CFG-amplified bf16 noise)."""
    from oracle import pipeline as op
    g = load(golden_dir, "interleaved_amp")
    llm, proj, _, c, u = _interleaved_ctx(g, Policy("autocast"), torch.bfloat16)
    head = tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11)
    tr = {}
    op.gen_tokens_from_context(llm, tm.TINY_LLM, head, proj, c, u, h=16, w=16, parallel_num=64, guidance_scale=float(g["cfg"]),
                               num_sampling_steps=int(g["n_steps"]), num_images=1, noise=list(g["noise"]), pol=Policy("autocast"),
                               trace=tr, force_tokens=g["tokens"])
    pred, ref = torch.stack(tr["pred"]), g["preds"][:, :1]
    assert (pred - ref).abs().mean() <= 0.25, (pred - ref).abs().mean()     # the bound of the 16x / T2I loops (CFG amplifies x7)
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.95


def test_text_sampler_helpers(golden_dir):
    """modeling/utils.py:64-124 (top_k_top_p_filtering / sample_codebook): the oracle's row-by-row restatement and the product's
    vectorised one (bitdance_amd.mllm, plain torch: runs on CPU here) against the reference's outputs -- filtered logits exact
    for top-k (incl. a tie at the threshold and k > vocabulary), top-p, both, min_tokens_to_keep; greedy tokens and embeddings;
    the multinomial draw under the same CPU seed."""
    from bitdance_amd.mllm import MLLModel
    from oracle import pipeline as op
    g = load(golden_dir, "text_sampling")
    cases = {"k5": dict(top_k=5), "p90": dict(top_p=0.9), "k20p50": dict(top_k=20, top_p=0.5),
             "p10keep3": dict(top_p=0.1, min_tokens_to_keep=3), "k500": dict(top_k=500)}
    for name, kw in cases.items():
        want = g["filt_" + name]
        assert torch.equal(op.filter_logits(g["logits"], **kw), want), name
        assert torch.equal(MLLModel.top_k_top_p_filtering(g["logits"].clone(), **kw), want), name
    tok, emb = op.sample_codebook_greedy(g["logits"], g["book"], 0.7, 10, 0.8)
    assert torch.equal(tok, g["greedy_tokens"]) and torch.equal(emb, g["greedy_embeds"])
    book = lambda t: g["book"][t]
    tok, emb = MLLModel.sample_codebook(g["logits"].clone(), "text", book, do_sample=False, temperature=0.7, top_k=10, top_p=0.8)
    assert torch.equal(tok, g["greedy_tokens"]) and torch.equal(emb, g["greedy_embeds"])
    torch.manual_seed(5)
    tok, emb = MLLModel.sample_codebook(g["logits"].clone(), "text", book, do_sample=True, temperature=1.3, top_k=12, top_p=0.95)
    assert torch.equal(tok, g["sampled_tokens"]) and torch.equal(emb, g["sampled_embeds"])


def test_mllm_gen_image_is_the_same_loop(golden_dir):
    """modeling/mllm.py:386-501 (MLLModel.gen_image_block_causal) on the same components, prompt and injected noise
    produced exactly the tokens of t2i_pipeline.gen_image (golden gen_fp32) with the same number of RNG draws, so one
    oracle / one native loop covers both copies of the hot path the north star names."""
    a, b = load(golden_dir, "mllm_equiv"), load(golden_dir, "gen_fp32")
    assert int(a["calls"]) == int(b["calls"]) and torch.equal(a["tokens"], b["tokens"])


def test_full_causal_loop_fp32_and_amp(golden_dir):
    """MLLModel.gen_image_full_causal (modeling/mllm.py:274-384: the loop of parallel_num == 1 models -- one token per AR step,
    causal prefill, no query tokens, ps = 1) on the reference with a parallel_num = 1 head (goldens full_causal_*; the generator
    also asserts the reference's t2i_pipeline.gen_image gives the same tokens at parallel_num = 1).  The oracle's one loop with
    P = 1 reproduces it: fp32 every token identical and the pre-sign latents to 2e-4; under the emulated autocast teacher-forced
    within the loops' bf16 bound.  So the full-causal entry point is the same native loop at P = 1 (bitdance_amd/mllm.py)."""
    g = load(golden_dir, "full_causal_fp32")
    assert int(g["calls"]) == 16 * (int(g["n_steps"]) + 1) and g["tokens"].shape == (1, 16, 32)
    tr = {}
    out = run_gen(g, "fp32", torch.float32, trace=tr, P=1, hw=4)
    assert torch.equal(out, g["tokens"])
    torch.testing.assert_close(torch.stack(tr["pred"]), g["preds"][:, :1], atol=2e-4, rtol=1e-3)
    g = load(golden_dir, "full_causal_amp")
    tr = {}
    run_gen(g, "autocast", torch.bfloat16, force=g["tokens"], trace=tr, P=1, hw=4)
    pred, ref = torch.stack(tr["pred"]), g["preds"][:, :1]
    assert (pred - ref).abs().mean() <= 0.25, (pred - ref).abs().mean()
    firm = ref.abs() > 0.5
    assert (torch.sign(pred)[firm] == torch.sign(ref)[firm]).float().mean() >= 0.95


def test_autoencoder_oracle_matches_reference(golden_dir):
    """oracle/autoencoder.py (functional restatement of the tokenizer's Encoder / Decoder from the state dict) against the unmodified
    reference module's outputs on seeded weights (tests/golden/ae_roundtrip.npz: image -> encoder latent -> sign; token map -> decoder
    image), fp32 policy; the autocast policy stays within bf16 noise of it."""
    import ast
    from oracle import autoencoder as oae
    g = load(golden_dir, "ae_roundtrip")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = tm.seeded_state(shapes, seed=44, gain=1.4)
    cfg = tm.TINY_AE["ddconfig"]
    fp32 = Policy("fp32")
    with torch.no_grad():
        h = oae.encoder_forward(fp32, sd, cfg, g["image"])
        q = oae.encode(fp32, sd, cfg, g["image"])
        dec = oae.decoder_forward(fp32, sd, cfg, g["quant"])
        torch.testing.assert_close(h, g["henc"], atol=1e-4, rtol=1e-4)
        assert (q == g["quant"]).float().mean() >= 0.999
        torch.testing.assert_close(dec, g["dec"], atol=2e-4, rtol=1e-3)
        amp = Policy("autocast")
        ha = oae.encoder_forward(amp, sd, cfg, g["image"])
        da = oae.decoder_forward(amp, sd, cfg, g["quant"])
    assert ha.dtype == torch.bfloat16 and da.dtype == torch.bfloat16
    assert (ha.float() - g["henc"]).abs().mean().item() <= 0.02 * g["henc"].abs().mean().item() + 2e-3
    assert (da.float() - g["dec"]).abs().mean().item() <= 0.03 * g["dec"].abs().mean().item() + 2e-3


def test_gan_decoder_variant_matches_reference(golden_dir):
    """VQModel(gan_decoder=True) (autoencoder.py:279-351: conv_in over the token map concatenated with a fresh torch.randn_like noise
    map): the product's torch module (checkpoint-compatible: the reference's state-dict keys and shapes) and the oracle restatement
    against the unmodified reference's decode with the global generator seeded the same way (tests/golden/ae_gan.npz) -- the noise
    draw comes out of the same generator in the same order (one normal of the token map's shape per decode)."""
    import ast
    from bitdance_amd.autoencoder import VQModel
    from oracle import autoencoder as oae
    g = load(golden_dir, "ae_gan")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    ae = VQModel(**tm.TINY_AE, gan_decoder=True).eval()
    assert {k: tuple(v.shape) for k, v in ae.state_dict().items()} == shapes          # drop-in for the reference checkpoint
    sd = tm.seeded_state(shapes, seed=47, gain=1.4)
    ae.load_state_dict(sd)
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        dec = ae.decode(g["quant"])
    torch.testing.assert_close(dec, g["dec"], atol=2e-4, rtol=1e-3)
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        od = oae.decoder_forward(Policy("fp32"), sd, tm.TINY_AE["ddconfig"], g["quant"])      # draws its own noise, like the reference
        on = oae.decoder_forward(Policy("fp32"), sd, tm.TINY_AE["ddconfig"], g["quant"], noise=g["noise"])
    torch.testing.assert_close(od, g["dec"], atol=2e-4, rtol=1e-3)
    assert torch.equal(od, on)


def test_autoencoder_oracle_config1(golden_dir):
    """BASELINE config 1 through the ORACLE: the ae_d16c32 tokenizer (released size: ch 256, ch_mult [1,1,2,2,4], 4 res-blocks) on one
    256 x 256 image, CPU fp32 -- encode -> binary quantise -> decode reproduces the reference's binary latent bit for bit and its
    decoded pixels to fp32 conv-order noise (tests/golden/ae_c1.npz, generated by the unmodified reference module)."""
    from oracle import autoencoder as oae
    z = np.load(os.path.join(golden_dir, "ae_c1.npz"))
    from bitdance_amd.autoencoder import VQModel              # (only for the key / shape table of the released architecture)
    shapes = {k: tuple(v.shape) for k, v in VQModel(**tm.AE_D16C32).state_dict().items()}
    assert len(shapes) == int(z["n_tensors"])
    sd = tm.seeded_state(shapes, seed=61, gain=1.4)
    cfg = tm.AE_D16C32["ddconfig"]
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    fp32 = Policy("fp32")
    with torch.no_grad():
        q = oae.encode(fp32, sd, cfg, img)
        dec = oae.decoder_forward(fp32, sd, cfg, q)
    assert tuple(q.shape) == (1, 32, 16, 16)
    assert np.array_equal(np.packbits((q > 0).numpy().reshape(-1)), z["quant_bits"])          # binary tokens: exact
    got = dec.reshape(-1)[torch.from_numpy(z["sample_idx"])]
    torch.testing.assert_close(got, torch.from_numpy(z["dec_samples"]), atol=1e-3, rtol=1e-3)
    assert abs(float(dec.mean()) - float(z["dec_mean"])) < 1e-3 and abs(float(dec.std()) - float(z["dec_std"])) < 1e-3
