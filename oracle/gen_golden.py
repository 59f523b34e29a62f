"""Generate tests/golden/*.npz by running the UNMODIFIED reference (test infrastructure).

    python -m oracle.gen_golden            # needs /root/reference; writes tests/golden/

The reference ships no tests/golden vectors, so these vectors -- the
reference's own outputs on seeded tiny models with injected noise -- are what pins the oracle.
Two regimes are recorded:
  *_fp32 : the reference exactly as it runs on CPU (autocast("cuda") inert, fp32 weights)
  *_amp  : the reference under the CUDA bf16-autocast policy emulated on CPU
           (oracle/ref_harness.py CudaAutocastOnCpu), LLM weights in bf16 as from_pretrained(bf16)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import ref_harness as rh
from . import tiny_models as tm

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().to(torch.float32).cpu().numpy() if v.is_floating_point() else v.cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def check_shapes(module, shapes, what):
    sd = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    want = {k: tuple(v) for k, v in shapes.items()}
    assert sd == want, f"{what}: state_dict mismatch: {set(sd) ^ set(want)}"


def build_head(cfg=None):
    from modeling.vision_head.flow_head_parallel_x import DiffHead
    cfg = cfg or tm.TINY_HEAD
    head = DiffHead(**cfg).eval()
    shapes = tm.head_shapes(cfg)
    check_shapes(head, shapes, "DiffHead")
    head.load_state_dict(tm.seeded_state(shapes, seed=11))
    return head


def build_llm(dtype):
    from transformers import Qwen3Config, Qwen3ForCausalLM
    c = tm.TINY_LLM
    cfg = Qwen3Config(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
                      intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                      num_attention_heads=c["num_attention_heads"], num_key_value_heads=c["num_key_value_heads"],
                      head_dim=c["head_dim"], rms_norm_eps=c["rms_norm_eps"], max_position_embeddings=8192,
                      rope_parameters={"rope_type": "default", "rope_theta": c["rope_theta"]},
                      tie_word_embeddings=False, attention_bias=False)
    m = Qwen3ForCausalLM(cfg).eval()
    shapes = tm.llm_shapes(c)
    sd = tm.seeded_state(shapes, seed=22)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items() if k != "lm_head.weight"}
    assert got == {k: tuple(v) for k, v in shapes.items()}, set(got) ^ set(shapes)
    sd["lm_head.weight"] = torch.zeros(c["vocab_size"], c["hidden_size"])
    m.load_state_dict(sd)
    m = m.to(dtype)
    rh.add_mask_slice_hook(m.model)
    return m


def build_projector():
    from modeling.utils import MLPconnector
    p = MLPconnector(32, tm.TINY_LLM["hidden_size"], "gelu_pytorch_tanh").eval()
    shapes = tm.proj_shapes(32, tm.TINY_LLM["hidden_size"])
    check_shapes(p, shapes, "MLPconnector")
    p.load_state_dict(tm.seeded_state(shapes, seed=33))
    return p


def build_ae():
    from modeling.vision_encoder.autoencoder import VQModel
    ae = VQModel(**tm.TINY_AE).eval()
    shapes = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(tm.seeded_state(shapes, seed=44, gain=1.4))
    return ae, shapes


def build_pipeline(dtype, head_cfg=None):
    from modeling.t2i_pipeline import BitDanceT2IPipeline
    head_cfg = head_cfg or tm.TINY_HEAD
    pipe = object.__new__(BitDanceT2IPipeline)
    pipe.device = "cpu"
    pipe.tokenizer = tm.FakeTokenizer()
    pipe.llm_model = build_llm(dtype)
    pipe.hidden_size = tm.TINY_LLM["hidden_size"]
    pipe.ae, _ = build_ae()
    pipe.vae_patch_size = 16
    pipe.vision_head = build_head(head_cfg)
    pipe.parallel_num = head_cfg["parallel_num"]
    pipe.ps = int(pipe.parallel_num ** 0.5)
    pipe.embed_vision_mlp = build_projector()
    pipe.build_pos_embed()
    return pipe


def gen_sampler():
    from modeling.vision_head import sampling_x as sx
    g = torch.Generator().manual_seed(5)
    A = torch.randn(4, 4, generator=g)
    Cm = torch.randn(6, 4, generator=g) * 0.3
    c = torch.randn(4, 8, 6, generator=g)

    def toy(x, t, cc):
        return torch.tanh(x @ A + cc @ Cm + t.view(-1, 1, 1))

    for tag, cfg, cc in (("cfg", 3.0, c), ("nocfg", 1.0, c[:2])):
        with rh.ReplayNoise(seed=7) as rn:
            out = sx.euler_maruyama(4, toy, cc, cfg, num_sampling_steps=6)
        save(f"sampler_{tag}", A=A, Cm=Cm, c=cc, cfg=np.float32(cfg), n_steps=6, out=out,
             noise=torch.stack(rn.record), calls=rn.calls)


def gen_head():
    head = build_head()
    g = torch.Generator().manual_seed(101)
    x = torch.randn(4, 64, 32, generator=g)
    t = torch.tensor([0.0, 0.3, 0.62, 0.95])
    c = torch.randn(4, 64, 256, generator=g)
    z = torch.randn(4, 64, 256, generator=g)
    for tag in ("fp32", "amp"):
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        grab = {}
        hk = head.net.res_blocks[0].register_forward_hook(lambda m, i, o: grab.__setitem__("x1", o.detach().clone()))
        with torch.no_grad(), ctx:
            y = head.net(x, t, c)
            hk.remove()
            with rh.ReplayNoise(seed=9) as rn:
                s = head.sample(z, cfg=2.5, num_sampling_steps=3)
        save(f"head_{tag}", x=x, t=t, c=c, y=y, x1=grab["x1"], z=z, sample=s, noise=torch.stack(rn.record),
             cfg=np.float32(2.5), n_steps=3)


def gen_llm():
    g = torch.Generator().manual_seed(202)
    ids = torch.randint(0, 256, (2, 11), generator=g)
    blk = torch.randn(2, 64, 256, generator=g) * 0.5
    dec = torch.randn(2, 64, 256, generator=g) * 0.5
    for tag, dtype in (("fp32", torch.float32), ("amp", torch.bfloat16)):
        m = build_llm(dtype)
        model = m.model
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        with torch.no_grad(), ctx:
            emb = model.embed_tokens(ids)
            o1 = model(inputs_embeds=emb, use_cache=True)
            pkv = o1.past_key_values
            past = pkv[0][0].shape[2]
            ones = torch.ones(2, 1, 64, 64 + past + 5, dtype=torch.bool)     # oversize on purpose (shim 3)
            o2 = model(inputs_embeds=blk.to(dtype), past_key_values=pkv, use_cache=True, attention_mask=ones)
            pkv = o2.past_key_values
            ones = torch.ones(2, 1, 64, 64 + pkv[0][0].shape[2], dtype=torch.bool)
            o3 = model(inputs_embeds=dec, past_key_values=pkv, use_cache=True, attention_mask=ones)
            k0 = o3.past_key_values[0][0]
        save(f"llm_{tag}", ids=ids, blk=blk, dec=dec, h1=o1.last_hidden_state, h2=o2.last_hidden_state,
             h3=o3.last_hidden_state, k0=k0)


def gen_pipeline():
    jobs = [("gen_fp32", torch.float32, None, [256, 256], 256), ("gen_amp", torch.bfloat16, None, [256, 256], 256),
            ("gen16_fp32", torch.float32, tm.TINY_HEAD16, [128, 128], 64), ("gen16_amp", torch.bfloat16, tm.TINY_HEAD16, [128, 128], 64),
            # two images per call (M = 256 rows with CFG) and guidance_scale <= 1 (single branch, no uncond prefill)
            ("genb2_fp32", torch.float32, None, [256, 128], 128), ("genb2_amp", torch.bfloat16, None, [256, 128], 128),
            ("gennocfg_fp32", torch.float32, None, [256, 128], 128), ("gennocfg_amp", torch.bfloat16, None, [256, 128], 128)]
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for name, dtype, hcfg, size, max_len in jobs:
        if only and not name.startswith(only):
            continue
        n_img = 2 if name.startswith("genb2") else 1
        gs = 1.0 if name.startswith("gennocfg") else 4.0
        tag = name.split("_")[1]
        pipe = build_pipeline(dtype, hcfg)
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        captured = {}
        orig_decode = pipe.decode_image

        def spy(lat, image_size=None, ps=1):
            captured["tokens"] = lat.clone()
            return orig_decode(lat, image_size, ps)

        pipe.decode_image = spy
        preds = []
        orig_sample = pipe.vision_head.sample

        def rec_sample(*a, **k):
            o = orig_sample(*a, **k)
            preds.append(o.detach().clone())
            return o

        pipe.vision_head.sample = rec_sample
        with torch.no_grad(), ctx, rh.ReplayNoise(seed=13) as rn:
            img = pipe.gen_image(cond_prompt="a red fox", uncond_prompt="<|", guidance_scale=gs,
                                 num_sampling_steps=4, max_length=max_len, num_images=n_img, image_size=size)
        extra = {} if name.startswith(("genb2", "gennocfg")) else {"image": img}     # keep the newer fixtures small
        save(name, tokens=captured["tokens"], preds=torch.stack(preds), noise=torch.stack(rn.record),
             calls=rn.calls, cfg=np.float32(gs), n_steps=4, **extra)


def gen_mllm_equiv():
    """MLLModel.gen_image_block_causal (modeling/mllm.py:386-501), the second copy of the hot loop the north star names,
    on the same components / prompt / injected noise as gen_fp32: its tokens are recorded so the tests can pin
    "t2i_pipeline.gen_image == mllm.gen_image" (SURVEY 8a P1).  MLLModel.__init__ needs liger_kernel / omegaconf
    (absent): the instance is assembled with object.__new__ and exactly the attributes the loop reads."""
    from types import SimpleNamespace
    import modeling.mllm as mm

    class Tiny(mm.MLLModel):
        device = "cpu"

    base = build_pipeline(torch.float32)
    tok = tm.FakeTokenizer()
    tk = SimpleNamespace(encode=tok.encode, start_of_image_id=tm.VISION_START)
    for n in range(1, 129):
        setattr(tk, f"res_{n}_id", tm.RES_BASE + n)
    for i in range(1, 64):
        setattr(tk, f"query_{i}_id", tm.QUERY_BASE + i)
    m = object.__new__(Tiny)
    torch.nn.Module.__init__(m)
    m.tokenizer = tk
    m.config = SimpleNamespace(vit_patch_size=16)
    m.llm_model = base.llm_model
    m.hidden_size = base.hidden_size
    m.parallel_num, m.ps = 64, 8
    m.vision_diffusion_head = base.vision_head
    m.embed_vision_mlp = base.embed_vision_mlp
    m.register_buffer("pos_embed_1d", m._get_1d_sincos_pos_embed(m.hidden_size // 2, 256), persistent=False)
    captured = {}
    m.decode_image = lambda lat, image_size=None, ps=1: captured.setdefault("tokens", lat.clone())
    with torch.no_grad(), rh.ReplayNoise(seed=13) as rn:
        m.gen_image_block_causal("a red fox", "<|", 4.0, 4, 256, 1, [256, 256], False)
    ref = np.load(os.path.join(OUT, "gen_fp32.npz"))
    assert rn.calls == int(ref["calls"]) and np.array_equal(captured["tokens"].numpy(), ref["tokens"]), \
        "mllm.gen_image_block_causal and t2i_pipeline.gen_image disagree"
    save("mllm_equiv", tokens=captured["tokens"], calls=rn.calls)


def gen_full_causal():
    """MLLModel.gen_image_full_causal (modeling/mllm.py:274-384): the loop gen_image dispatches to when the head's
    parallel_num is 1 (mllm.py:268-272) -- one token per AR step, plain causal prefill, no query tokens, ps = 1.  Only the
    diffusion_parallel_x head builds ``vision_diffusion_head`` (mllm.py:133-150), so that is the head it can run with.  Tiny
    components of gen_fp32 with a parallel_num = 1 head, 64 x 64 px = 4 x 4 tokens = 16 AR steps, CFG 4, 4 sampling steps;
    fp32 as on CPU and under the emulated CUDA autocast.  The reference's t2i_pipeline.gen_image on the same components has
    to produce the same tokens with the same RNG draws (block-causal with blocks of one token IS causal)."""
    from types import SimpleNamespace
    import modeling.mllm as mm

    class Tiny(mm.MLLModel):
        device = "cpu"

    head1 = dict(tm.TINY_HEAD, parallel_num=1)
    for tag, dtype in (("fp32", torch.float32), ("amp", torch.bfloat16)):
        base = build_pipeline(dtype, head1)
        tok = tm.FakeTokenizer()
        tk = SimpleNamespace(encode=tok.encode, start_of_image_id=tm.VISION_START)
        for n in range(1, 129):
            setattr(tk, f"res_{n}_id", tm.RES_BASE + n)
        m = object.__new__(Tiny)
        torch.nn.Module.__init__(m)
        m.tokenizer = tk
        m.config = SimpleNamespace(vit_patch_size=16)
        m.llm_model = base.llm_model
        m.hidden_size = base.hidden_size
        m.vision_head_type = "diffusion_parallel_x"
        m.parallel_num, m.ps = 1, 1
        m.vision_diffusion_head = base.vision_head
        m.embed_vision_mlp = base.embed_vision_mlp
        m.register_buffer("pos_embed_1d", m._get_1d_sincos_pos_embed(m.hidden_size // 2, 256), persistent=False)
        captured = {}
        m.decode_image = lambda lat, image_size=None, ps=1: captured.setdefault("tokens", lat.clone())
        preds = []
        orig_sample = base.vision_head.sample

        def rec_sample(*a, **k):
            o = orig_sample(*a, **k)
            preds.append(o.detach().clone())
            return o

        base.vision_head.sample = rec_sample
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        with torch.no_grad(), ctx, rh.ReplayNoise(seed=17) as rn:
            m.gen_image_full_causal("a red fox", "<|", 4.0, 4, 16, 1, [64, 64], False)
        n_model = len(preds)
        # the same loop through the T2I pipeline's block-causal code at parallel_num = 1, same injected noise
        cap2 = {}
        base.decode_image = lambda lat, image_size=None, ps=1: cap2.setdefault("tokens", lat.clone())
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        with torch.no_grad(), ctx, rh.ReplayNoise(seq=[t.clone() for t in rn.record]) as rn2:
            base.gen_image(cond_prompt="a red fox", uncond_prompt="<|", guidance_scale=4.0, num_sampling_steps=4,
                           max_length=16, num_images=1, image_size=[64, 64])
        assert rn2.calls == rn.calls and torch.equal(cap2["tokens"], captured["tokens"]), \
            "mllm.gen_image_full_causal and t2i_pipeline.gen_image (parallel_num = 1) disagree"
        save(f"full_causal_{tag}", tokens=captured["tokens"], preds=torch.stack(preds[:n_model]), noise=torch.stack(rn.record),
             calls=rn.calls, cfg=np.float32(4.0), n_steps=4)


def gen_interleaved():
    """MLLModel.forward_inference_block_causal (modeling/mllm.py:695-897) for an image-EDITING plan: a user text, a user image
    (encode_image :899-930 = VQModel.vt_forward -> MLPconnector -> + 2-D pos embed) and a model-generated image, CFG on (the
    unconditional context drops the first user block of the text, utils.py:206-216, and keeps the image).  Same tiny components
    as gen_fp32 / gen_amp; 256x256 output (4 AR steps x 4 sampling steps), 128x128 input image."""
    from types import SimpleNamespace
    import modeling.mllm as mm

    class Tiny(mm.MLLModel):
        device = "cpu"

    for tag, dtype in (("fp32", torch.float32), ("amp", torch.bfloat16)):
        base = build_pipeline(dtype)
        tok = tm.FakeTokenizer()
        tk = SimpleNamespace(encode=tok.encode, start_of_image_id=tm.VISION_START, end_of_image_id=tm.VISION_END,
                             im_start_id=tm.IM_START, im_end_id=tm.IM_END)
        for n in range(1, 129):
            setattr(tk, f"res_{n}_id", tm.RES_BASE + n)
        for i in range(1, 64):
            setattr(tk, f"query_{i}_id", tm.QUERY_BASE + i)
        m = object.__new__(Tiny)
        torch.nn.Module.__init__(m)
        m.tokenizer = tk
        m.config = SimpleNamespace(vit_patch_size=16, encoder={"max_bs": 32})
        m.head_config = {}
        m.llm_model = base.llm_model
        m.hidden_size = base.hidden_size
        m.parallel_num, m.ps = 64, 8
        m.vision_head_type = "diffusion_parallel_x"
        m.vision_diffusion_head = base.vision_head
        m.embed_vision_mlp = base.embed_vision_mlp
        m.vision_encoder = base.ae
        m.register_buffer("pos_embed_1d", m._get_1d_sincos_pos_embed(m.hidden_size // 2, 256), persistent=False)
        m.eval()
        captured, preds, ctx_len = {}, [], []
        m.decode_image = lambda lat, image_size=None, ps=1: captured.setdefault("tokens", lat.clone())
        orig_sample = base.vision_head.sample

        def rec_sample(*a, **k):
            o = orig_sample(*a, **k)
            preds.append(o.detach().clone())
            return o

        base.vision_head.sample = rec_sample
        orig_fwd = base.llm_model.model.forward

        def rec_fwd(*a, **k):
            ctx_len.append(int(k["inputs_embeds"].shape[1]))
            return orig_fwd(*a, **k)

        base.llm_model.model.forward = rec_fwd
        g = torch.Generator().manual_seed(77)
        img = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
        text = "<|im_start|>user\nmake the fox red<|im_end|>\n<|im_start|>assistant\n"
        plan = [{"type": "text", "from": "user"}, {"type": "image", "from": "user"}, {"type": "image", "from": "model"}]
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        with torch.no_grad(), ctx, rh.ReplayNoise(seed=21) as rn:
            emb_img, lat_img = m.encode_image([img])
            m.forward_inference_block_causal(plan, [text], [img], max_length_vision=256, sample_steps=4, image_size=[256, 256],
                                             cfg_scale=4.0)
        save(f"interleaved_{tag}", tokens=captured["tokens"], preds=torch.stack(preds), noise=torch.stack(rn.record), calls=rn.calls,
             cfg=np.float32(4.0), n_steps=4, image=img, image_latents=lat_img, image_embeds=emb_img.float(),
             prefill_lens=np.array(ctx_len[:4]))


def gen_text_sampling():
    """modeling/utils.py:64-124 (top_k_top_p_filtering, sample_codebook): the token sampler of the reference's text / standard
    vision-head branches.  Deterministic pieces recorded exactly; the multinomial draw under a fixed CPU seed."""
    from modeling.utils import sample_codebook, top_k_top_p_filtering
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(4, 97, generator=g) * 3.0
    logits[1, 5] = logits[1, 6]                                   # a tie at the top-k threshold
    logits[2] = logits[2].sort(descending=True)[0]
    book = torch.nn.Embedding(97, 8)
    with torch.no_grad():
        book.weight.copy_(torch.randn(97, 8, generator=g))
    out = {}
    for name, kw in (("k5", dict(top_k=5)), ("p90", dict(top_p=0.9)), ("k20p50", dict(top_k=20, top_p=0.5)),
                     ("p10keep3", dict(top_p=0.1, min_tokens_to_keep=3)), ("k500", dict(top_k=500))):
        out["filt_" + name] = top_k_top_p_filtering(logits.clone(), **kw)
    with torch.no_grad():
        tok_g, emb_g = sample_codebook(logits.clone(), "text", book, do_sample=False, temperature=0.7, top_k=10, top_p=0.8)
        torch.manual_seed(5)
        tok_s, emb_s = sample_codebook(logits.clone(), "text", book, do_sample=True, temperature=1.3, top_k=12, top_p=0.95)
    save("text_sampling", logits=logits, book=book.weight.detach(), greedy_tokens=tok_g, greedy_embeds=emb_g, sampled_tokens=tok_s,
         sampled_embeds=emb_s, **out)


def gen_ae_c1():
    """BASELINE config 1: ae_d16c32 encode -> binary quantise -> decode of one 256x256 image on CPU (fp32), the
    reference VQModel at full size with seeded weights.  Stored small: the packed sign pattern of the 32x16x16 latent,
    4096 sampled output pixels and the output statistics."""
    from modeling.vision_encoder.autoencoder import VQModel
    ae = VQModel(**tm.AE_D16C32).eval()
    shapes = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(tm.seeded_state(shapes, seed=61, gain=1.4))
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        q = ae.encode(img)
        dec = ae.decode(q)
    idx = torch.randperm(dec.numel(), generator=g)[:4096]
    save("ae_c1", quant_bits=np.packbits((q > 0).numpy().reshape(-1)), quant_shape=np.array(q.shape),
         sample_idx=idx, dec_samples=dec.reshape(-1)[idx], dec_mean=dec.mean(), dec_std=dec.std(),
         n_tensors=np.int64(len(shapes)), n_params=np.int64(sum(int(np.prod(v)) for v in shapes.values())))


def gen_ae_gan():
    """The GAN-decoder variant of the tokenizer (VQModel(gan_decoder=True), autoencoder.py:279-351): one decode of a +-1 token map on
    CPU (fp32) with the global generator seeded right before it -- the decoder draws its noise map with torch.randn_like(z)."""
    from modeling.vision_encoder.autoencoder import VQModel
    ae = VQModel(**tm.TINY_AE, gan_decoder=True).eval()
    shapes = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(tm.seeded_state(shapes, seed=47, gain=1.4))
    g = torch.Generator().manual_seed(5)
    q = torch.sign(torch.randn(2, 32, 4, 6, generator=g))
    torch.manual_seed(77)
    with torch.no_grad():
        dec = ae.decode(q)
    torch.manual_seed(77)
    noise = torch.randn_like(q)                                  # what the decoder drew (first draw after the seed)
    save("ae_gan", quant=q, dec=dec, noise=noise, seed=np.int64(77),
         keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))


def gen_misc():
    pipe = build_pipeline(torch.float32)
    save("posembed", table=pipe.pos_embed_1d, e_4_6_2=pipe.get_2d_embed(4, 6, ps=2),
         e_16_16_8=pipe.get_2d_embed(16, 16, ps=8))
    ae, shapes = build_ae()
    g = torch.Generator().manual_seed(303)
    img = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        q = ae.encode(img)
        dec = ae.decode(q)
        henc = ae.encoder(img)
    save("ae_roundtrip", image=img, henc=henc, quant=q, dec=dec,
         keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
    # GFQ bit/index math, exhaustive over one 8-bit codebook + random multi-codebook tokens
    sys.path.insert(0, os.path.join(rh.REF_ROOT, "imagenet_gen"))
    from src.gfq import GFQ
    q = GFQ(dim=32, num_codebooks=4)
    idx = torch.arange(256)
    bits = q.indices_to_bits(idx)
    back = q.bits_to_indices(bits)
    g = torch.Generator().manual_seed(404)
    z = torch.randn(2, 32, 3, 5, generator=g)
    z[0, :, 0, 0] = 0.0                                           # exact zeros quantise to -1 (x > 0 is false)
    z[1, 5, 1, 2] = -0.0
    q.eval()
    with torch.no_grad():
        quant, _, indices = q(z, return_loss=False)              # GFQ.forward :196-291, 4 codebooks x 8 bits
    save("gfq", idx=idx, bits=bits, back=back, codebook=q.codebook, fwd_z=z, fwd_quant=quant,
         fwd_indices=torch.stack(indices))


def gen_imagenet():
    """Class-conditional ImageNet BitDance (imagenet_gen/src/model_parallel.py), tiny dims, 4 AR steps x 3 sampling steps."""
    os.environ["TORCHDYNAMO_DISABLE"] = "1"                      # the reference decorates with @torch.compile
    sys.path.insert(0, os.path.join(rh.REF_ROOT, "imagenet_gen"))
    from src.model_parallel import BitDance
    c = tm.TINY_IN
    shapes = tm.imagenet_shapes(c)
    for tag in ("fp32", "amp"):
        torch.manual_seed(0)
        m = BitDance(dim=c["dim"], n_layer=c["n_layer"], n_head=c["n_head"], diff_layers=c["diff_layers"],
                     diff_dim=c["diff_dim"], diff_adanln_layers=c["diff_adanln_layers"], latent_dim=c["latent_dim"],
                     down_size=c["down_size"], patch_size=c["patch_size"], resolution=c["resolution"], diff_batch_mul=1,
                     cls_token_num=c["cls_token_num"], num_classes=c["num_classes"], parallel_num=c["parallel_num"],
                     parallel_mode="patch", time_shift=c["time_shift"]).eval()
        sd = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("vae.")}
        assert sd == {k: tuple(v) for k, v in shapes.items()}, set(sd) ^ set(shapes)
        m.load_state_dict(tm.seeded_state(shapes, seed=29), strict=False)
        m.vae.decode = lambda x: x                               # stop at the latent: the AE is pinned by ae_roundtrip
        conds, preds = [], []
        orig = m.head.sample

        def rec(z, cfg, num_sampling_steps):
            conds.append(z.detach().float().clone())
            o = orig(z, cfg=cfg, num_sampling_steps=num_sampling_steps)
            preds.append(o.detach().clone())
            return o

        m.head.sample = rec
        ids = torch.tensor([3, 7])
        ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
        with torch.no_grad(), ctx, rh.ReplayNoise(seed=17) as rn:
            lat = m.sample(ids, sample_steps=3, cfg_scale=3.0, cfg_schedule="linear")
        n0 = [t for t in rn.record if t.shape[0] == 4]
        n1 = [t for t in rn.record if t.shape[0] == 2]
        save(f"imagenet_{tag}", ids=ids, latent=lat, preds=torch.cat(preds, dim=1), conds=torch.cat(conds, dim=1),
             noise0=torch.stack(n0), noise1=torch.stack(n1), calls=rn.calls, cfg=np.float32(3.0), n_steps=3,
             rope=m.freqs_cis, mask=m.attn_mask[0, 0])
        # the other two head_sample branches (model_parallel.py:356-365): constant CFG (mixed from the first step on)
        # and no CFG (cfg_scale <= 1: a single branch, no null-class rows)
        for name, kw in (("const", dict(cfg_scale=2.0, cfg_schedule="constant")), ("nocfg", dict(cfg_scale=1.0))):
            conds.clear(); preds.clear()
            ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
            with torch.no_grad(), ctx, rh.ReplayNoise(seed=23) as rn:
                lat = m.sample(ids, sample_steps=2, **kw)
            save(f"imagenet_{name}_{tag}", ids=ids, latent=lat, preds=torch.cat(preds, dim=1), noise=torch.stack(rn.record),
                 calls=rn.calls, cfg=np.float32(kw["cfg_scale"]), n_steps=2)


def gen_imagenet_variants():
    """The other released ImageNet variants at tiny dims: 1x (imagenet_gen/src/model.py: one token per AR step, causal
    transformer, MLP head, 16 AR steps) and 4x (src/model_parallel.py with parallel_num 4, 4 AR steps); 2 sampling steps,
    linear CFG ramp."""
    os.environ["TORCHDYNAMO_DISABLE"] = "1"
    sys.path.insert(0, os.path.join(rh.REF_ROOT, "imagenet_gen"))
    from src.model import BitDance as BitDance1x
    from src.model_parallel import BitDance as BitDancePar
    for name, c in (("1x", tm.TINY_IN_1X), ("4x", tm.TINY_IN_4X)):
        shapes = tm.imagenet_shapes(c)
        for tag in ("fp32", "amp"):
            torch.manual_seed(0)
            kw = dict(dim=c["dim"], n_layer=c["n_layer"], n_head=c["n_head"], diff_layers=c["diff_layers"], diff_dim=c["diff_dim"],
                      diff_adanln_layers=c["diff_adanln_layers"], latent_dim=c["latent_dim"], down_size=c["down_size"],
                      patch_size=c["patch_size"], resolution=c["resolution"], diff_batch_mul=1, cls_token_num=c["cls_token_num"],
                      num_classes=c["num_classes"], time_shift=c["time_shift"])
            m = (BitDance1x(**kw) if name == "1x" else BitDancePar(parallel_num=4, parallel_mode="patch", **kw)).eval()
            sd = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("vae.")}
            assert sd == {k: tuple(v) for k, v in shapes.items()}, set(sd) ^ set(shapes)
            m.load_state_dict(tm.seeded_state(shapes, seed=31), strict=False)
            m.vae.decode = lambda x: x
            conds, preds = [], []
            orig = m.head.sample

            def rec(z, cfg, num_sampling_steps):
                conds.append(z.detach().float().clone())
                o = orig(z, cfg=cfg, num_sampling_steps=num_sampling_steps)
                preds.append(o.detach().clone())
                return o

            m.head.sample = rec
            ids = torch.tensor([3, 7])
            ctx = rh.CudaAutocastOnCpu() if tag == "amp" else torch.no_grad()
            with torch.no_grad(), ctx, rh.ReplayNoise(seed=19) as rn:
                lat = m.sample(ids, sample_steps=2, cfg_scale=3.0, cfg_schedule="linear")
            P = c["parallel_num"]
            C = c["latent_dim"]
            pr = torch.cat([p.reshape(p.shape[0], P, C) for p in preds], dim=1)          # [bsz, h*w, C]
            n0 = [t.reshape(4, P, C) for t in rn.record if t.shape[0] == 4]
            n1 = [t.reshape(2, P, C) for t in rn.record if t.shape[0] == 2]
            save(f"imagenet{name}_{tag}", ids=ids, latent=lat, preds=pr, noise0=torch.stack(n0), noise1=torch.stack(n1),
                 calls=rn.calls, cfg=np.float32(3.0), n_steps=2, rope=m.freqs_cis)


def main():
    rh.install()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "imagenet":
        return gen_imagenet()
    if len(sys.argv) > 1 and sys.argv[1] == "imagenet_variants":
        return gen_imagenet_variants()
    if len(sys.argv) > 1 and sys.argv[1] == "text_sampling":
        return gen_text_sampling()
    if len(sys.argv) > 1 and sys.argv[1] == "interleaved":
        return gen_interleaved()
    if len(sys.argv) > 1 and sys.argv[1] == "pipeline":
        return gen_pipeline()
    if len(sys.argv) > 1 and sys.argv[1] == "mllm":
        return gen_mllm_equiv()
    if len(sys.argv) > 1 and sys.argv[1] == "full_causal":
        return gen_full_causal()
    if len(sys.argv) > 1 and sys.argv[1] == "ae_c1":
        return gen_ae_c1()
    if len(sys.argv) > 1 and sys.argv[1] == "misc":
        return gen_misc()
    if len(sys.argv) > 1 and sys.argv[1] == "ae_gan":
        return gen_ae_gan()
    gen_sampler()
    gen_head()
    gen_llm()
    gen_pipeline()
    gen_misc()
    gen_imagenet()
    gen_imagenet_variants()
    gen_mllm_equiv()
    gen_full_causal()
    gen_interleaved()
    gen_text_sampling()
    gen_ae_c1()
    gen_ae_gan()


if __name__ == "__main__":
    main()
