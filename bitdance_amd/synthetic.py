"""Random-weight models at true (or tiny) shapes, generated directly on the device.

There is no network in the build/bench environment, hence no real checkpoints: benchmarks and smoke tests use
random weights of the released architectures (SURVEY.md section 8d "Concrete synthetic inputs").  The head's
zero-initialised tensors (adaLN, output layer: flow_head_parallel_x.py:315-323) are drawn N(0, 0.02) like the rest,
otherwise a random-weight head would be degenerate.
"""
from __future__ import annotations

import math

import torch

BF16 = torch.bfloat16

QWEN3_14B = dict(hidden_size=5120, num_hidden_layers=40, num_attention_heads=40, num_key_value_heads=8, head_dim=128,
                 intermediate_size=17408, vocab_size=151936 + 320, rms_norm_eps=1e-6, rope_theta=1000000.0)
HEAD_14B_64X = dict(ch_target=32, ch_cond=5120, ch_latent=5120, depth_latent=6, depth_adanln=2, parallel_num=64,
                    use_swiglu=True, time_shift=1.0, time_schedule="logit_normal", P_mean=-0.8, P_std=0.8,
                    diff_batch_mul=1)                      # train/configs/bitdance_14b_64x.yaml:22-33
HEAD_14B_16X = dict(HEAD_14B_64X, parallel_num=16)          # BitDance-14B-16x: 16-token patches (README.md:77-78)
# imagenet_gen BitDance-B-16x (model_parallel.py:456-465): 24 layers, width 768, 12 heads of 64, head 6 blocks / 2 adaLN
IMAGENET_B_16X = dict(dim=768, n_layer=24, n_head=12, diff_layers=6, diff_dim=768, diff_adanln_layers=2, latent_dim=32,
                      down_size=16, patch_size=1, resolution=256, cls_token_num=64, num_classes=1000, parallel_num=16,
                      time_shift=1.0)
# every released ImageNet checkpoint (imagenet_gen/README.md:10-15; model.py:394-430 / model_parallel.py:437-473)
IMAGENET_MODELS = {
    "b16x": IMAGENET_B_16X,
    "b4x": dict(IMAGENET_B_16X, parallel_num=4),
    "b1x": dict(IMAGENET_B_16X, parallel_num=1),
    "l1x": dict(IMAGENET_B_16X, parallel_num=1, dim=1024, n_layer=32, n_head=16, diff_layers=8, diff_dim=1024),
    "h1x": dict(IMAGENET_B_16X, parallel_num=1, dim=1280, n_layer=40, n_head=20, diff_layers=12, diff_dim=1280,
                diff_adanln_layers=3),
}
AE_D16C32 = dict(ddconfig=dict(double_z=False, z_channels=32, in_channels=3, out_ch=3, ch=256, ch_mult=[1, 1, 2, 2, 4],
                               num_res_blocks=4), gan_decoder=False)      # bitdance_14b_64x.yaml:9-16
# ae_d32c256 (README.md:69: 2^256 codebook, 32x down-sampling): z 256, patch 32.  Its config json is not in the reference tree
# (hosted next to the weights); the shape below continues ae_d16c32's ladder one level (ch_mult [1,1,2,2,4,4]) -- an ASSUMPTION,
# used only by the standalone decoder benchmark (bench.py --workload ae-d32c256-decode).
AE_D32C256 = dict(ddconfig=dict(double_z=False, z_channels=256, in_channels=3, out_ch=3, ch=256, ch_mult=[1, 1, 2, 2, 4, 4],
                                num_res_blocks=4), gan_decoder=False)

TINY_LLM = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                intermediate_size=512, vocab_size=512, rms_norm_eps=1e-6, rope_theta=1000000.0)
TINY_HEAD = dict(ch_target=32, ch_cond=256, ch_latent=256, depth_latent=4, depth_adanln=2, parallel_num=64,
                 use_swiglu=True, time_shift=1.0)
TINY_AE = dict(ddconfig=dict(double_z=False, z_channels=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 1, 2, 2, 4],
                             num_res_blocks=1))


def _normal(shape, std, gen, device, dtype=BF16):
    return (torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)


def random_llm_state(cfg: dict, device, seed: int = 0, std: float = 0.02) -> dict:
    g = torch.Generator(device=device).manual_seed(seed)
    D, nh, nkv, hd, ff = (cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"],
                          cfg["head_dim"], cfg["intermediate_size"])
    sd = {"model.embed_tokens.weight": _normal((cfg["vocab_size"], D), std, g, device),
          "model.norm.weight": torch.ones(D, dtype=BF16, device=device)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = _normal((nh * hd, D), std, g, device)
        sd[p + "self_attn.k_proj.weight"] = _normal((nkv * hd, D), std, g, device)
        sd[p + "self_attn.v_proj.weight"] = _normal((nkv * hd, D), std, g, device)
        sd[p + "self_attn.o_proj.weight"] = _normal((D, nh * hd), std, g, device)
        sd[p + "mlp.gate_proj.weight"] = _normal((ff, D), std, g, device)
        sd[p + "mlp.up_proj.weight"] = _normal((ff, D), std, g, device)
        sd[p + "mlp.down_proj.weight"] = _normal((D, ff), std, g, device)
        for n in ("self_attn.q_norm", "self_attn.k_norm"):
            sd[p + n + ".weight"] = torch.ones(hd, dtype=BF16, device=device)
        for n in ("input_layernorm", "post_attention_layernorm"):
            sd[p + n + ".weight"] = torch.ones(D, dtype=BF16, device=device)
    return sd


def random_head_state(cfg: dict, device, seed: int = 1, std: float = 0.02, mlp: bool = False) -> dict:
    """``mlp``: the MLP head of the 1x ImageNet models (imagenet_gen/src/diff_head.py:165-225) instead of the transformer head."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, C, Z = cfg["ch_latent"], cfg["ch_target"], cfg["ch_cond"]
    H = int(D * 1.5)
    sd = {}

    def lin(name, n, k):
        sd[name + ".weight"] = _normal((n, k), std, g, device)
        sd[name + ".bias"] = _normal((n,), std, g, device)

    lin("net.time_embed.mlp.0", D, 256)
    lin("net.time_embed.mlp.2", D, D)
    lin("net.cond_embed", D, Z)
    lin("net.input_proj", D, C)
    for i in range(cfg["depth_latent"]):
        p = f"net.res_blocks.{i}."
        for n in (("norm",) if mlp else ("norm1", "norm2")):
            sd[p + n + ".weight"] = torch.ones(D, dtype=torch.float32, device=device)
            sd[p + n + ".bias"] = torch.zeros(D, dtype=torch.float32, device=device)
        if not mlp:
            lin(p + "attn.wqkv", 3 * D, D)
            lin(p + "attn.wo", D, D)
        lin(p + "w1", 2 * H, D)
        lin(p + "w2", D, H)
    for j in range(cfg["depth_adanln"]):
        lin(f"net.ada_ln_blocks.{j}", (3 if mlp else 6) * D, D)
    lin("net.final_layer.ada_ln_modulation", 2 * D, D)
    lin("net.final_layer.linear", C, D)
    return sd


def random_proj_state(c: int, d: int, device, seed: int = 2, std: float = 0.02) -> dict:
    g = torch.Generator(device=device).manual_seed(seed)
    return {"fc1.weight": _normal((d, c), 1.0 / math.sqrt(c), g, device), "fc1.bias": _normal((d,), std, g, device),
            "fc2.weight": _normal((d, d), std, g, device), "fc2.bias": _normal((d,), std, g, device)}


def random_ae_state(ae_config: dict, device, seed: int = 3) -> dict:
    from .autoencoder import VQModel
    with torch.device("meta"):
        m = VQModel(**ae_config)
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, v in m.state_dict().items():
        if v.dim() >= 2:
            fan_in = math.prod(v.shape[1:])
            sd[k] = _normal(tuple(v.shape), 1.0 / math.sqrt(fan_in), g, device, torch.float32)
        elif k.endswith("bias"):
            sd[k] = torch.zeros(v.shape, device=device)
        else:
            sd[k] = torch.ones(v.shape, device=device)
    return sd


def random_imagenet_state(cfg: dict, device, seed: int = 4, std: float = 0.02) -> dict:
    """state_dict() of imagenet_gen's BitDance minus ``vae.*`` (model_parallel.py:104-196) with random values: transformer
    (``layers.{i}.attention.wqkv / wo``, ``feed_forward.w1 / w2`` with find_multiple(2*4*dim/3, 256) hidden features,
    RMSNorm scales), ``proj_in`` SwiGLU connector, class / query / position embeddings and the ``head.net.*`` diffusion head."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, L = cfg["dim"], cfg["latent_dim"] * cfg["patch_size"] ** 2
    hid = int(D * 1.5)
    ff = int(2 * 4.0 * D / 3)
    ff = ff if ff % 256 == 0 else ff + 256 - ff % 256
    hw = cfg["resolution"] // (cfg["down_size"] * cfg["patch_size"])
    f32 = torch.float32
    sd = {"query_token": _normal((1, cfg["parallel_num"] - 1, D), std, g, device, f32),     # absent from the 1x models (dropped below)
          "cls_embedding.weight": _normal((cfg["num_classes"] + 1, D * cfg["cls_token_num"]), std, g, device, f32),
          "proj_in.w1.weight": _normal((2 * hid, L), 1.0 / math.sqrt(L), g, device, f32),
          "proj_in.w1.bias": _normal((2 * hid,), std, g, device, f32),
          "proj_in.w2.weight": _normal((D, hid), 1.0 / math.sqrt(hid), g, device, f32),
          "proj_in.w2.bias": _normal((D,), std, g, device, f32),
          "emb_norm.weight": torch.ones(D, device=device), "norm.weight": torch.ones(D, device=device),
          "pos_for_diff.weight": _normal((hw * hw, D), std, g, device, f32)}
    for i in range(cfg["n_layer"]):
        p = f"layers.{i}."
        sd[p + "attention.wqkv.weight"] = _normal((3 * D, D), 1.0 / math.sqrt(D), g, device, f32)
        sd[p + "attention.wo.weight"] = _normal((D, D), 1.0 / math.sqrt(D), g, device, f32)
        sd[p + "feed_forward.w1.weight"] = _normal((2 * ff, D), 1.0 / math.sqrt(D), g, device, f32)
        sd[p + "feed_forward.w2.weight"] = _normal((D, ff), 1.0 / math.sqrt(ff), g, device, f32)
        sd[p + "attention_norm.weight"] = torch.ones(D, device=device)
        sd[p + "ffn_norm.weight"] = torch.ones(D, device=device)
    hcfg = dict(ch_target=L, ch_cond=D, ch_latent=cfg["diff_dim"], depth_latent=cfg["diff_layers"],
                depth_adanln=cfg["diff_adanln_layers"])
    one_x = cfg["parallel_num"] == 1                     # imagenet_gen/src/model.py: no query tokens, MLP head
    if one_x:
        del sd["query_token"]
    for k, v in random_head_state(hcfg, device, seed=seed + 1, std=1.0 / math.sqrt(cfg["diff_dim"]), mlp=one_x).items():
        sd["head." + k] = v.float()
    return sd


def build_imagenet(device: str = "cuda", cfg: dict | None = None, with_vae: bool = True):
    """A class-conditional ImageNet model (default BitDance-B-16x, 256 px; ``cfg`` = an IMAGENET_MODELS entry) on random
    weights, optionally with the ae_d16c32 decoder."""
    from .autoencoder import VQModel
    from .imagenet import BitDance
    cfg = dict(cfg or IMAGENET_B_16X)
    vae = None
    if with_vae:
        vae = VQModel(**AE_D16C32).eval()
        vae.load_state_dict(random_ae_state(AE_D16C32, device), strict=True, assign=True)
        vae.to(device)
    return BitDance(random_imagenet_state(cfg, device), device=device, vae=vae, **cfg)


class SyntheticTokenizer:
    """Fixed-length stand-in for the HF tokenizer: no tokenizer files exist offline.  ``encode`` maps a prompt to
    a deterministic id list (77 ids for a user prompt, 3 for the bare assistant prefix)."""

    def __init__(self, vocab_size: int):
        self.special0 = vocab_size - 320

    def encode(self, text: str):
        n = 3 if len(text) < 32 else 77
        base = sum(map(ord, text))
        return [(base + 13 * i) % self.special0 for i in range(n)]

    def convert_tokens_to_ids(self, tok: str) -> int:
        if tok == "<|vision_start|>":
            return self.special0
        if tok.startswith("<|res_"):
            return self.special0 + 1 + int(tok[6:-2])
        if tok.startswith("<|query_"):
            return self.special0 + 130 + int(tok[8:-2])
        raise KeyError(tok)


def build_pipeline(size: str = "14b-64x", device: str = "cuda", with_ae: bool = True, tp=None, weights: str = "bf16"):
    """A BitDanceT2IPipeline on random weights: ``14b-64x`` / ``14b-16x`` (BitDance-14B shapes) or ``tiny``.  ``tp``: a
    tp.TPComm -- every rank draws the SAME full model from the same seeds and keeps its slices."""
    from .t2i_pipeline import BitDanceT2IPipeline
    if size == "14b-64x":
        lc, hc, ac = QWEN3_14B, HEAD_14B_64X, AE_D16C32
    elif size == "14b-16x":
        lc, hc, ac = QWEN3_14B, HEAD_14B_16X, AE_D16C32
    elif size == "tiny":
        lc, hc, ac = TINY_LLM, TINY_HEAD, TINY_AE
    else:
        raise ValueError(size)
    head_cfg = dict(hc)
    pipe = BitDanceT2IPipeline.from_components(
        tokenizer=SyntheticTokenizer(lc["vocab_size"]), llm_cfg=lc, llm_sd=random_llm_state(lc, device),
        ae_config=ac, ae_sd=random_ae_state(ac, device) if with_ae else None, head_config=head_cfg,
        head_sd=random_head_state(hc, device), proj_sd=random_proj_state(hc["ch_target"], lc["hidden_size"], device),
        device=device, tp=tp, weights=weights)
    return pipe
