"""Seeded tiny model definitions shared by oracle/gen_golden.py and tests (test infrastructure).

Shapes/key names follow the reference checkpoints; gen_golden.py
asserts them against the state_dicts of the instantiated reference modules.
All values are bf16-representable so fp32 and bf16 runs see identical weights.
"""
from __future__ import annotations

import math

import torch

TINY_LLM = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                head_dim=128, intermediate_size=512, vocab_size=512, rms_norm_eps=1e-6,
                rope_theta=1000000.0)
TINY_HEAD = dict(ch_target=32, ch_cond=256, ch_latent=256, depth_latent=4, depth_adanln=2,
                 parallel_num=64, use_swiglu=True, time_shift=1.0)
TINY_HEAD16 = dict(TINY_HEAD, parallel_num=16)      # the 16x models: 16-token patches, <=32-token attention branch
TINY_AE = dict(ddconfig=dict(double_z=False, z_channels=32, in_channels=3, out_ch=3, ch=32,
                             ch_mult=[1, 1, 2, 2, 4], num_res_blocks=1))

# class-conditional ImageNet model (imagenet_gen/src/model_parallel.py BitDance.__init__), 128 px -> 8x8 tokens, 4 AR steps
TINY_IN = dict(dim=256, n_layer=2, n_head=4, diff_layers=4, diff_dim=256, diff_adanln_layers=2, latent_dim=16,
               down_size=16, patch_size=1, resolution=128, cls_token_num=8, num_classes=10, parallel_num=16,
               time_shift=1.0)

# the other released ImageNet variants (imagenet_gen/README.md:10-15): 1x = src/model.py (one token per AR step, causal
# transformer, MLP diffusion head src/diff_head.py), 4x = src/model_parallel.py with parallel_num 4
TINY_IN_1X = dict(TINY_IN, resolution=64, parallel_num=1)
TINY_IN_4X = dict(TINY_IN, resolution=64, parallel_num=4)

# BASELINE config 1: the released ae_d16c32 tokenizer at full size (bitdance_14b_64x.yaml:9-16), 256x256 round trip on CPU
AE_D16C32 = dict(ddconfig=dict(double_z=False, z_channels=32, in_channels=3, out_ch=3, ch=256, ch_mult=[1, 1, 2, 2, 4],
                               num_res_blocks=4), gan_decoder=False)

VISION_START, RES_BASE, QUERY_BASE = 300, 301, 430
VISION_END, IM_START, IM_END = 495, 496, 497          # <|vision_end|>, <|im_start|>, <|im_end|> (data/data_utils.py:95-109)


def llm_shapes(cfg: dict) -> dict:
    D, nh, nkv, hd, ff = (cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"],
                          cfg["head_dim"], cfg["intermediate_size"])
    s = {"model.embed_tokens.weight": (cfg["vocab_size"], D), "model.norm.weight": (D,)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (nh * hd, D)
        s[p + "self_attn.k_proj.weight"] = (nkv * hd, D)
        s[p + "self_attn.v_proj.weight"] = (nkv * hd, D)
        s[p + "self_attn.o_proj.weight"] = (D, nh * hd)
        s[p + "self_attn.q_norm.weight"] = (hd,)
        s[p + "self_attn.k_norm.weight"] = (hd,)
        s[p + "mlp.gate_proj.weight"] = (ff, D)
        s[p + "mlp.up_proj.weight"] = (ff, D)
        s[p + "mlp.down_proj.weight"] = (D, ff)
        s[p + "input_layernorm.weight"] = (D,)
        s[p + "post_attention_layernorm.weight"] = (D,)
    return s


def head_shapes(cfg: dict) -> dict:
    D, C, Z = cfg["ch_latent"], cfg["ch_target"], cfg["ch_cond"]
    H = int(D * 1.5)
    s = {}

    def lin(name, n, k):
        s[name + ".weight"] = (n, k)
        s[name + ".bias"] = (n,)

    lin("net.time_embed.mlp.0", D, 256)
    lin("net.time_embed.mlp.2", D, D)
    lin("net.cond_embed", D, Z)
    lin("net.input_proj", D, C)
    for i in range(cfg["depth_latent"]):
        p = f"net.res_blocks.{i}."
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (D,)
            s[p + n + ".bias"] = (D,)
        lin(p + "attn.wqkv", 3 * D, D)
        lin(p + "attn.wo", D, D)
        lin(p + "w1", 2 * H, D)
        lin(p + "w2", D, H)
    for j in range(cfg["depth_adanln"]):
        lin(f"net.ada_ln_blocks.{j}", 6 * D, D)
    lin("net.final_layer.ada_ln_modulation", 2 * D, D)
    lin("net.final_layer.linear", C, D)
    return s


def mlp_head_shapes(cfg: dict) -> dict:
    """state_dict() of imagenet_gen/src/diff_head.py DiffHead (MlpEncoder :165-225): ResBlock = LayerNorm + SwiGLU MLP of
    width 1.5 x channels (:126-137), adaLN blocks of 3 chunks (:199), final layer as in the transformer head."""
    D, C, Z = cfg["ch_latent"], cfg["ch_target"], cfg["ch_cond"]
    H = int(D * 1.5)
    s = {}

    def lin(name, n, k):
        s[name + ".weight"] = (n, k)
        s[name + ".bias"] = (n,)

    lin("net.time_embed.mlp.0", D, 256)
    lin("net.time_embed.mlp.2", D, D)
    lin("net.cond_embed", D, Z)
    lin("net.input_proj", D, C)
    for i in range(cfg["depth_latent"]):
        p = f"net.res_blocks.{i}."
        s[p + "norm.weight"] = (D,)
        s[p + "norm.bias"] = (D,)
        lin(p + "w1", 2 * H, D)
        lin(p + "w2", D, H)
    for j in range(cfg["depth_adanln"]):
        lin(f"net.ada_ln_blocks.{j}", 3 * D, D)
    lin("net.final_layer.ada_ln_modulation", 2 * D, D)
    lin("net.final_layer.linear", C, D)
    return s


def imagenet_shapes(cfg: dict) -> dict:
    """state_dict() of imagenet_gen BitDance minus ``vae.*`` (names/shapes asserted against the reference module)."""
    D, L = cfg["dim"], cfg["latent_dim"] * cfg["patch_size"] ** 2
    hid = int(D * 1.5)
    ff = int(2 * 4.0 * D / 3)
    ff = ff if ff % 256 == 0 else ff + 256 - ff % 256                    # find_multiple(.., 256)
    hw = cfg["resolution"] // (cfg["down_size"] * cfg["patch_size"])
    s = {"cls_embedding.weight": (cfg["num_classes"] + 1, D * cfg["cls_token_num"]),
         "proj_in.w1.weight": (2 * hid, L), "proj_in.w1.bias": (2 * hid,),
         "proj_in.w2.weight": (D, hid), "proj_in.w2.bias": (D,),
         "emb_norm.weight": (D,), "norm.weight": (D,), "pos_for_diff.weight": (hw * hw, D)}
    for i in range(cfg["n_layer"]):
        p = f"layers.{i}."
        s[p + "attention.wqkv.weight"] = (3 * D, D)
        s[p + "attention.wo.weight"] = (D, D)
        s[p + "feed_forward.w1.weight"] = (2 * ff, D)
        s[p + "feed_forward.w2.weight"] = (D, ff)
        s[p + "attention_norm.weight"] = (D,)
        s[p + "ffn_norm.weight"] = (D,)
    if cfg["parallel_num"] > 1:                       # src/model_parallel.py; src/model.py (1x) has no query tokens
        s["query_token"] = (1, cfg["parallel_num"] - 1, D)
    hcfg = dict(ch_target=L, ch_cond=D, ch_latent=cfg["diff_dim"], depth_latent=cfg["diff_layers"],
                depth_adanln=cfg["diff_adanln_layers"])
    for k, v in (mlp_head_shapes(hcfg) if cfg["parallel_num"] == 1 else head_shapes(hcfg)).items():
        s["head." + k] = v
    return s


def proj_shapes(c: int, d: int) -> dict:
    return {"fc1.weight": (d, c), "fc1.bias": (d,), "fc2.weight": (d, d), "fc2.bias": (d,)}


def seeded_state(shapes: dict, seed: int, gain: float = 1.0) -> dict:
    """Deterministic weights: matrices ~ N(0, gain/sqrt(fan_in)), norm scales ~ 1+0.1N, biases ~ 0.1N.
    Iterates names in sorted order from one CPU generator; values rounded to bf16."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        x = torch.empty(shp, dtype=torch.float32).normal_(generator=g)
        if len(shp) >= 2:
            fan_in = math.prod(shp[1:])
            x = x * (gain / math.sqrt(fan_in))
        elif name.endswith("bias"):
            x = x * 0.1
        else:                               # 1-D scale of a norm layer
            x = 1.0 + 0.1 * x
        out[name] = x.to(torch.bfloat16).to(torch.float32)
    return out


class FakeTokenizer:
    """Stands in for the HF tokenizer (t2i_pipeline.py:175-194): deterministic char -> id map."""

    def encode(self, text: str):
        return [ord(ch) % 256 for ch in text]

    def convert_tokens_to_ids(self, tok: str) -> int:
        if tok == "<|vision_start|>":
            return VISION_START
        if tok in ("<|vision_end|>", "<|im_start|>", "<|im_end|>"):
            return {"<|vision_end|>": VISION_END, "<|im_start|>": IM_START, "<|im_end|>": IM_END}[tok]
        if tok.startswith("<|res_"):
            return RES_BASE + int(tok[6:-2])
        if tok.startswith("<|query_"):
            return QUERY_BASE + int(tok[8:-2])
        raise KeyError(tok)
