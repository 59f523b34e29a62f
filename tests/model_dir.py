"""A released-layout BitDance model directory written from the seeded tiny models (test helper).

Layout read by the reference's ``BitDanceT2IPipeline.__init__`` (modeling/t2i_pipeline.py:45-75): HF tokenizer files,
``config.json`` (Qwen3), ``model.safetensors`` or the sharded ``model-0000x-of-0000y.safetensors`` +
``model.safetensors.index.json``, ``ae_config.json`` / ``ae.safetensors``, ``vision_head_config.json`` /
``vision_head.safetensors``, ``projector.safetensors``.  The tokenizer is a stub WordLevel vocabulary (no tokenizer files
exist offline) that carries the special tokens the pipeline looks up: ``<|vision_start|>``, ``<|res_N|>``, ``<|query_i|>``."""
import json
import os

import torch

from oracle import tiny_models as tm


def write_tokenizer(d: str) -> None:
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {}
    for i in range(256):
        vocab[f"<0x{i:02X}>" if i < 33 or i > 126 else chr(i)] = len(vocab)
    special = ["<|im_start|>", "<|im_end|>", "<|vision_start|>"] + [f"<|res_{i}|>" for i in range(1, 65)] + \
              [f"<|query_{i}|>" for i in range(1, 64)]
    for t in special:
        vocab[t] = len(vocab)
    vocab["<unk>"] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", "isolated")
    tok.add_special_tokens(special)
    tok.save(os.path.join(d, "tokenizer.json"))
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "unk_token": "<unk>"}, f)


def write_model_dir(d: str, sharded: bool = False, head_cfg: dict | None = None) -> dict:
    """Returns the state dicts that were written (for comparisons)."""
    from safetensors.torch import save_file
    from bitdance_amd.autoencoder import VQModel
    os.makedirs(d, exist_ok=True)
    write_tokenizer(d)
    head_cfg = dict(head_cfg or tm.TINY_HEAD)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"model_type": "qwen3", "architectures": ["Qwen3ForCausalLM"], **tm.TINY_LLM}, f)
    with open(os.path.join(d, "ae_config.json"), "w") as f:
        json.dump(tm.TINY_AE, f)
    with open(os.path.join(d, "vision_head_config.json"), "w") as f:
        json.dump(head_cfg, f)
    llm = {k: v.to(torch.bfloat16).contiguous() for k, v in tm.seeded_state(tm.llm_shapes(tm.TINY_LLM), seed=22).items()}
    llm["lm_head.weight"] = llm["model.embed_tokens.weight"].clone()           # present in HF checkpoints, unused for T2I
    if sharded:
        keys = sorted(llm)
        half = len(keys) // 2
        parts = {"model-00001-of-00002.safetensors": keys[:half], "model-00002-of-00002.safetensors": keys[half:]}
        for fn, ks in parts.items():
            save_file({k: llm[k] for k in ks}, os.path.join(d, fn))
        with open(os.path.join(d, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": {k: fn for fn, ks in parts.items() for k in ks}}, f)
    else:
        save_file(llm, os.path.join(d, "model.safetensors"))
    ae_shapes = {k: tuple(v.shape) for k, v in VQModel(**tm.TINY_AE).state_dict().items()}
    ae = {k: v.contiguous() for k, v in tm.seeded_state(ae_shapes, seed=44, gain=1.4).items()}
    head = {k: v.contiguous() for k, v in tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11).items()}
    proj = {k: v.contiguous() for k, v in tm.seeded_state(tm.proj_shapes(32, 256), seed=33).items()}
    save_file(ae, os.path.join(d, "ae.safetensors"))
    save_file(head, os.path.join(d, "vision_head.safetensors"))
    save_file(proj, os.path.join(d, "projector.safetensors"))
    return dict(llm=llm, ae=ae, head=head, proj=proj)
