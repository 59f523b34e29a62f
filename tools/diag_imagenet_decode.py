"""Native vs torch transformer decode step of the ImageNet models at real dimensions: relative error of norm(x) for the
first decode steps (diagnostic for tests/test_gpu_parity.py::test_imagenet_released_variants_real_dims_run).
python tools/diag_imagenet_decode.py [b16x b4x b1x]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitdance_amd import synthetic as syn                   # noqa: E402
from bitdance_amd.imagenet import BitDance                  # noqa: E402


def main():
    dev = "cuda"
    for v in (sys.argv[1:] or ["b16x", "b4x", "b1x"]):
        c = dict(syn.IMAGENET_MODELS[v])
        m = BitDance(syn.random_imagenet_state(c, dev), device=dev, **c)
        P, n_cls, bsz = m.P, m.cls_token_num, 8
        ids = torch.arange(bsz, device=dev)
        hd = m.dim // m.n_head
        caches = [(torch.zeros(bsz, m.n_head, m.total_tokens, hd, device=dev), torch.zeros(bsz, m.n_head, m.total_tokens, hd, device=dev))
                  for _ in range(m.n_layer)]
        w = m.w_
        T0 = n_cls + P - 1
        with torch.autocast("cuda", dtype=torch.bfloat16):
            cemb = F.embedding(ids, w["cls_embedding.weight"]).view(bsz, n_cls, -1)
            x = torch.cat([cemb, w["query_token"].repeat(bsz, 1, 1)], dim=1) if P > 1 else cemb
            m._forward_model(x, m.attn_mask[:, :, :T0, :T0], 0, T0, caches)
        eng = m._tr_engine(bsz)
        m._load_cache(eng, caches, T0)
        g = torch.Generator(device=dev).manual_seed(3)
        for i in range(1, 4):
            tok = torch.sign(torch.randn(bsz, P, c["latent_dim"], device=dev, generator=g))
            s0 = P * (i - 1) + n_cls + P - 1
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ref = m._forward_model(m._proj_in(tok), m.attn_mask[:, :, s0:s0 + P, :s0 + P], s0, s0 + P, caches).float()
            got = m._decode_step(eng, tok).float()
            d = (got - ref).abs()
            print(f"{v} step {i}: |ref| mean {ref.abs().mean():.4f}  err mean {d.mean():.5f} max {d.max():.4f}  rel {d.mean() / ref.abs().mean():.4f}", flush=True)
        del m, eng, caches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
