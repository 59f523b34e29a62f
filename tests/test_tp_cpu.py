"""Tensor-parallel sharding math on CPU (no GPU): the slices tp.shard_* hand to each rank, multiplied rank by rank and
summed, reproduce the unsharded operator -- including the q/k/v thirds of ``wqkv`` (split by attention head), the
(h1, h2) SwiGLU pairs of ``w1``, GQA head <-> kv-head co-location in the LLM, and biases applied exactly once.
Plus a world-2 gloo run of the same reduction through ``torch.distributed`` (what the N > 1 path's collectives compute)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bitdance_amd.tp import shard_head_state, shard_llm_state          # noqa: E402
from oracle import tiny_models as tm                                   # noqa: E402

F64 = torch.float64
HEAD = dict(ch_target=32, ch_cond=256, ch_latent=1024, depth_latent=1, depth_adanln=1)        # 8 heads of 128, H = 1536
LLM = dict(hidden_size=512, num_hidden_layers=1, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
           intermediate_size=1024, vocab_size=16, rms_norm_eps=1e-6, rope_theta=1e6)


def _attn(q, k, v):                                  # [B,T,h,d] non-causal
    s = torch.einsum("bthd,bshd->bhts", q, k) / q.shape[-1] ** 0.5
    return torch.einsum("bhts,bshd->bthd", s.softmax(-1), v)


def head_block_partials(sd, x, n_head_local):
    """attention branch and SwiGLU branch of one TransBlock (flow_head_parallel_x.py:192-220,242-252) WITHOUT the output
    biases: what one rank contributes to the all-reduce."""
    p = "net.res_blocks.0."
    B, T, D = x.shape
    qkv = x @ sd[p + "attn.wqkv.weight"].T + sd[p + "attn.wqkv.bias"]
    q, k, v = qkv.chunk(3, dim=-1)
    shp = (B, T, n_head_local, 128)
    o = _attn(q.reshape(shp), k.reshape(shp), v.reshape(shp)).reshape(B, T, -1)
    a = o @ sd[p + "attn.wo.weight"].T
    h1, h2 = (x @ sd[p + "w1.weight"].T + sd[p + "w1.bias"]).chunk(2, dim=-1)
    m = (torch.nn.functional.silu(h1) * h2) @ sd[p + "w2.weight"].T
    return a, m


@pytest.mark.parametrize("tp", [2, 4, 8])
def test_head_shards_sum_to_unsharded(tp):
    sd = {k: v.to(F64) for k, v in tm.seeded_state(tm.head_shapes(HEAD), seed=3).items()}
    x = torch.randn(2, 16, 1024, dtype=F64, generator=torch.Generator().manual_seed(1))
    a_full, m_full = head_block_partials(sd, x, 8)
    a_sum, m_sum = 0, 0
    for r in range(tp):
        loc = shard_head_state(sd, r, tp)
        assert loc["net.res_blocks.0.attn.wqkv.weight"].shape == (3 * 1024 // tp, 1024)
        assert loc["net.res_blocks.0.w1.weight"].shape == (2 * 1536 // tp, 1024)
        assert loc["net.res_blocks.0.attn.wo.weight"].shape == (1024, 1024 // tp)
        assert loc["net.res_blocks.0.w2.weight"].shape == (1024, 1536 // tp)
        assert loc["net.res_blocks.0.attn.wo.bias"].shape == (1024,)            # row-split biases stay whole
        assert torch.equal(loc["net.ada_ln_blocks.0.weight"], sd["net.ada_ln_blocks.0.weight"])   # adaLN replicated
        a, m = head_block_partials(loc, x, 8 // tp)
        a_sum, m_sum = a_sum + a, m_sum + m
    torch.testing.assert_close(a_sum, a_full, rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(m_sum, m_full, rtol=1e-10, atol=1e-10)


def llm_layer_partials(w, cfg, x, nh, nkv):
    """o_proj / down_proj partial outputs of one Qwen3 layer's attention and MLP (HF modeling_qwen3.py:81-83,241-280;
    norms / RoPE omitted: they act per head and are replicated) for the heads / features in ``w``."""
    p = "model.layers.0."
    B, T, D = x.shape
    q = (x @ w[p + "self_attn.q_proj.weight"].T).reshape(B, T, nh, 128)
    k = (x @ w[p + "self_attn.k_proj.weight"].T).reshape(B, T, nkv, 128)
    v = (x @ w[p + "self_attn.v_proj.weight"].T).reshape(B, T, nkv, 128)
    rep = nh // nkv
    o = _attn(q, k.repeat_interleave(rep, dim=2), v.repeat_interleave(rep, dim=2)).reshape(B, T, nh * 128)
    a = o @ w[p + "self_attn.o_proj.weight"].T
    g, u = x @ w[p + "mlp.gate_proj.weight"].T, x @ w[p + "mlp.up_proj.weight"].T
    return a, (torch.nn.functional.silu(g) * u) @ w[p + "mlp.down_proj.weight"].T


@pytest.mark.parametrize("tp", [2, 4, 8])
def test_llm_shards_sum_to_unsharded(tp):
    w = {k: v.to(F64) for k, v in tm.seeded_state(tm.llm_shapes(LLM), seed=4).items()}
    x = torch.randn(1, 12, 512, dtype=F64, generator=torch.Generator().manual_seed(2))
    a_full, m_full = llm_layer_partials(w, LLM, x, 16, 8)
    a_sum, m_sum = 0, 0
    for r in range(tp):
        loc = shard_llm_state(w, LLM, r, tp)
        a, m = llm_layer_partials(loc, LLM, x, 16 // tp, 8 // tp)            # GQA: a rank's q heads use exactly its kv heads
        a_sum, m_sum = a_sum + a, m_sum + m
    torch.testing.assert_close(a_sum, a_full, rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(m_sum, m_full, rtol=1e-10, atol=1e-10)


def test_shard_rejects_indivisible():
    from bitdance_amd._lib import BitDanceHipError
    sd = tm.seeded_state(tm.head_shapes(tm.TINY_HEAD), seed=11)              # 2 heads
    with pytest.raises(BitDanceHipError):
        shard_head_state(sd, 0, 4)
    with pytest.raises(BitDanceHipError):
        shard_llm_state(tm.seeded_state(tm.llm_shapes(tm.TINY_LLM), seed=22), tm.TINY_LLM, 0, 4)   # 2 kv heads


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = {k: v.to(F64) for k, v in tm.seeded_state(tm.head_shapes(HEAD), seed=3).items()}
    x = torch.randn(2, 16, 1024, dtype=F64, generator=torch.Generator().manual_seed(1))
    a, m = head_block_partials(shard_head_state(sd, rank, world), x, 8 // world)
    dist.all_reduce(a)
    dist.all_reduce(m)
    a_full, m_full = head_block_partials(sd, x, 8)
    q.put((rank, float((a - a_full).abs().max()), float((m - m_full).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_allreduce_of_rank_partials():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ea, em in res:
        assert ea < 1e-9 and em < 1e-9, (rank, ea, em)
