"""CPU-only checks of the host side: C-ABI surface, scalar schedule, tokenizer (AE) module, loud failure modes."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" and z[k].ndim > 0 else z[k]) for k in z.files}


def test_library_exports_every_declared_symbol():
    """libbitdance_hip.so loads (no GPU needed) and exports every function include/bitdance_hip.h declares."""
    from bitdance_amd import build
    from bitdance_amd._lib import EXPORTED_SYMBOLS, LIB_PATH
    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "bitdance_hip.h")).read()
    declared = set(re.findall(r"\b(bd_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    so = ctypes.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    so.bd_version.restype = ctypes.c_int
    assert so.bd_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    from bitdance_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libbitdance_hip.so")
    with pytest.raises(_lib.BitDanceHipError):
        _lib.lib()


def test_pipeline_refuses_cpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bitdance_amd.t2i_pipeline import BitDanceT2IPipeline
    with pytest.raises(RuntimeError):
        BitDanceT2IPipeline.from_components(tokenizer=None, llm_cfg={}, llm_sd={}, ae_config={}, ae_sd={},
                                            head_config={}, head_sd={}, proj_sd={}, device="cpu")


@pytest.mark.parametrize("n,shift", [(3, 1.0), (6, 1.0), (50, 1.0), (20, 3.0)])
def test_sampler_scalars_match_oracle(n, shift):
    """Host schedule == the oracle's restatement of the reference's 0-dim tensor arithmetic, bit for bit (also with a
    head-config time_shift != 1, sampling_x.py:3-4,62-63)."""
    from bitdance_amd.engine import sampler_scalars
    from oracle import sampler
    sc, ts = sampler_scalars(n, "cpu", time_shift=shift)
    ots, odts = sampler.step_table(n, time_shift=shift)
    for i in range(n):
        t, dt = ots[i], odts[i]
        want = torch.stack([t, dt, (1 - t).clamp_min(0.05), (1 - t) ** 2 - (t / 1) * -1 * (1 - t), 1 - t,
                            (2.0 * (1.0 - t) * dt) ** 0.5])
        assert torch.equal(sc[i], want), i
    assert float(sc[n, 0]) == float(torch.tensor(0.95)) and float(sc[n, 1]) == float(torch.tensor(0.05))
    assert float(sc[n, 2]) == float((1 - torch.tensor(0.95)).clamp_min(0.05))


def test_autoencoder_matches_reference(golden_dir):
    """Our tokenizer module (MIOpen/rocBLAS via torch) == the reference VQModel on the same seeded weights:
    identical state-dict keys/shapes, encode sign pattern and decode output (config 1 round trip, tiny shape)."""
    from bitdance_amd.autoencoder import VQModel
    from oracle import tiny_models as tm
    g = load(golden_dir, "ae_roundtrip")
    ae = VQModel(**tm.TINY_AE).eval()
    shapes = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    assert sorted(shapes) == [str(k) for k in g["keys"]]
    assert [str(shapes[k]) for k in sorted(shapes)] == [str(s) for s in g["shapes"]]
    ae.load_state_dict(tm.seeded_state(shapes, seed=44, gain=1.4))
    with torch.no_grad():
        h = ae.encoder(g["image"])
        q = ae.encode(g["image"])
        dec = ae.decode(g["quant"])
    torch.testing.assert_close(h, g["henc"], atol=1e-4, rtol=1e-4)
    assert (q == g["quant"]).float().mean() >= 0.999          # sign of a near-zero activation may differ by rounding
    torch.testing.assert_close(dec, g["dec"], atol=2e-4, rtol=1e-3)


def test_pos_embed_and_unraster_match_reference(golden_dir):
    from bitdance_amd.t2i_pipeline import BitDanceT2IPipeline
    g = load(golden_dir, "posembed")
    p = object.__new__(BitDanceT2IPipeline)
    p.device, p.hidden_size, p.vae_patch_size = "cpu", 256, 16
    p.build_pos_embed()
    assert torch.equal(p.pos_embed_1d, g["table"])
    assert torch.equal(p.get_2d_embed(4, 6, ps=2), g["e_4_6_2"])
    assert torch.equal(p.get_2d_embed(16, 16, ps=8), g["e_16_16_8"])


def test_row_block_padding():
    from bitdance_amd.engine import row_blocks
    assert [row_blocks(m) for m in (1, 32, 33, 64, 65, 128, 129, 256, 512)] == [1, 1, 2, 2, 4, 4, 8, 8, 16]


def test_imagenet_rope_and_mask_tables_match_reference(golden_dir):
    """The product's own 2-D RoPE table (patch-raster order) and block-causal mask against the reference's buffers
    (golden imagenet_fp32: model.freqs_cis / model.attn_mask), exact; plus BitDance-B's real geometry."""
    import numpy as np
    from bitdance_amd.imagenet import block_causal_mask, rope_table_2d
    from oracle import tiny_models as tm
    z = np.load(os.path.join(golden_dir, "imagenet_fp32.npz"))
    c = tm.TINY_IN
    fc = rope_table_2d(c["dim"] // c["n_head"], c["resolution"], 16, c["cls_token_num"], c["parallel_num"])
    assert torch.equal(fc, torch.from_numpy(z["rope"]))
    hw = c["resolution"] // 16
    m = block_causal_mask(hw * hw + c["cls_token_num"] - 1, c["cls_token_num"] - 1, c["parallel_num"])
    assert torch.equal(m, torch.from_numpy(z["mask"]))
    fc = rope_table_2d(64, 256, 16, 64, 16)                    # BitDance-B-16x: 64 cls + 15 query + 256 - 16 image positions
    assert fc.shape == (64 + 15 + 256 - 16, 32, 2) and bool((fc[:79, :, 0] == 1).all()) and bool((fc[:79, :, 1] == 0).all())


def test_ae_d16c32_round_trip_config1(golden_dir):
    """BASELINE config 1 (ae_d16c32 tokenizer, one 256x256 image, CPU fp32): our VQModel with the reference's keys at the
    released size reproduces the reference's binary latent bit for bit and its decoded pixels to fp32 conv-order noise."""
    import numpy as np
    from bitdance_amd.autoencoder import VQModel
    from oracle import tiny_models as tm
    z = np.load(os.path.join(golden_dir, "ae_c1.npz"))
    ae = VQModel(**tm.AE_D16C32).eval()
    shapes = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    assert len(shapes) == int(z["n_tensors"]) and sum(int(np.prod(v)) for v in shapes.values()) == int(z["n_params"])
    ae.load_state_dict(tm.seeded_state(shapes, seed=61, gain=1.4))
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        q = ae.encode(img)
        dec = ae.decode(q)
    assert tuple(q.shape) == tuple(z["quant_shape"]) == (1, 32, 16, 16) and set(q.unique().tolist()) <= {-1.0, 1.0}
    assert np.array_equal(np.packbits((q > 0).numpy().reshape(-1)), z["quant_bits"])          # binary tokens: exact
    got = dec.reshape(-1)[torch.from_numpy(z["sample_idx"])]
    torch.testing.assert_close(got, torch.from_numpy(z["dec_samples"]), atol=1e-3, rtol=1e-3)
    assert abs(float(dec.mean()) - float(z["dec_mean"])) < 1e-3 and abs(float(dec.std()) - float(z["dec_std"])) < 1e-3


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sharded", [False, True])
def test_load_model_dir_released_layout(tmp_path, sharded):
    """Checkpoint loading stays drop-in: a directory in the released layout (t2i_pipeline.py:45-75: HF tokenizer, config.json,
    single-file or index-sharded model safetensors, ae / vision_head / projector files) is read by the same code path the
    real constructor uses; keys, shapes, configs and the special-token lookups come back intact."""
    from bitdance_amd.t2i_pipeline import load_model_dir
    from tests.model_dir import write_model_dir
    written = write_model_dir(str(tmp_path), sharded=sharded)
    c = load_model_dir(str(tmp_path))
    assert c["llm_cfg"]["hidden_size"] == 256 and c["llm_cfg"]["head_dim"] == 128 and c["llm_cfg"]["rope_theta"] == 1000000.0
    assert c["head_config"]["parallel_num"] == 64 and c["ae_config"]["ddconfig"]["z_channels"] == 32
    assert set(c["llm_sd"]) == set(written["llm"])
    for k, v in written["llm"].items():
        assert torch.equal(c["llm_sd"][k], v), k
    assert set(c["head_sd"]) == set(written["head"]) and set(c["proj_sd"]) == {"fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"}
    assert c["head_sd"]["net.res_blocks.0.attn.wqkv.weight"].shape == (768, 256)
    tok = c["tokenizer"]
    ids = [tok.convert_tokens_to_ids(t) for t in ("<|vision_start|>", "<|res_16|>", "<|query_1|>", "<|query_63|>")]
    assert len(set(ids)) == 4 and all(isinstance(i, int) and 0 <= i < 512 for i in ids)
    assert len(tok.encode("<|im_start|>user\na red fox<|im_end|>")) > 5


def test_gen_image_rejects_inconsistent_token_budget():
    """The reference fails loudly when max_length and the image grid disagree (pos-embed slice / rearrange); so do we, before
    any device buffer is touched (no GPU needed to hit the check: it precedes engine creation)."""
    from bitdance_amd.t2i_pipeline import BitDanceT2IPipeline
    from oracle import tiny_models as tm
    p = object.__new__(BitDanceT2IPipeline)
    p.parallel_num, p.vae_patch_size, p.tokenizer = 64, 16, tm.FakeTokenizer()
    for bad in (192, 320, 100):
        with pytest.raises(ValueError):
            p.gen_image("a", "b", guidance_scale=2.0, max_length=bad, image_size=[256, 256])


# ---------------------------------------------------------------------------------------------------------
def _ctx(ints: dict):
    from bitdance_amd._lib import lib
    l = lib()
    c = l.bd_ctx_create()
    for k, v in ints.items():
        assert l.bd_ctx_set_int(c, k.encode(), int(v)) == 0, (k, l.bd_last_error())
    return l, c


DIMS_14B = {"B": 1, "branches": 2, "P": 64, "head.D": 5120, "head.C": 32, "head.Dz": 5120, "head.H": 7680, "head.nblocks": 6,
            "head.nada": 2, "head.T": 4096, "proj.D": 5120, "proj.C": 32, "llm.D": 5120, "llm.L": 40, "llm.nh": 40, "llm.nkv": 8,
            "llm.F": 17408, "llm.head_dim": 128, "llm.Lmax": 4352, "llm.splits": 8}


def _cfg(l, c, name):
    s, nw = ctypes.c_int(), ctypes.c_int()
    assert l.bd_gemm_config(c, name.encode(), ctypes.byref(s), ctypes.byref(nw)) == 0
    return s.value, nw.value & 15, ((nw.value >> 8) & 3) + 1


def test_context_planning_is_host_only_and_rejects_unknown_keys():
    """bd_ctx_* up to bd_ctx_finalize is host code (no GPU): unknown keys are errors, never silent defaults; the launch plan at
    BitDance-14B-64x dimensions is the one the headline benchmark runs (9-wave ragged adaLN tiles, 2-slice qkv / w1, the N = 5120
    shapes as 64-column tiles of 2 panels x 2 K-parts over 3 slices) and every workspace has a positive size."""
    l, c = _ctx(DIMS_14B)
    assert l.bd_ctx_set_int(c, b"head.Dd", 1) != 0 and b"unknown key" in l.bd_last_error()
    assert l.bd_ctx_set_ptr(c, b"head.blk0.wqkx", 0) != 0
    assert l.bd_ctx_set_ptr(c, b"head.blk12.wqkv", 0) == 0 and l.bd_ctx_set_ptr(c, b"llm.l39.wdown_s", 0) == 0
    assert l.bd_ctx_set_float(c, b"llm.epsilon", 1e-6) != 0 and l.bd_ctx_set_float(c, b"llm.eps", 1e-6) == 0
    assert l.bd_ctx_set_int(c, b"tune.head.wo.S", 2) == 0 and l.bd_ctx_set_int(c, b"tune.head.wq.S", 6) != 0
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert _cfg(l, c, "head.ada") == (1, 9, 1)            # 2240 panels over 249 workgroups of 9 waves, ragged last tile
    assert _cfg(l, c, "head.qkv") == (2, 4, 1) and _cfg(l, c, "head.w1") == (2, 4, 1)
    assert _cfg(l, c, "head.wo") == (2, 4, 2)             # the per-GEMM override above
    assert _cfg(l, c, "head.w2") == (3, 4, 2) and _cfg(l, c, "llm.o") == (3, 4, 2) and _cfg(l, c, "head.cond") == (3, 4, 2)
    assert _cfg(l, c, "llm.qkv") == (4, 8, 2) and _cfg(l, c, "llm.gu") == (1, 8, 1) and _cfg(l, c, "llm.down") == (9, 8, 1)
    n = l.bd_ctx_ws_count(c)
    names = [l.bd_ctx_ws_name(c, i).decode() for i in range(n)]
    assert "llm.k_cache" in names and "head.ada_bf" in names and "head.tp_part" not in names
    assert all(l.bd_ctx_ws_bytes(c, i) > 0 for i in range(n))
    assert l.bd_ctx_ws_name(c, n) is None and l.bd_ctx_ws_bytes(c, -1) < 0           # bounds-checked
    assert l.bd_ctx_bind(c) != 0                                                      # workspaces not provided yet
    l.bd_ctx_destroy(c)


@pytest.mark.parametrize("tp", [2, 4, 8])
def test_context_planning_tensor_parallel(tp):
    """Per-rank plan of the 14B model (bd_ctx_set_tp: planning needs no communicator): local widths divide, the row-split
    Linears (wo / w2 / o_proj / down_proj) produce ONE finished partial (at most 3 grid slices, reduced in the launch), the KV
    cache shrinks with the kv heads, an indivisible size is rejected."""
    l, c = _ctx(DIMS_14B)
    assert l.bd_ctx_set_tp(c, tp - 1, tp) == 0
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    for name in ("head.wo", "head.w2", "llm.o", "llm.down"):
        assert _cfg(l, c, name)[0] <= 3, (name, _cfg(l, c, name))
    ws = {l.bd_ctx_ws_name(c, i).decode(): l.bd_ctx_ws_bytes(c, i) for i in range(l.bd_ctx_ws_count(c))}
    assert "head.tp_part" in ws and ws["head.tp_part"] == 128 * 5120 * 4 and "llm.tp_part" in ws
    l1, c1 = _ctx(DIMS_14B)
    assert l1.bd_ctx_finalize(c1) == 0
    ws1 = {l1.bd_ctx_ws_name(c1, i).decode(): l1.bd_ctx_ws_bytes(c1, i) for i in range(l1.bd_ctx_ws_count(c1))}
    assert ws["llm.k_cache"] * tp == ws1["llm.k_cache"] and ws["head.act_frag"] * tp == ws1["head.act_frag"]
    l.bd_ctx_destroy(c); l1.bd_ctx_destroy(c1)
    l3, c3 = _ctx(DIMS_14B)
    assert l3.bd_ctx_set_tp(c3, 0, 3) == 0 and l3.bd_ctx_finalize(c3) != 0 and b"divide" in l3.bd_last_error()
    assert l3.bd_ctx_set_tp(c3, 3, 3) != 0
    l3.bd_ctx_destroy(c3)


def test_tensor_parallel_shard_launch_rules():
    """choose_cfg's rules for the tensor-parallel shards of the 128-row passes (round 5; measured on one rank in loop-back,
    profiles/r05_tp_rank_critical_path.log): a column-split Linear with few columns per rank runs 64-column tiles (2 panels x 2 K
    parts) over up to 8 short K slices whose slabs a row-parallel pass sums; a row-split Linear with a short local K runs ONE slice
    (no in-launch reduction in front of the push epilogue); "tune.tp_shapes" = 0 restores the tp = 1 rules; "tp.seq" needs a
    communicator with an operand landing buffer."""
    want = {8: {"head.qkv": (8, 4, 2), "head.w1": (8, 4, 2), "head.wo": (1, 2, 2), "head.w2": (1, 2, 1)},
            4: {"head.qkv": (4, 4, 2), "head.w1": (4, 4, 2)},
            2: {}}
    for tp, cfgs in want.items():
        l, c = _ctx(DIMS_14B)
        assert l.bd_ctx_set_tp(c, 0, tp) == 0 and l.bd_ctx_finalize(c) == 0, l.bd_last_error()
        for name, cfg in cfgs.items():
            assert _cfg(l, c, name) == cfg, (tp, name, _cfg(l, c, name))
        for name in ("head.wo", "head.w2", "llm.o", "llm.down"):
            assert _cfg(l, c, name)[0] <= 3
        l.bd_ctx_destroy(c)
    l, c = _ctx({**DIMS_14B, "tune.tp_shapes": 0})
    assert l.bd_ctx_set_tp(c, 0, 8) == 0 and l.bd_ctx_finalize(c) == 0
    assert _cfg(l, c, "head.qkv")[0] > 8 and _cfg(l, c, "head.wo")[0] == 3          # the tp = 1 rules: 14 slices of 128-column tiles; 3 reduced slices
    l.bd_ctx_destroy(c)
    l, c = _ctx({**DIMS_14B, "tp.seq": 1})
    assert l.bd_ctx_set_tp(c, 0, 2) == 0 and l.bd_ctx_finalize(c) != 0 and b"tp.seq" in l.bd_last_error()
    l.bd_ctx_destroy(c)


def test_tensor_parallel_buffer_sizes():
    """tp.ada_gather_bytes: TWO slots of one group's modulation tensor (double-buffered by group parity: a peer may push group
    g + 1 while this rank still reads g); tp.seq_hbuf_bytes: the operand rows of a 128-row pass (+ the fp32 landing area of the Qwen3 step's
    final hand-off), nothing for other row counts."""
    from bitdance_amd.tp import ada_gather_bytes, seq_hbuf_bytes
    assert ada_gather_bytes(128, 14 * 5120) == 2 * 4 * 128 * 14 * 5120 * 2
    assert ada_gather_bytes(32, 14 * 5120) == 2 * 16 * 32 * 14 * 5120 * 2
    assert ada_gather_bytes(512, 14 * 5120) == 2 * 2 * 512 * 14 * 5120 * 2 and ada_gather_bytes(2048, 1024) == 0
    assert seq_hbuf_bytes(128, 5120) == 128 * 5120 * 6 and seq_hbuf_bytes(512, 5120) == 0 and seq_hbuf_bytes(32, 5120) == 0


def test_small_weight_tile_rule_for_the_imagenet_batches():
    """choose_cfg (bd_api.hip): small weights under a few thousand rows run 256 x 128 (4 waves) or 128 x 64 (2 waves) tiles at one K
    slice with the fused epilogues instead of 256 x 256 tiles split 12-18 ways (B-1x: 768 rows; B-4x: 3072 rows); the 12 288-row
    batch of B-16x and the 14B shapes at two / four images keep the 256-row kernels."""
    tr = {"llm.D": 768, "llm.L": 12, "llm.nh": 12, "llm.nkv": 12, "llm.F": 2048, "llm.head_dim": 64, "llm.variant": 1,
          "llm.Lmax": 320, "llm.splits": 8, "proj.D": 768, "proj.C": 32, "proj.hid": 1152, "proj.variant": 1}
    head = {"head.D": 768, "head.C": 32, "head.Dz": 768, "head.H": 1152, "head.nblocks": 6, "head.nada": 2, "head.dh": 64,
            "head.sigmoid": 0}
    l, c = _ctx({"B": 384, "branches": 2, "P": 1, "head.variant": 1, **head, **tr})          # B-1x: 768 rows
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert _cfg(l, c, "head.w1") == (1, 2, 1)              # 36 x 6 = 216 tiles of 128 x 64, SwiGLU in the epilogue
    assert _cfg(l, c, "head.w2") == (3, 2, 1)              # 12 x 6 tiles x 3 slices (was 18 slabs)
    assert _cfg(l, c, "llm.o") == (3, 2, 1) and _cfg(l, c, "llm.qkv") == (1, 2, 1)
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 384, "branches": 2, "P": 4, **head, **tr})                             # B-4x: 3072 rows
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert _cfg(l, c, "head.qkv") == (1, 4, 1) and _cfg(l, c, "head.w1") == (1, 4, 1)      # 18 x 12 = 216 tiles of 256 x 128
    assert _cfg(l, c, "head.wo") == (1, 2, 1) and _cfg(l, c, "head.w2") == (1, 2, 1)       # 12 x 24 = 288 tiles of 128 x 64
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 384, "branches": 2, "P": 16, **head, **tr})                            # B-16x: 12 288 rows stay on 256 x 256
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert all(_cfg(l, c, n) == (1, 8, 1) for n in ("head.qkv", "head.w1", "head.wo", "head.w2"))
    l.bd_ctx_destroy(c)
    l, c = _ctx({**DIMS_14B, "B": 4})                                                        # 14B, four images: 52-356 MB weights
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert _cfg(l, c, "head.qkv")[1] == 8 and _cfg(l, c, "head.wo")[1] == 8
    l.bd_ctx_destroy(c)


def test_context_planning_imagenet_1x_and_4x_variants():
    """Host-only planning of the other ImageNet variants: the MLP head (head.variant = 1) at BitDance-B-1x
    dimensions with one token per step -- 2 adaLN blocks of THREE chunks + the final layer's two = 8 x 768 adaLN columns -- and
    the 4-token parallel variant; the transformer head and the Qwen3 path also plan at 4 / 1 tokens per step (the full-causal T2I
    loop); an unknown head variant and a token count outside {1, 4, 16, 64} are rejected."""
    head_b = {"head.D": 768, "head.C": 32, "head.Dz": 768, "head.H": 1152, "head.nblocks": 6, "head.nada": 2, "head.dh": 64,
              "head.sigmoid": 0}
    l, c = _ctx({"B": 384, "branches": 2, "P": 1, "head.variant": 1, **head_b})
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    ws = {l.bd_ctx_ws_name(c, i).decode(): l.bd_ctx_ws_bytes(c, i) for i in range(l.bd_ctx_ws_count(c))}
    assert ws["head.ada_bf"] == 768 * (2 * 3 + 2) * 768 * 2                 # [768 rows][8 x 768] bf16
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 384, "branches": 2, "P": 4, **head_b, "head.H": 2048})
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    ws = {l.bd_ctx_ws_name(c, i).decode(): l.bd_ctx_ws_bytes(c, i) for i in range(l.bd_ctx_ws_count(c))}
    assert ws["head.ada_bf"] == 3072 * (2 * 6 + 2) * 768 * 2
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 2, "branches": 2, "P": 1, **head_b})                  # a transformer head with one token per step: attention
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()                      # over one key (out = v) -- the full-causal T2I loop
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 2, "branches": 2, "P": 1, "head.variant": 2, **head_b})
    assert l.bd_ctx_finalize(c) != 0
    l.bd_ctx_destroy(c)
    for P in (4, 1):                                                         # the Qwen3 decode path at 4 / 1 tokens per step
        l, c = _ctx({**DIMS_14B, "P": P})                                    # (MLLModel.gen_image_full_causal, mllm.py:274-384)
        assert l.bd_ctx_finalize(c) == 0, (P, l.bd_last_error())
        l.bd_ctx_destroy(c)
    l, c = _ctx({**DIMS_14B, "P": 8})
    assert l.bd_ctx_finalize(c) != 0 and b"parallel_num" in l.bd_last_error()
    l.bd_ctx_destroy(c)
    tr = {"llm.D": 768, "llm.L": 24, "llm.nh": 12, "llm.nkv": 12, "llm.F": 2048, "llm.head_dim": 64, "llm.variant": 1,
          "llm.Lmax": 320, "llm.splits": 8, "proj.D": 768, "proj.C": 32, "proj.hid": 1152, "proj.variant": 1}
    for P in (1, 4, 16):
        l, c = _ctx({"B": 768, "branches": 1, "P": P, **tr})
        assert l.bd_ctx_finalize(c) == 0, (P, l.bd_last_error())
        l.bd_ctx_destroy(c)


def test_mllm_vt_forward_and_plan_helpers(golden_dir):
    """Host pieces of MLLModel.forward_inference_block_causal / encode_image (modeling/mllm.py:695-930) without a GPU: the
    tokenizer pass of encode_image (VQModel.vt_forward: encode -> binary -> 'c (h p1) (w p2) -> (h w p1 p2) c', images grouped by
    size, list order kept) reproduces the reference's latents of the interleaved golden exactly; remove_first_user_block
    (utils.py:206-216) as the reference."""
    from bitdance_amd.autoencoder import VQModel
    from bitdance_amd.mllm import MLLModel
    from oracle import tiny_models as tm
    g = load(golden_dir, "interleaved_fp32")
    ae = VQModel(**tm.TINY_AE).eval()
    ae.load_state_dict(tm.seeded_state({k: tuple(v.shape) for k, v in ae.state_dict().items()}, seed=44, gain=1.4))
    m = object.__new__(MLLModel)
    m.device, m.vision_encoder, m.ps, m.vae_patch_size = "cpu", ae, 8, 16
    lat = m.vt_forward([g["image"]], ps=8)
    assert torch.equal(lat, g["image_latents"])
    small = torch.flip(g["image"], dims=[-1])[..., :128, :128]
    other = torch.nn.functional.interpolate(g["image"], size=(256, 128))
    both = m.vt_forward([g["image"], other, small], ps=8)                     # two sizes, grouped; output in list order
    assert both.shape == (64 + 128 + 64, 32) and torch.equal(both[:64], lat) and torch.equal(both[192:], m.vt_forward([small], ps=8))
    r = MLLModel.remove_first_user_block
    assert r("<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n") == "<|im_start|>assistant\n"
    assert r("<|im_start|>user\nhi") == "<|im_start|>user\nhi" and r("plain") == "plain"
    assert r("A<|im_start|>user\nx<|im_end|>\nB<|im_start|>user\ny<|im_end|>\n") == "AB<|im_start|>user\ny<|im_end|>\n"


def test_launch_plan_rules_measured_in_round_2():
    """Host-only pins of the launch rules the round-2 sweeps decided (profiles/r02_gemm_sweep3.log, r02_bench_imagenet_*):
    ragged 9-wave tiles for the adaLN projection only (gate/up keeps 8 waves: the 5-wave and 5x2 forms measured slower and stay
    options), the slab cap from 8 row tiles up (ImageNet B-4x: one slice everywhere) but not at 3 row tiles (B-1x keeps 6 / 18)."""
    l, c = _ctx(DIMS_14B)
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert _cfg(l, c, "head.ada") == (1, 9, 1) and _cfg(l, c, "llm.gu") == (1, 8, 1) and _cfg(l, c, "llm.down")[0] == 9
    l.bd_ctx_destroy(c)
    l, c = _ctx({**DIMS_14B, "tune.ragged": 0})
    assert l.bd_ctx_finalize(c) == 0 and _cfg(l, c, "head.ada") == (1, 10, 1)
    l.bd_ctx_destroy(c)
    l, c = _ctx({**DIMS_14B, "tune.ragged52": 1})
    assert l.bd_ctx_finalize(c) == 0 and _cfg(l, c, "llm.gu") == (1, 10, 2)
    l.bd_ctx_destroy(c)
    head_b = {"head.D": 768, "head.C": 32, "head.Dz": 768, "head.nblocks": 6, "head.nada": 2, "head.dh": 64, "head.sigmoid": 0}
    l, c = _ctx({"B": 384, "branches": 2, "P": 4, "head.H": 2048, **head_b})           # B-4x: 3072 rows = 12 row tiles
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert all(_cfg(l, c, "head." + n)[0] == 1 for n in ("qkv", "wo", "w1", "w2", "ada"))
    l.bd_ctx_destroy(c)
    # (round 4: these batches now take the small-weight tile rule, test_small_weight_tile_rule_for_the_imagenet_batches;
    #  tune.small_tiles_rows = 0 keeps the 256 x 256 tiles whose rules are pinned here)
    old = {"tune.small_tiles_rows": 0}
    l, c = _ctx({"B": 384, "branches": 2, "P": 4, "head.H": 2048, "tune.slab_cap": 0, **old, **head_b})
    assert l.bd_ctx_finalize(c) == 0 and _cfg(l, c, "head.w2")[0] > 1
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 384, "branches": 2, "P": 4, "head.H": 2048, **old, **head_b})
    assert l.bd_ctx_finalize(c) == 0 and all(_cfg(l, c, "head." + n) == (1, 8, 1) for n in ("qkv", "wo", "w1", "w2"))
    l.bd_ctx_destroy(c)
    l, c = _ctx({"B": 384, "branches": 2, "P": 1, "head.variant": 1, "head.H": 1152, **old, **head_b})   # B-1x: 768 rows = 3 row tiles
    assert l.bd_ctx_finalize(c) == 0, l.bd_last_error()
    assert _cfg(l, c, "head.w1")[0] == 6 and _cfg(l, c, "head.w2")[0] == 18
    l.bd_ctx_destroy(c)


def test_bench_roofline_accounting():
    """bench.py's roofline arithmetic on a synthetic profile (no GPU): `achieved` = bytes PHYSICALLY streamed by all GEMM launches over the
    summed event time (a launch named "<gemm>[xG]" serves G evaluations in one pass and moved its weights once), frac == achieved / peak
    and can never exceed 1 by double counting; `algorithmic` = N*K*2 per Linear AND evaluation over the same time (not a fraction of
    the peak); `hbm_bound_launches` = the one-evaluation launches alone; the grouped launch carries its TFLOP/s."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_roofline_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    wq, wa = 15360 * 5120 * 2.0, 71680 * 5120 * 2.0

    class FakeEngine:
        wdtype = 0

        def profile_gemms(self, run):
            return {"head.qkv": dict(count=10, ms=10 * 0.040, bytes=10 * wq), "head.ada[x4]": dict(count=2, ms=2 * 0.400, bytes=2 * wa)}

        def gemm_config(self, name):
            return (2, 4 + 16 * 3) if name == "head.qkv" else (1, 9 + 16 * 2)
    r = bench.gemm_roofline(FakeEngine(), lambda: None, 128)
    ms = 10 * 0.040 + 2 * 0.400
    assert r["bound"] == "hbm" and r["launches"] == 12
    assert abs(r["achieved"] - (10 * wq + 2 * wa) / ms / 1e6) < 0.5 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert abs(r["algorithmic"]["GBs"] - (10 * wq + 2 * 4 * wa) / ms / 1e6) < 0.5 and "frac" not in r["algorithmic"]
    assert r["hbm_bound_launches"]["launches"] == 10 and abs(r["hbm_bound_launches"]["achieved"] - wq / 0.040 / 1e6) < 0.5
    grouped = [g for g in r["per_gemm"] if g["name"] == "head.ada[x4]"][0]
    assert grouped["evaluations_per_launch"] == 4 and grouped["rows_per_pass"] == 512
    assert abs(grouped["TFLOPs"] - 2.0 * 512 * 71680 * 5120 / 0.400e-3 / 1e12) < 1.0
    assert abs(grouped["algorithmic_GBs"] - 4 * wa / 0.400 / 1e6) < 0.5 and abs(grouped["GBs"] - wa / 0.400 / 1e6) < 0.5


def test_bench_pmc_annotation_reads_the_committed_passes():
    """bench.py attaches `mfma_busy` / `eff_clock_ghz` of the committed PMC passes (profiles/r06_pmc_gemm_traffic*.json) to a per-GEMM
    row only when that pass measured the SAME launch configuration; the 512-row pass feeds the `b4` object."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    per = [{"name": "head.qkv", "splitk": 2, "nwaves": 4, "kparts": 1}, {"name": "head.qkv", "splitk": 4, "nwaves": 4, "kparts": 1},
           {"name": "head.nope", "splitk": 1, "nwaves": 4, "kparts": 1}]
    src = b.annotate_pmc(per, 128)
    assert src and src.endswith("r06_pmc_gemm_traffic.json")
    assert 0.1 < per[0]["mfma_busy"] < 0.5 and 1.0 < per[0]["eff_clock_ghz"] < 4.0
    assert "mfma_busy" not in per[1] and "mfma_busy" not in per[2]          # another split-K / an unknown GEMM: no numbers invented
    per4 = [{"name": "head.wo", "splitk": 3, "nwaves": 8, "kparts": 1}, {"name": "head.wo", "splitk": 5, "nwaves": 8, "kparts": 1}]   # (round 6: 3 slabs on the 256 x 128-tile kernel)
    assert b.annotate_pmc(per4, 512).endswith("r06_pmc_gemm_traffic_rows512.json") and "mfma_busy" in per4[0] and "mfma_busy" not in per4[1]
    assert b.pmc_entries(256) == {}
