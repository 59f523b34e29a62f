"""CPU restatement of the binary tokenizer's conv encoder / decoder (test infrastructure, see oracle/__init__.py).

Follows /root/reference/modeling/vision_encoder/autoencoder.py function by function, from a plain state dict (reference key
names), with the dtype policy made explicit:

  * ``Policy("fp32")``     -- what the reference computes on a CPU with fp32 weights; pinned against tests/golden/ae_roundtrip.npz
                              (the unmodified reference module's encoder latent and decoder output on seeded weights).
  * ``Policy("autocast")`` -- the CUDA/HIP bf16 autocast flow the pipelines decode / encode under (t2i_pipeline.py:130):
                              ``conv2d`` and ``linear`` cast their inputs to bf16, accumulate in fp32 and round the result once to
                              bf16; ``group_norm`` is on autocast's fp32 list (fp32 in, fp32 out); swish, the AdaGN affine and the
                              residual additions follow ordinary type promotion (bf16 + bf16 -> bf16, bf16 + fp32 -> fp32).

The native kernels (bitdance_amd/csrc/bd_conv.hip behind bitdance_amd/ae_native.py) are compared with the "autocast" policy
(tests/test_gpu_ae.py); nothing under bitdance_amd/ imports this file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .numerics import BF16, F32, Policy


def conv2d(pol: Policy, x, w, b=None, stride: int = 1, padding: int = 0):
    """nn.Conv2d.forward under the policy (autocast: bf16 operands, fp32 accumulation, one rounding of sum + bias to bf16)."""
    if not pol.amp:
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    y = F.conv2d(x.to(BF16).to(F32), w.to(BF16).to(F32), None if b is None else b.to(BF16).to(F32), stride=stride, padding=padding)
    return y.to(BF16)


def group_norm(pol: Policy, x, w=None, b=None, eps: float = 1e-6):
    """nn.GroupNorm(32, C, eps) (autoencoder.py:28-29, 254): fp32 arithmetic and fp32 output under autocast (fp32 list)."""
    if not pol.amp:
        return F.group_norm(x, 32, w, b, eps)
    return F.group_norm(x.to(F32), 32, None if w is None else w.to(F32), None if b is None else b.to(F32), eps)


def swish(x):
    """autoencoder.py:10-11"""
    return x * torch.sigmoid(x)


def res_block(pol: Policy, sd: dict, pre: str, x):
    """ResBlock.forward (autoencoder.py:41-57; use_agn = False, 1x1 nin_shortcut on a channel change)."""
    h = conv2d(pol, swish(group_norm(pol, x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])), sd[pre + "conv1.weight"], padding=1)
    h = conv2d(pol, swish(group_norm(pol, h, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])), sd[pre + "conv2.weight"], padding=1)
    res = conv2d(pol, x, sd[pre + "nin_shortcut.weight"]) if (pre + "nin_shortcut.weight") in sd else x
    return h + res


def encoder_forward(pol: Policy, sd: dict, cfg: dict, x):
    """Encoder.forward (autoencoder.py:107-127): conv_in, per level ResBlocks (+ the stride-2 3x3 down-sampling convolution), mid
    blocks, norm_out -> swish -> 1x1 conv_out.  ``sd``: the tokenizer's state dict (keys ``encoder.*``)."""
    nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    p = "encoder."
    h = conv2d(pol, x, sd[p + "conv_in.weight"], padding=1)
    for lv in range(nlev):
        for i in range(nres):
            h = res_block(pol, sd, f"{p}down.{lv}.block.{i}.", h)
        if lv < nlev - 1:
            h = conv2d(pol, h, sd[f"{p}down.{lv}.downsample.weight"], sd[f"{p}down.{lv}.downsample.bias"], stride=2, padding=1)
    for i in range(nres):
        h = res_block(pol, sd, f"{p}mid_block.{i}.", h)
    h = swish(group_norm(pol, h, sd[p + "norm_out.weight"], sd[p + "norm_out.bias"]))
    return conv2d(pol, h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"])


def encode(pol: Policy, sd: dict, cfg: dict, x):
    """VQModel.encode's binarisation of the encoder latent (autoencoder.py:354-521: +1 where h > 0, else -1)."""
    h = encoder_forward(pol, sd, cfg, x)
    one = torch.ones((), dtype=h.dtype)
    return torch.where(h > 0, one, -one)


def depth_to_space(x, r: int = 2):
    """DCR depth-to-space (autoencoder.py:198-230): channel index = (dy, dx, c)."""
    b, c, h, w = x.shape
    x = x.view(b, r, r, c // (r * r), h, w).permute(0, 3, 4, 1, 5, 2)
    return x.reshape(b, c // (r * r), h * r, w * r)


def adaptive_group_norm(pol: Policy, sd: dict, pre: str, x, tokens, eps: float = 1e-6):
    """AdaptiveGroupNorm.forward (autoencoder.py:260-277): scale = gamma(sqrt(var_hw(tokens) + eps)), bias = beta(mean_hw(tokens)),
    out = scale * GroupNorm(x, no affine) + bias.  Under autocast the two Linears return bf16, the GroupNorm fp32: the product and the
    sum promote to fp32."""
    b, c = x.shape[:2]
    flat = tokens.flatten(2)
    scale = pol.linear((flat.var(dim=-1) + eps).sqrt(), sd[pre + "gamma.weight"], sd[pre + "gamma.bias"], quant=False).view(b, c, 1, 1)
    bias = pol.linear(flat.mean(dim=-1), sd[pre + "beta.weight"], sd[pre + "beta.bias"], quant=False).view(b, c, 1, 1)
    return scale * group_norm(pol, x, None, None, eps) + bias


def decoder_forward(pol: Policy, sd: dict, cfg: dict, z, noise=None):
    """Decoder.forward (autoencoder.py:169-196): conv_in, mid blocks, then from the coarsest level up: AdaptiveGroupNorm on the
    token map, ResBlocks, Upsampler (3x3 conv to 4x channels + depth-to-space, :232-250); norm_out -> swish -> conv_out.
    A ``conv_in`` of twice the token channels is the GANDecoder (:279-351): its input is the token map concatenated with
    ``torch.randn_like(z)`` (:329-330) -- drawn here from the global generator like the reference does, or given as ``noise``."""
    nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    p = "decoder."
    tokens = z
    if sd[p + "conv_in.weight"].shape[1] == 2 * z.shape[1]:
        z = torch.cat([z, torch.randn_like(z) if noise is None else noise], dim=1)
    h = conv2d(pol, z, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1)
    for i in range(nres):
        h = res_block(pol, sd, f"{p}mid_block.{i}.", h)
    for lv in reversed(range(nlev)):
        h = adaptive_group_norm(pol, sd, f"{p}adaptive.{lv}.", h, tokens)
        for i in range(nres):
            h = res_block(pol, sd, f"{p}up.{lv}.block.{i}.", h)
        if lv > 0:
            h = depth_to_space(conv2d(pol, h, sd[f"{p}up.{lv}.upsample.conv1.weight"], sd[f"{p}up.{lv}.upsample.conv1.bias"], padding=1), 2)
    h = swish(group_norm(pol, h, sd[p + "norm_out.weight"], sd[p + "norm_out.bias"]))
    return conv2d(pol, h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)
