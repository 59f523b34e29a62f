"""Binary tokenizer (conv autoencoder + sign quantiser).  The torch modules below carry the checkpoint surface (and run the
encoder, and the decoder on CPU / behind ``native_decoder = False``, on MIOpen / rocBLAS through torch.nn); on a GPU
``VQModel.decode`` runs the DECODER on the hand-written gfx950 kernels (ae_native.py / csrc/bd_conv.hip, SURVEY.md section 8f
row 2).  Module/parameter names mirror the reference checkpoint
(``ae.safetensors``: ``encoder.*`` / ``decoder.*``, modeling/vision_encoder/autoencoder.py) so that
``load_state_dict(strict=True)`` works on the released files; the implementation itself is written against the
checkpoint layout, not copied.
"""
from __future__ import annotations

import os

# MIOpen has no tuned entries for gfx950 in this image, so the first decode at a new shape runs its find pass; by default that
# pass also BENCHMARKS the naive direct-convolution solver (0.5 s per 1024-px layer: 62 of the 64 s of a first 1024 x 1024 decode).
# Keeping that one solver out of the search leaves the chosen kernels and the steady state (0.17 s) unchanged and the first call
# at 26 s (tools/ae_first_decode.py).  Read by MIOpen at its first use; a user's own setting wins.
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")

import torch
import torch.nn as nn
import torch.nn.functional as F


def _act(x):
    return x * torch.sigmoid(x)


def _gn(ch):
    return nn.GroupNorm(32, ch, eps=1e-6)


class ResBlock(nn.Module):
    """GN -> swish -> conv3 -> GN -> swish -> conv3 (+ 1x1 shortcut on channel change).  autoencoder.py:13-57"""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = _gn(cin)
        self.norm2 = _gn(cout)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1, bias=False)

    def forward(self, x):
        h = self.conv1(_act(self.norm1(x)))
        h = self.conv2(_act(self.norm2(h)))
        return h + (self.nin_shortcut(x) if self.cin != self.cout else x)


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    """autoencoder.py:59-127"""

    def __init__(self, *, ch, out_ch, in_channels, num_res_blocks, z_channels, ch_mult=(1, 2, 2, 4),
                 resolution=None, double_z=False):
        super().__init__()
        self.nlev, self.nres = len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1, bias=False)
        self.down = nn.ModuleList()
        mults = (1,) + tuple(ch_mult)
        cin = ch
        for lv in range(self.nlev):
            cin, cout = ch * mults[lv], ch * ch_mult[lv]
            level = _Level()
            level.block = nn.ModuleList()
            for _ in range(num_res_blocks):
                level.block.append(ResBlock(cin, cout))
                cin = cout
            if lv < self.nlev - 1:
                level.downsample = nn.Conv2d(cout, cout, 3, stride=2, padding=1)
            self.down.append(level)
        self.mid_block = nn.ModuleList([ResBlock(cin, cin) for _ in range(num_res_blocks)])
        self.norm_out = _gn(cin)
        self.conv_out = nn.Conv2d(cin, z_channels, 1)

    def forward(self, x):
        x = self.conv_in(x)
        for lv, level in enumerate(self.down):
            for blk in level.block:
                x = blk(x)
            if lv < self.nlev - 1:
                x = level.downsample(x)
        for blk in self.mid_block:
            x = blk(x)
        return self.conv_out(_act(self.norm_out(x)))


def depth_to_space(x: torch.Tensor, r: int) -> torch.Tensor:
    """DCR depth-to-space: channel index = (dy, dx, c).  autoencoder.py:198-230"""
    b, c, h, w = x.shape
    x = x.view(b, r, r, c // (r * r), h, w).permute(0, 3, 4, 1, 5, 2)
    return x.reshape(b, c // (r * r), h * r, w * r)


class Upsampler(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim * 4, 3, padding=1)

    def forward(self, x):
        return depth_to_space(self.conv1(x), 2)


class AdaptiveGroupNorm(nn.Module):
    """GroupNorm whose scale/bias come from per-channel std / mean of the token map.  autoencoder.py:251-277"""

    def __init__(self, z_channel: int, ch: int, eps: float = 1e-6):
        super().__init__()
        self.gn = nn.GroupNorm(32, ch, eps=eps, affine=False)
        self.gamma = nn.Linear(z_channel, ch)
        self.beta = nn.Linear(z_channel, ch)
        self.eps = eps

    def forward(self, x, tokens):
        b, c = x.shape[:2]
        flat = tokens.flatten(2)
        scale = self.gamma((flat.var(dim=-1) + self.eps).sqrt()).view(b, c, 1, 1)
        bias = self.beta(flat.mean(dim=-1)).view(b, c, 1, 1)
        return scale * self.gn(x) + bias


class Decoder(nn.Module):
    """autoencoder.py:129-196; ``gan=True``: the ``GANDecoder`` variant (:279-351) -- the same ladder with ``conv_in`` taking the
    token map concatenated with a fresh ``torch.randn_like`` noise map (one more normal draw from the global generator per decode,
    :329-330), same state-dict keys."""

    def __init__(self, *, ch, out_ch, in_channels, num_res_blocks, z_channels, ch_mult=(1, 2, 2, 4),
                 resolution=None, double_z=False, gan: bool = False):
        super().__init__()
        self.nlev, self.nres = len(ch_mult), num_res_blocks
        self.gan = bool(gan)
        cin = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels * (2 if gan else 1), cin, 3, padding=1, bias=True)
        self.mid_block = nn.ModuleList([ResBlock(cin, cin) for _ in range(num_res_blocks)])
        self.up = nn.ModuleList()
        self.adaptive = nn.ModuleList()
        for lv in reversed(range(self.nlev)):
            cout = ch * ch_mult[lv]
            self.adaptive.insert(0, AdaptiveGroupNorm(z_channels, cin))
            level = _Level()
            level.block = nn.ModuleList()
            for _ in range(num_res_blocks):
                level.block.append(ResBlock(cin, cout))
                cin = cout
            if lv > 0:
                level.upsample = Upsampler(cin)
            self.up.insert(0, level)
        self.norm_out = _gn(cin)
        self.conv_out = nn.Conv2d(cin, out_ch, 3, padding=1)

    def forward(self, z):
        tokens = z
        if self.gan:
            z = torch.cat([z, torch.randn_like(z).to(z.device)], dim=1)      # GANDecoder.forward :329-330
        z = self.conv_in(z)
        for blk in self.mid_block:
            z = blk(z)
        for lv in reversed(range(self.nlev)):
            z = self.adaptive[lv](z, tokens)
            for blk in self.up[lv].block:
                z = blk(z)
            if lv > 0:
                z = self.up[lv].upsample(z)
        return self.conv_out(_act(self.norm_out(z)))


class VQModel(nn.Module):
    """encode -> where(h>0,+1,-1) ; decode.  autoencoder.py:354-521 (``gan_decoder=True``: the GANDecoder ladder, :279-351)."""

    def __init__(self, ddconfig, checkpoint=None, gan_decoder=False):
        super().__init__()
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig, gan=bool(gan_decoder))
        # GPU decode on the native kernels (False: torch / MIOpen, the cross-check path).  Built on first use: the weights have to
        # be loaded first.  The native path implements the bf16-autocast flow the pipelines decode under.
        self.native_decoder = True
        self.native_encoder = True                            # the same for encode() (image-conditioned generation: mllm.encode_image)
        self._native_state = {}                               # {"dec" / "enc": packed twin + the weights it was packed from}

    @staticmethod
    def _bf16_autocast_on(t) -> bool:
        return t.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16

    @staticmethod
    def _weights_key(module, device):
        """Identity of the weights a packed native copy was made from: any load_state_dict (also through a parent module or on the
        sub-module), ``.to()``, or in-place update changes a parameter's storage pointer or version counter."""
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in module.parameters())

    def _native_for(self, which, module, device):
        """The packed native twin of ``module`` on ``device``, rebuilt when the weights changed; None when the native kernels do not
        cover this configuration (channel counts off the 32-multiples, other GroupNorm groupings, ...) or a gradient is wanted
        (the native path is inference only) -- the caller then runs the torch module, on the same device."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
            return None
        key = self._weights_key(module, device)
        slot = self._native_state.setdefault(which, {"key": None, "obj": None, "why": None})
        if slot["key"] != key:
            from . import ae_native
            from ._lib import BitDanceUnsupported
            slot.update(key=key, obj=None, why=None)
            try:
                slot["obj"] = (ae_native.NativeDecoder if which == "dec" else ae_native.NativeEncoder)(module, device)
            except (BitDanceUnsupported, ValueError, NotImplementedError) as e:   # a configuration the kernels do not cover; launch failures propagate
                slot["why"] = str(e)                          # decided once per set of weights; VQModel.native_fallback_reason reports it
        return slot["obj"]

    @property
    def native_fallback_reason(self):
        """{"dec" / "enc": why the torch module runs instead of the native kernels} for the paths that fell back."""
        return {k: v["why"] for k, v in self._native_state.items() if v["why"]}

    def encode(self, x):
        nat = None
        if self.native_encoder and self._bf16_autocast_on(x) and x.shape[1] <= 32 and \
                x.shape[-1] % (1 << (self.encoder.nlev - 1)) == 0 and x.shape[-2] % (1 << (self.encoder.nlev - 1)) == 0:
            nat = self._native_for("enc", self.encoder, x.device)
        h = self._run_native("enc", nat, "encode", x) if nat is not None else None
        if h is None:
            h = self.encoder(x)
        one = torch.ones((), dtype=h.dtype, device=h.device)
        return torch.where(h > 0, one, -one)

    def decode(self, quant):
        nat = self._native_for("dec", self.decoder, quant.device) if (self.native_decoder and self._bf16_autocast_on(quant)) else None
        out = self._run_native("dec", nat, "decode", quant) if nat is not None else None
        return out if out is not None else self.decoder(quant)

    def _run_native(self, which, nat, method, t):
        """A launcher that refuses a shape (host-side validation, nothing has been launched: BitDanceUnsupported / the C ABI's
        BD_ERR_UNSUPPORTED) retires the native twin for these weights and the torch module runs; every other native error -- a
        failed launch, a runtime fault after kernels were issued -- propagates (it must not turn into a silent change of path,
        speed and numerics)."""
        from ._lib import BitDanceUnsupported
        try:
            return getattr(nat, method)(t)
        except BitDanceUnsupported as e:
            self._native_state[which].update(obj=None, why=str(e))
            return None

    def forward(self, x):
        q = self.encode(x)
        return self.decode(q), q
