"""Native conv decoder (and encoder: NativeEncoder, below) of the binary tokenizer: ``Decoder.forward`` (/root/reference/modeling/vision_encoder/autoencoder.py:129-196,
ResBlock :13-57, Upsampler / depth_to_space :198-250, AdaptiveGroupNorm :251-277) on the hand-written gfx950 kernels of
csrc/bd_conv.hip -- 3x3 / 1x1 convolutions as implicit GEMMs on the matrix pipe, GroupNorm statistics + fused
normalise / AdaGN / swish passes, depth-to-space in the upsampling convolution's epilogue -- instead of MIOpen.

This module is plumbing: it takes the weights out of the (checkpoint-compatible) torch ``Decoder`` module, packs them once, owns
the activation buffers (torch tensors) and sequences the launches with the dtype flow the reference has under bf16 autocast:
convolution outputs bf16, GroupNorm / swish / AdaGN in fp32, the residual stream fp32 after an AdaptiveGroupNorm until a
channel-changing block (bf16 conv + bf16 shortcut) or an upsampler makes it bf16 again.  No CPU path.
"""
from __future__ import annotations

import torch

from ._lib import BitDanceHipError, BitDanceUnsupported, check, lib

BF16 = torch.bfloat16


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Conv:
    """One nn.Conv2d (3x3 pad 1, or 1x1) packed for bd_conv: [Cout -> 256-multiple][taps * Cin], K ordered (ky, kx, ci)."""

    def __init__(self, conv: torch.nn.Conv2d, device, cin_pad: int = 0):
        w = conv.weight.detach().to(device=device, dtype=torch.float32)
        if cin_pad > w.shape[1]:                                # the encoder's conv_in: 3 image channels in a 32-channel (zero) operand
            w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], *w.shape[2:])], dim=1)
        self.cout, self.cin, kh, kw = w.shape
        self.taps = kh * kw
        self.stride = int(conv.stride[0])
        if (kh, kw) not in ((3, 3), (1, 1)) or self.cin % 32 or conv.stride[0] != conv.stride[1] or self.stride not in (1, 2) or \
                (self.stride == 2 and (kh != 3 or tuple(conv.padding) != (1, 1))):
            raise BitDanceUnsupported(f"native tokenizer: unsupported convolution {tuple(w.shape)} stride {tuple(conv.stride)}")
        npad = (self.cout + 255) // 256 * 256
        m = torch.zeros(npad, self.taps * self.cin, dtype=BF16, device=device)
        m[: self.cout] = w.permute(0, 2, 3, 1).reshape(self.cout, -1).to(BF16)
        self.w = torch.empty(npad * self.taps * self.cin, dtype=BF16, device=device)
        check(lib().bd_pack_weight(self.w.data_ptr(), m.data_ptr(), npad, self.taps * self.cin, 0, npad, _st()), "bd_pack_weight")
        self.bias = None if conv.bias is None else conv.bias.detach().to(device=device, dtype=BF16).contiguous()
        torch.cuda.current_stream().synchronize()


class _Norm:
    def __init__(self, gn: torch.nn.GroupNorm, device):
        if gn.num_groups != 32:
            raise BitDanceUnsupported("native decoder: GroupNorm(32) only")
        self.eps = float(gn.eps)
        self.gamma = None if gn.weight is None else gn.weight.detach().to(device, torch.float32).contiguous()
        self.beta = None if gn.bias is None else gn.bias.detach().to(device, torch.float32).contiguous()


class NativeDecoder:
    """``decode(z)`` == ``Decoder.forward(z)`` under ``torch.autocast('cuda', bfloat16)`` (values within bf16 accumulation noise of
    the MIOpen path; tests/test_gpu_ae.py).  ``dec``: a loaded ``autoencoder.Decoder``."""

    def __init__(self, dec, device):
        self.device = torch.device(device)
        self.dec = dec
        self.nlev, self.nres = dec.nlev, dec.nres
        self.gan = bool(getattr(dec, "gan", False))          # GANDecoder (autoencoder.py:279-351): conv_in over [tokens | fresh noise]
        C = lambda m: _Conv(m, self.device)
        self.conv_in = C(dec.conv_in)
        self.mid = [self._block(b) for b in dec.mid_block]
        self.levels = []
        for lv in range(self.nlev):
            level = dec.up[lv]
            self.levels.append({"blocks": [self._block(b) for b in level.block],
                                "up": C(level.upsample.conv1) if lv > 0 else None,
                                "ada": dec.adaptive[lv]})
        self.norm_out = _Norm(dec.norm_out, self.device)
        self.conv_out = C(dec.conv_out)
        self._buf: dict = {}

    def _block(self, b):
        d = {"n1": _Norm(b.norm1, self.device), "n2": _Norm(b.norm2, self.device), "c1": _Conv(b.conv1, self.device),
             "c2": _Conv(b.conv2, self.device), "sc": None}
        if b.cin != b.cout:
            d["sc"] = _Conv(b.nin_shortcut, self.device)
        return d

    # -- buffers: one per (role, shape, dtype), reused across calls; padded buffers keep their zero border ---------------------
    def _get(self, role: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        key = (role, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    def _padded(self, role, n, H, W, C):
        return self._get(role, (n, H + 2, W + 2, C), BF16, zero=True)

    def _new_call(self, n, H, W) -> None:
        """Buffers are kept per (role, shape) for the next call of the same size; a different batch / image size drops them, so a
        process that walks through many aspect ratios (t2i_pipeline.py:27-31) holds one size's activations, not all of them.
        (Everything runs on the current stream: a freed buffer is only reused behind the launches that read it.)"""
        if getattr(self, "_call_key", None) != (n, H, W):
            self._buf.clear()
            self._call_key = (n, H, W)

    # -- operators ---------------------------------------------------------------------------------------------------
    def _conv(self, cv: _Conv, x, out, n, H, W, *, mode=0, res=None):
        l = lib()
        check(l.bd_conv_strided(x.data_ptr(), cv.w.data_ptr(), None if cv.bias is None else cv.bias.data_ptr(),
                                None if res is None else res.data_ptr(), int(res is not None and res.dtype == torch.float32),
                                out.data_ptr(), mode, int(out.dtype == torch.float32), n, H, W, cv.cin, cv.cout, cv.taps, cv.stride, _st()),
              "bd_conv_strided")
        return out

    def _stats(self, x, n, H, W, C, eps):
        chunks = (H * W + 255) // 256
        part = self._get("gn.partial", (n, chunks, 32, 2), torch.float32)
        st = self._get("gn.stats", (n, 32, 2), torch.float32)
        check(lib().bd_gn_stats(x.data_ptr(), int(x.dtype == torch.float32), part.data_ptr(), st.data_ptr(), n, H * W, C, eps, _st()), "bd_gn_stats")
        return st

    def _apply(self, x, st, out, mode, swish, n, H, W, C, gamma=None, beta=None, scale=None, bias=None):
        p = lambda t: None if t is None else t.data_ptr()
        check(lib().bd_gn_apply(x.data_ptr(), int(x.dtype == torch.float32), p(st), p(gamma), p(beta), p(scale), p(bias), out.data_ptr(), mode,
                                int(swish), n, H, W, C, _st()), "bd_gn_apply")
        return out

    def _norm_swish_pad(self, nm: _Norm, x, role, n, H, W, C):
        st = self._stats(x, n, H, W, C, nm.eps)
        return self._apply(x, st, self._padded(role, n, H, W, C), 0, True, n, H, W, C, gamma=nm.gamma, beta=nm.beta)

    def _resblock(self, blk, x, n, H, W, tag):
        cin, cout = blk["c1"].cin, blk["c1"].cout
        p1 = self._norm_swish_pad(blk["n1"], x, "p.a", n, H, W, cin)
        t = self._conv(blk["c1"], p1, self._get("t", (n, H, W, cout), BF16), n, H, W)
        p2 = self._norm_swish_pad(blk["n2"], t, "p.b", n, H, W, cout)
        if blk["sc"] is not None:                              # bf16 conv + bf16 shortcut(x) -> bf16 stream
            xb = x if x.dtype == BF16 else self._apply(x, None, self._get("xb", (n, H, W, cin), BF16), 2, False, n, H, W, cin)
            sc = self._conv(blk["sc"], xb, self._get("sc", (n, H, W, cout), BF16), n, H, W)
            out = self._get("s." + tag, (n, H, W, cout), BF16)
            return self._conv(blk["c2"], p2, out, n, H, W, res=sc)
        out = self._get("s." + tag, (n, H, W, cout), x.dtype)    # bf16 + bf16 -> bf16 ; bf16 + fp32 -> fp32
        return self._conv(blk["c2"], p2, out, n, H, W, res=x)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, C, h, w] (the +-1 token map) -> [B, 3, H, W] bf16 (what ``Decoder.forward`` returns under bf16 autocast)."""
        if not z.is_cuda:
            raise BitDanceUnsupported("native decoder: CUDA/HIP tensors only (no CPU path)")
        n, Cz, H, W = z.shape
        self._new_call(n, H, W)
        zf = z.to(torch.float32).contiguous()
        zin, Cin = zf, Cz
        if self.gan:                                         # the reference's draw: torch.randn_like(z) from the global generator, :329
            zin = torch.cat([zf, torch.randn_like(z).to(torch.float32)], dim=1).contiguous()
            Cin = 2 * Cz
        p0 = self._padded("p.in", n, H, W, Cin)
        check(lib().bd_tokens_to_padded(zin.data_ptr(), p0.data_ptr(), n, Cin, H, W, _st()), "bd_tokens_to_padded")
        c = self.conv_in.cout
        x = self._conv(self.conv_in, p0, self._get("s.in", (n, H, W, c), BF16), n, H, W)
        for i, blk in enumerate(self.mid):
            x = self._resblock(blk, x, n, H, W, f"mid{i & 1}")
        for lv in reversed(range(self.nlev)):
            L = self.levels[lv]
            c = x.shape[-1]
            # AdaptiveGroupNorm: scale / bias from the token statistics (tiny: torch), then scale * gn(x) + bias in fp32
            ada = L["ada"]
            with torch.autocast("cuda", dtype=BF16):
                flat = zf.flatten(2)
                scale = ada.gamma((flat.var(dim=-1) + ada.eps).sqrt())
                bias = ada.beta(flat.mean(dim=-1))
            st = self._stats(x, n, H, W, c, float(ada.eps))
            x = self._apply(x, st, self._get(f"s.ada", (n, H, W, c), torch.float32), 1, False, n, H, W, c,
                            scale=scale.float().contiguous(), bias=bias.float().contiguous())
            for i, blk in enumerate(L["blocks"]):
                x = self._resblock(blk, x, n, H, W, f"l{i & 1}")
            if lv > 0:
                c = x.shape[-1]
                px = self._apply(x, None, self._padded("p.up", n, H, W, c), 0, False, n, H, W, c)
                x = self._conv(L["up"], px, self._get("s.up", (n, 2 * H, 2 * W, c), BF16), n, H, W, mode=1)
                H, W = 2 * H, 2 * W
        c = x.shape[-1]
        pn = self._norm_swish_pad(self.norm_out, x, "p.a", n, H, W, c)
        img = torch.empty(n, self.conv_out.cout, H, W, dtype=torch.float32, device=self.device)
        self._conv(self.conv_out, pn, img, n, H, W, mode=2)
        return img.to(BF16)


class NativeEncoder(NativeDecoder):
    """``encode(x)`` == ``Encoder.forward(x)`` under ``torch.autocast('cuda', bfloat16)`` (autoencoder.py:59-127: conv_in, per level
    ResBlocks + a stride-2 3x3 convolution, mid blocks, norm_out -> swish -> 1x1 conv_out).  The dtype flow is simpler than the
    decoder's: no AdaptiveGroupNorm, so every convolution output and the residual stream are bf16, GroupNorm / swish fp32.
    The same kernels as the decoder (bd_conv.hip); H, W of the image must be multiples of 2^(levels - 1)."""

    def __init__(self, enc, device):
        self.device = torch.device(device)
        self.enc = enc
        self.nlev = enc.nlev
        C = lambda m, **k: _Conv(m, self.device, **k)
        self.conv_in = C(enc.conv_in, cin_pad=32)
        self.levels = []
        for lv in range(self.nlev):
            level = enc.down[lv]
            self.levels.append({"blocks": [self._block(b) for b in level.block],
                                "down": C(level.downsample) if lv < self.nlev - 1 else None})
        self.mid = [self._block(b) for b in enc.mid_block]
        self.norm_out = _Norm(enc.norm_out, self.device)
        self.conv_out = C(enc.conv_out)
        self._buf: dict = {}

    def decode(self, z):                                       # (only the operator helpers are inherited)
        raise BitDanceHipError("NativeEncoder has no decode()")

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, 3, H, W] image in [-1, 1] -> h [B, z_channels, H / 2^(levels-1), W / 2^(levels-1)] bf16 (pre-sign latent)."""
        if not x.is_cuda:
            raise BitDanceUnsupported("native encoder: CUDA/HIP tensors only (no CPU path)")
        n, cimg, H, W = x.shape
        f = 1 << (self.nlev - 1)
        if H % f or W % f or cimg > 32:
            raise BitDanceUnsupported(f"native encoder: image sides must be multiples of {f}")
        self._new_call(n, H, W)
        p0 = self._padded("p.img", n, H, W, 32)                # zero border AND zero channels 3 .. 31 (written once: stay zero)
        p0[:, 1:-1, 1:-1, :cimg] = x.permute(0, 2, 3, 1).to(BF16)           # the conv's input cast under autocast
        c = self.conv_in.cout
        h = self._conv(self.conv_in, p0, self._get("s.in", (n, H, W, c), BF16), n, H, W)
        for lv in range(self.nlev):
            L = self.levels[lv]
            for i, blk in enumerate(L["blocks"]):
                h = self._resblock(blk, h, n, H, W, f"l{i & 1}")
            if L["down"] is not None:
                c = h.shape[-1]
                ph = self._apply(h, None, self._padded("p.dn", n, H, W, c), 0, False, n, H, W, c)       # re-layout: zero border
                H, W = H // 2, W // 2
                h = self._conv(L["down"], ph, self._get("s.dn", (n, H, W, c), BF16), n, H, W)
        for i, blk in enumerate(self.mid):
            h = self._resblock(blk, h, n, H, W, f"mid{i & 1}")
        c = h.shape[-1]
        st = self._stats(h, n, H, W, c, self.norm_out.eps)
        a = self._apply(h, st, self._get("a.out", (n, H, W, c), BF16), 2, True, n, H, W, c, gamma=self.norm_out.gamma, beta=self.norm_out.beta)
        z = self._conv(self.conv_out, a, self._get("z", (n, H, W, self.conv_out.cout), BF16), n, H, W)
        return z.permute(0, 3, 1, 2).contiguous()
