"""Oracle: bit <-> index math of the group-wise lookup-free quantiser (test infrastructure only).

Restates /root/reference/imagenet_gen/src/gfq.py:
  indices_to_bits :152-160   bits_to_indices :173-187   codebook :144-147
  forward index computation :221-239 (sign-quantise, then sum_k (q>0) * 2^k per codebook)
Integer work: numpy, bit exact.
"""
from __future__ import annotations

import numpy as np


def indices_to_bits(idx: np.ndarray, nbits: int) -> np.ndarray:
    """:152-160 -- bit k of the index at position k (LSB first)."""
    mask = (1 << np.arange(nbits, dtype=np.int64))
    return (idx[..., None].astype(np.int64) & mask) != 0


def bits_to_indices(bits: np.ndarray) -> np.ndarray:
    """:173-187."""
    w = (1 << np.arange(bits.shape[-1], dtype=np.int64))
    return (bits.astype(np.int64) * w).sum(-1)


def codes_from_indices(idx: np.ndarray, nbits: int) -> np.ndarray:
    """:144-147 codebook = bits*2-1 as float32."""
    return indices_to_bits(idx, nbits).astype(np.float32) * 2.0 - 1.0


def quantize_to_indices(z: np.ndarray, num_codebooks: int) -> tuple[np.ndarray, np.ndarray]:
    """:217-239.  z [..., C] float -> (quantized +-1 [..., C], indices [..., num_codebooks])."""
    q = np.where(z > 0, np.float32(1.0), np.float32(-1.0))
    d = z.shape[-1] // num_codebooks
    bits = (q > 0).reshape(*z.shape[:-1], num_codebooks, d)
    return q, bits_to_indices(bits)
