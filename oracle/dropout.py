"""Oracle for the counter-based dropout of the HIP training path -- TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference draws its dropout masks from torch's generator (F.dropout, gnn_model.py:274; nn.Dropout, :46), a
stream no other implementation can reproduce; what parity can hold is (a) the arithmetic given a mask --
``oracle.model.gossip_single_query(layer_masks=, post_mask=)`` -- and (b) that the mask the kernels use is the
documented function of (seed, step, site, row, col).  This file restates (b) in numpy:

    bits(row, col) = Philox4x32-10(counter = {row >> 2, col | site << 24, lo32(step), hi32(step)},
                                   key = {lo32(seed), hi32(seed)})[row & 3]
    factor = 0 if bits < round(p 2^32) else 1 / (1 - p)

Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11), the Random123
library's round function, multipliers 0xD2511F53 / 0xCD9E8D57 and Weyl key increments 0x9E3779B9 / 0xBB67AE85; pinned
by Random123's published known-answer vectors (tests/test_oracle_dropout.py).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: four uint32 arrays (broadcastable), key: two uint32 scalars/arrays -> four uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & _MASK for c in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(10):
        p0 = M0 * c0                       # 32 x 32 -> 64 bit products
        p1 = M1 * c2
        h0, l0 = p0 >> np.uint64(32), p0 & _MASK
        h1, l1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = h1 ^ c1 ^ np.uint64(k0), l1, h0 ^ c3 ^ np.uint64(k1), l0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def dropout_bits(seed: int, step: int, site: int, num_rows: int, num_cols: int) -> np.ndarray:
    """[num_rows, num_cols] uint32: the random word of every element."""
    r4 = (np.arange(num_rows, dtype=np.uint64) >> np.uint64(2))[:, None]
    col = (np.arange(num_cols, dtype=np.uint64) | np.uint64(site << 24))[None, :]
    w = philox4x32_10((r4, col, np.uint64(step & 0xFFFFFFFF), np.uint64((step >> 32) & 0xFFFFFFFF)),
                      (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    sel = (np.arange(num_rows) & 3)[:, None]
    out = np.where(sel == 0, w[0], np.where(sel == 1, w[1], np.where(sel == 2, w[2], w[3])))
    return out.astype(np.uint32)


def dropout_factor(seed: int, step: int, site: int, p: float, num_rows: int, num_cols: int) -> np.ndarray:
    """[num_rows, num_cols] float32: 0 where the element is dropped, 1 / (1 - p) where it is kept."""
    if p >= 1.0:
        return np.zeros((num_rows, num_cols), np.float32)
    thr = min(int(round(p * 4294967296.0)), 0xFFFFFFFF)
    bits = dropout_bits(seed, step, site, num_rows, num_cols)
    return np.where(bits < np.uint32(thr), np.float32(0.0), np.float32(1.0 / (1.0 - p))).astype(np.float32)
