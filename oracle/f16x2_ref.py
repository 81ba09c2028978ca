"""CPU model of the f16x2 arithmetic the HIP conv kernels multiply in (csrc/conv_internal.h, conv_tp.hip AR = 4, conv_px3.hip F2).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): this is NOT a restatement of reference code -- the reference multiplies in
fp32 (torchsparse v1.4.0: torch.mm per kernel offset) -- but a numpy model of the product's OWN
arithmetic, so that its accuracy claim can be checked without a GPU: every fp32 operand is scaled by a power of two and split
into two fp16 planes, ``x s = h + l`` with ``h = fp16(x s)``, ``l = fp16(x s - h)`` (round to nearest even, as
``v_cvt_pk_f16_f32`` / ``v_fma_mixlo_f16`` do), and ``a b`` is taken as ``hh + hl + lh`` accumulated in fp32; the scales -- per
gathered row (or per 32-channel step of a row) and per weight tensor -- put the largest |x| of the scaled set into
[2^14, 2^15) and are taken out exactly.  The accumulation ORDER of the matrix instruction is not modelled (numpy sums in its
own order): the model is held to the same bound as the kernels, not to their bits."""
from __future__ import annotations

import numpy as np

__all__ = ['pow2_scale', 'split_planes', 'matmul_f16x2']


def pow2_scale(m):
    """(s, 1/s) with s the power of two that puts m >= 0 into [2^14, 2^15); exponents clamped as f16x2_scale does
    (conv_internal.h): biased exponent e of m in [15, 253], s = 2^(141 - e)."""
    m = np.asarray(m, dtype=np.float32)
    e = (m.view(np.uint32) >> 23).astype(np.int64)
    e = np.clip(e, 15, 253)
    s = np.ldexp(np.float32(1), (141 - e).astype(np.int32)).astype(np.float32)
    inv = np.ldexp(np.float32(1), (e - 141).astype(np.int32)).astype(np.float32)
    return s, inv


def split_planes(x_scaled):
    """h = fp16(x), l = fp16(x - h) (both round to nearest even), returned as float32 arrays of the fp16 values."""
    x_scaled = np.asarray(x_scaled, dtype=np.float32)
    h = x_scaled.astype(np.float16)
    l = (x_scaled - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)


def matmul_f16x2(x, w, step=None):
    """x [n, cin] @ w [cin, cout] in f16x2 arithmetic, fp32 accumulate.  ``step`` = channels per scaled segment of a row
    (None: the whole row carries one scale -- the tile kernel; 32: every 32-channel step its own -- the pair kernel)."""
    x = np.asarray(x, dtype=np.float32)
    w = np.asarray(w, dtype=np.float32)
    n, cin = x.shape
    sw, iw = pow2_scale(np.abs(w).max() if w.size else np.float32(0))
    wh, wl = split_planes(w * sw)
    out = np.zeros((n, w.shape[1]), dtype=np.float32)
    step = cin if step is None else step
    for c0 in range(0, cin, step):
        xs = x[:, c0:c0 + step]
        sr, ir = pow2_scale(np.abs(xs).max(axis=1) if xs.size else np.zeros(n, np.float32))
        xh, xl = split_planes(xs * sr[:, None])
        acc = (xh @ wl[c0:c0 + step] + xl @ wh[c0:c0 + step] + xh @ wh[c0:c0 + step]).astype(np.float32)
        out += acc * (ir * iw)[:, None]
    return out
