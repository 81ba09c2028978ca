"""The f16x2 arithmetic of the conv kernels, modelled in numpy (oracle/f16x2_ref.py), against float64 on the CPU: the bound the
GPU tests hold the kernels to (tests/test_gpu_conv_f16x2.py: |error| <= 2^-20 sum |x||w|) is a property of the arithmetic
itself -- whatever the magnitudes, because of the power-of-two scales -- and a plain fp16 product is not in that class."""
import numpy as np
import pytest

from oracle import f16x2_ref as R


def _case(spread, n=600, cin=128, cout=64, seed=0):
    g = np.random.default_rng(seed)
    x = g.standard_normal((n, cin)).astype(np.float32)
    w = (g.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    if spread == 'rows':          # every row its own magnitude, 1e-18 .. 1e18; zero rows; rows with one entry
        x *= (10.0 ** g.integers(-18, 19, (n, 1))).astype(np.float32)
        x[g.random(n) < 0.1] = 0
        one = g.random(n) < 0.05
        x[one, 1:] = 0
    elif spread == 'tiny':
        x *= np.float32(1e-20)
        w *= np.float32(1e-6)
    elif spread == 'huge':        # beyond fp16's 65504 on both sides
        x *= np.float32(3e7)
        w *= np.float32(1e9)
    elif spread == 'segments':    # magnitudes that differ ALONG a row: per 32-channel step
        x *= np.repeat((10.0 ** g.integers(-6, 7, (n, cin // 32))).astype(np.float32), 32, axis=1)
    return x, w


@pytest.mark.parametrize('spread', ['unit', 'rows', 'tiny', 'huge', 'segments'])
@pytest.mark.parametrize('step', [None, 32])
def test_f16x2_product_is_an_fp32_grade_product_at_any_magnitude(spread, step):
    x, w = _case(spread)
    got = R.matmul_f16x2(x, w, step).astype(np.float64)
    want = x.astype(np.float64) @ w.astype(np.float64)
    mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    assert (err <= mag * 2.0 ** -20 + 1e-300).all(), float((err / (mag + 1e-300)).max())
    # and it is in the class of the fp32 product itself (numpy's float32 matmul of the same operands)
    f32 = np.abs((x @ w).astype(np.float64) - want)
    assert err.max() <= 16 * f32.max() + 1e-300


def test_the_scale_is_what_makes_it_so():
    """Without the scales the same two-plane product overflows (huge) or loses its low plane to fp16's subnormals (tiny)."""
    for spread in ('huge', 'tiny'):
        x, w = _case(spread)
        with np.errstate(over='ignore', invalid='ignore'):
            xh, xl = R.split_planes(x)
            wh, wl = R.split_planes(w)
            got = (xh @ wl + xl @ wh + xh @ wh).astype(np.float64)
        want = x.astype(np.float64) @ w.astype(np.float64)
        mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)
        bad = ~np.isfinite(got) | (np.abs(got - want) > mag * 2.0 ** -20)
        assert bad.any(), spread


def test_pow2_scale_puts_the_maximum_into_the_fp16_sweet_spot():
    m = np.float32([0.0, 1e-38, 3e-20, 1.0, 65504.0, 7e9, 3e38])
    s, inv = R.pow2_scale(m)
    assert (s * inv == 1).all()
    scaled = m * s
    ok = (scaled >= 2.0 ** 14) & (scaled < 2.0 ** 15)
    assert ok[2:6].all() and scaled[0] == 0           # (0 and the near-subnormal get the clamped scale ...
    assert 2.0 ** 15 <= scaled[6] < 65504             # ... and so does the top binade: still inside fp16's range)
    assert (np.log2(s) == np.round(np.log2(s))).all()
