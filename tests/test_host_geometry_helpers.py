"""Host-side pieces of the geometry pre-pass and of the decoder's up-sampling that need no GPU: the deferred-size
unique (torch ops only), the one-round-trip count read, and the per-axis tap lists of the gathering bilinear backward
(numpy) against torch's own CPU arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize('n,hi', [(0, 5), (1, 5), (7, 1), (1000, 50), (5000, 2 ** 40)])
def test_unique_sorted_deferred_equals_torch_unique(n, hi):
    from u2mkd_amd.torchsparse.nn import functional as spf
    g = torch.Generator().manual_seed(n + 1)
    keys = torch.randint(-hi, hi, (n,), generator=g, dtype=torch.int64)
    buf, cnt = spf.unique_sorted_deferred(keys)
    assert buf.shape == keys.shape and cnt.dim() == 0 and cnt.dtype == torch.int64
    (count,) = spf.read_counts([cnt])
    want = torch.unique(keys)
    assert count == want.numel() and torch.equal(buf[:count], want)
    assert spf.read_counts([]) == []
    a, b = spf.read_counts([cnt, torch.tensor([3], dtype=torch.int32)])
    assert (a, b) == (count, 3)


def test_level_strides_follow_the_encoder_specs():
    from u2mkd_amd.lidar import point_voxel as PV
    assert PV._level_strides(PV.KMAP_SPECS) == [2, 4, 8, 16]
    assert PV._level_strides([(3, 1), (2, 2), (3, 1)]) == [2]


@pytest.mark.parametrize('n_in,n_out', [(2, 4), (8, 16), (12, 23), (29, 57), (50, 100), (7, 7), (1, 3), (90, 180)])
def test_up_taps_are_the_transpose_of_torchs_interpolation(n_in, n_out):
    from u2mkd_amd import camera
    scale, taps, wts, span = camera._up_taps(n_in, n_out, 'cpu')
    assert taps is not None and taps.shape == (n_in, 2) and wts.shape == (n_in, 8)
    # the dense interpolation matrix torch applies along one axis (align_corners=True), read off its CPU kernel
    eye = torch.eye(n_in, dtype=torch.float64).view(n_in, 1, 1, n_in)
    mat = F.interpolate(eye, (1, n_out), mode='bilinear', align_corners=True).view(n_in, n_out)      # [input, output]
    dense = torch.zeros(n_in, n_out, dtype=torch.float64)
    first, cnt = taps[:, 0].tolist(), taps[:, 1].tolist()
    for i in range(n_in):
        assert 1 <= cnt[i] <= 8 and first[i] + cnt[i] <= n_out
        dense[i, first[i]:first[i] + cnt[i]] = wts[i, :cnt[i]].double()
        assert float(wts[i, cnt[i]:].abs().sum()) == 0.0
    assert float((dense - mat).abs().max()) < 1e-5          # the taps follow the kernel's fp32 index arithmetic
    assert float((dense.sum(0) - 1).abs().max()) < 1e-6      # every output's weights sum to one
    assert abs(scale - (n_in - 1) / max(n_out - 1, 1)) < 1e-7
    # windows of the LDS form: monotone first outputs, the widest window of a tile covers its inputs' taps
    assert all(a <= b for a, b in zip(first[:-1], first[1:]))
    for t in (16, 64):
        widest = max(first[min(a + t, n_in) - 1] + cnt[min(a + t, n_in) - 1] - first[a] for a in range(0, n_in, t))
        assert span[t] == widest


def test_up_taps_refuse_large_factors():
    from u2mkd_amd import camera
    assert camera._up_taps(3, 64, 'cpu')[1] is None          # more than 8 outputs per input: torch's kernel serves those


def test_every_level_of_the_chain_can_be_taken_from_the_base_coordinates():
    """What spf.DownsamplePyramid relies on: level l of a chain of k = 2, s = 2 down-samplings (the oracle's spdownsample
    applied level by level, as torchsparse does) equals ONE down-sampling of the stride-1 coordinates by 2^l -- same set,
    same (b, x, y, z) order -- also for negative coordinates (floor division)."""
    from oracle import ts_ref as R
    rng = np.random.default_rng(3)
    base = np.concatenate([rng.integers(-300, 300, (4000, 3)), rng.integers(0, 3, (4000, 1))], 1).astype(np.int32)
    base = np.unique(base, axis=0)
    level, ts = base, 1
    for _ in range(4):
        level = R.spdownsample(level, 2, 2, ts)
        ts *= 2
        direct = R.spdownsample(base, ts, ts, 1)
        assert np.array_equal(level, direct), ts
        assert np.all(level[:, :3] % ts == 0)
