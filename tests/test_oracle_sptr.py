"""Pins for oracle.sptr_ref: the reference's own known-answer fixture
(third_party/SparseTransformer/test/test_precompute_all.py:9-19, counts = [3,2,6])
and a brute-force dense per-window attention."""
import numpy as np
import torch

from oracle import sptr_ref as S


def test_precompute_all_known_answer():
    i0o, i1o, i0, i1 = S.precompute_all(np.array([3, 2, 6]))
    assert i0o.tolist() == [0, 3, 6, 9, 11, 13, 19, 25, 31, 37, 43, 49]
    assert i1o.tolist() == [0, 1, 2, 9, 10, 13, 14, 15, 16, 17, 18]
    assert i0.tolist() == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 2 + [4] * 2 + sum(([t] * 6 for t in range(5, 11)), [])
    assert i1.tolist() == [0, 1, 2] * 3 + [3, 4] * 2 + list(range(5, 11)) * 6
    # invariants of the fixture's pure-torch expectations (test_precompute_all.py:29-42)
    counts = np.array([3, 2, 6])
    v2p = np.repeat(np.arange(3), counts)
    assert (i0o[1:] - i0o[:-1]).tolist() == counts[v2p].tolist()
    sq = np.concatenate([[0], np.cumsum(counts ** 2)])
    rank = np.concatenate([np.arange(c) for c in counts])
    assert i1o.tolist() == (sq[v2p] + rank).tolist()


def _tokens(n=400, seed=0, batch=2):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([6.0, 6.0, 2.0])
    b = torch.sort(torch.randint(0, batch, (n,), generator=g))[0]
    return xyz, b


def test_grid_cluster_is_a_window_partition():
    xyz, b = _tokens()
    c = S.grid_cluster(xyz, b, [0.6, 0.6, 0.6])
    cell = torch.div(xyz - xyz.min(0)[0], 0.6, rounding_mode='floor').long()
    same_c = c[:, None] == c[None, :]
    same_cell = (cell[:, None] == cell[None, :]).all(-1) & (b[:, None] == b[None, :])
    assert torch.equal(same_c, same_cell)


def _run(split_a, window, quant, qgl, L, seed):
    xyz, b = _tokens(seed=seed)
    if split_a is not None:
        xyz = S.cart2sphere(xyz - torch.tensor([3.0, 3.0, 1.0]))
    h, d = 3, 16
    g = torch.Generator().manual_seed(seed + 1)
    q, k, v = (torch.randn(len(xyz), h, d, generator=g, dtype=torch.float64) for _ in range(3))
    tq, tk, tv = (0.2 * torch.randn(L, 3, h, d, generator=g, dtype=torch.float64) for _ in range(3))
    i0, i0o, n_max, i1, i1o, sort_idx = S.get_indices_params(xyz, b, np.array(window))
    out = S.sparse_self_attention(q, k, v, xyz.double(), i0, i0o, n_max, i1, i1o, sort_idx, np.array(window),
                                  np.array(quant), qgl, tq, tk, tv, split_a)
    ref = S.dense_window_attention(q, k, v, xyz.double(), b, np.array(window), np.array(quant), qgl, tq, tk, tv, split_a)
    return out, ref, n_max


def test_attention_cubic_matches_dense():
    out, ref, n_max = _run(None, [0.6, 0.6, 0.6], [0.025, 0.025, 0.025], 24, 47, 3)
    assert n_max > 1
    assert torch.allclose(out, ref, atol=1e-10)


def test_attention_sphere_matches_dense():
    out, ref, n_max = _run(0.0125, [20.0, 20.0, 120.0], [20 / 24, 20 / 24, 5.0], 24, 48, 4)
    assert n_max > 4
    assert torch.allclose(out, ref, atol=1e-10)
