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


# ---- oracle.sptr_ops_ref: the ten sptr_cuda operator kernels in the launchers' layouts ---------------------

def _op_fixture(seed=2, N=70, n=9, h=3, d=16, L=31, dtype=torch.float64):
    """The recipe of the reference's op tests (third_party/SparseTransformer/test/
    test_relative_pos_encoding_op_step2.py:8-35: random window sizes, hdim 16, L 31), scaled down."""
    from oracle import sptr_ops_ref as R
    g = torch.Generator().manual_seed(seed)
    cuts = torch.sort(torch.randperm(N - 1, generator=g)[:n - 1] + 1)[0]
    counts = torch.diff(torch.cat([torch.zeros(1, dtype=torch.long), cuts, torch.tensor([N])])).int()
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(-1)]).int()
    sq_offsets = torch.cat([counts.new_zeros(1), (counts ** 2).cumsum(-1)]).int()
    i0o, i1o, i0, i1 = R.precompute_all(N, n, int(counts.max()), counts, offsets, sq_offsets)
    M = int(sq_offsets[-1])
    q, k, v = (torch.randn(N, h, d, generator=g, dtype=dtype) for _ in range(3))
    tq, tk, tv = (torch.randn(L, 3, h, d, generator=g, dtype=dtype) for _ in range(3))
    rel = torch.randint(0, L, (M, 3), generator=g).int()
    attn = torch.rand(M, h, generator=g, dtype=dtype)
    return dict(N=N, n=n, M=M, h=h, d=d, L=L, counts=counts, offsets=offsets, sq_offsets=sq_offsets, i0o=i0o, i1o=i1o,
                i0=i0, i1=i1, q=q, k=k, v=v, tq=tq, tk=tk, tv=tv, rel=rel, attn=attn, g=g)


def test_ops_precompute_all_known_answer_in_launcher_form():
    """test/test_precompute_all.py:9-42,65-70 through the launcher-shaped restatement."""
    from oracle import sptr_ops_ref as R
    counts = torch.tensor([3, 2, 6], dtype=torch.int32)
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(-1)]).int()
    sq = torch.cat([counts.new_zeros(1), (counts ** 2).cumsum(-1)]).int()
    i0o, i1o, i0, i1 = R.precompute_all(11, 3, 6, counts, offsets, sq)
    v2p = torch.repeat_interleave(torch.arange(3), counts.long())
    assert torch.equal(torch.cat([i0o.long(), torch.tensor([49])]),
                       torch.cat([torch.zeros(1, dtype=torch.long), counts.long()[v2p].cumsum(-1)]))
    to_add = torch.cat([torch.arange(c) for c in counts.tolist()])
    assert torch.equal(i1o.long(), sq.long()[v2p] + to_add)
    assert i0.dtype == torch.int32 and len(i0) == 49 and len(i1) == 49


def test_ops_decomposition_identity():
    """test_relative_pos_encoding_op_step1_all.py:87-89: all == dot_prod_with_idx + attention_step1."""
    from oracle import sptr_ops_ref as R
    f = _op_fixture()
    qt, kt = f['q'].permute(1, 2, 0).contiguous(), f['k'].permute(1, 2, 0).contiguous()
    tqt, tkt = f['tq'].permute(2, 3, 1, 0).contiguous(), f['tk'].permute(2, 3, 1, 0).contiguous()
    relt = f['rel'].t().contiguous()
    a = R.dot_prod_with_idx_all_forward(qt, f['i0'], kt, f['i1'], tqt, tkt, relt)
    b = R.dot_prod_with_idx_forward(qt, f['i0'], kt, f['i1'], tqt, tkt, relt) + R.attention_step1_forward(qt, kt, f['i0'], f['i1'])
    assert torch.allclose(a, b, atol=1e-12)


def test_ops_backwards_are_the_derivatives_of_the_forwards():
    from oracle import sptr_ops_ref as R
    f = _op_fixture()
    i0, i1, rel = f['i0'], f['i1'], f['rel']
    q, k, v, tq, tk, tv, attn = (f[x].clone().requires_grad_(True) for x in ('q', 'k', 'v', 'tq', 'tk', 'tv', 'attn'))
    go = torch.randn(f['M'], f['h'], generator=f['g'], dtype=torch.float64)
    # scores (the `all` form = rpe backward + step-1 backward, sptr/functional.py:317-325)
    s = R.dot_prod_with_idx_all_forward(q.permute(1, 2, 0), i0, k.permute(1, 2, 0), i1, tq.permute(2, 3, 1, 0),
                                        tk.permute(2, 3, 1, 0), rel.t())
    s.t().backward(go)
    gq, gk, gtq, gtk = R.dot_prod_with_idx_backward(go, f['q'], i0, f['k'], i1, f['tq'], f['tk'], rel)
    gq2, gk2 = R.attention_step1_backward(go, i0, i1, f['q'], f['k'])
    assert torch.allclose(gq + gq2, q.grad, atol=1e-10) and torch.allclose(gk + gk2, k.grad, atol=1e-10)
    assert torch.allclose(gtq, tq.grad, atol=1e-10) and torch.allclose(gtk, tk.grad, atol=1e-10)
    # values with tables
    go2 = torch.randn(f['N'], f['h'], f['d'], generator=f['g'], dtype=torch.float64)
    R.attention_step2_with_rel_pos_value_forward(attn, v, i0, i1, tv, rel).backward(go2)
    ga, gv, gt = R.attention_step2_with_rel_pos_value_backward(go2, i0, i1, f['attn'], f['v'].permute(1, 2, 0),
                                                               f['tv'].permute(2, 3, 1, 0), rel.t())
    assert torch.allclose(ga, attn.grad, atol=1e-10) and torch.allclose(gv, v.grad, atol=1e-10)
    assert torch.allclose(gt, tv.grad, atol=1e-10)
    # values without tables
    attn.grad = None
    v.grad = None
    R.attention_step2_forward(attn, v, i0, i1).backward(go2)
    ga, gv = R.attention_step2_backward(go2, i0, i1, f['attn'], f['v'].permute(1, 2, 0))
    assert torch.allclose(ga, attn.grad, atol=1e-10) and torch.allclose(gv, v.grad, atol=1e-10)


def test_ops_compose_to_the_fused_attention():
    """The five operators the reference's modules.py:52-62 chains == oracle.sptr_ref.sparse_self_attention."""
    from oracle import sptr_ops_ref as R
    xyz, b = _tokens(seed=6)
    h, d, L, qgl = 2, 16, 47, 24
    g = torch.Generator().manual_seed(8)
    q, k, v = (torch.randn(len(xyz), h, d, generator=g, dtype=torch.float64) for _ in range(3))
    tq, tk, tv = (0.2 * torch.randn(L, 3, h, d, generator=g, dtype=torch.float64) for _ in range(3))
    window, quant = np.array([0.6] * 3), np.array([0.025] * 3)
    i0, i0o, n_max, i1, i1o, sort_idx = S.get_indices_params(xyz, b, window)
    ref = S.sparse_self_attention(q, k, v, xyz.double(), i0, i0o, n_max, i1, i1o, sort_idx, window, quant, qgl, tq, tk, tv)
    qs, ks, vs = q[sort_idx], k[sort_idx], v[sort_idx]
    rel = S.relative_position_index(xyz.double()[sort_idx], i0, i1, window, quant, qgl)
    a = R.dot_prod_with_idx_all_forward(qs.permute(1, 2, 0), i0, ks.permute(1, 2, 0), i1, tq.permute(2, 3, 1, 0),
                                        tk.permute(2, 3, 1, 0), rel.t()).t()
    p = S._segment_softmax(a, i0o)
    x = R.attention_step2_with_rel_pos_value_forward(p, vs, i0, i1, tv, rel)
    out = torch.empty_like(x)
    out[sort_idx] = x
    assert torch.allclose(out, ref, atol=1e-10)


def test_python_layer_over_the_operator_backend_equals_the_fused_oracle():
    """oracle.sptr_layer_ref (sptr/functional.py's autograd Functions + modules.py:11-66 + utils.py:49-95 restated over a
    `sptr_cuda`-shaped backend) over the CPU operator oracle == oracle.sptr_ref.sparse_self_attention (the formulation the
    fused kernel is held to), forward and every gradient; cubic and spherical (exponential split) branches."""
    from functools import partial
    from oracle import sptr_layer_ref as P
    lay = P.layer(P.CpuBackend())
    for split_a, window, quant, qgl, L in ((None, [0.6, 0.6, 0.6], [0.025, 0.025, 0.025], 24, 47),
                                           (0.0125, [2.0, 2.0, 120.0], [2 / 24, 2 / 24, 5.0], 24, 48)):
        xyz, b = _tokens(seed=5)
        if split_a is not None:
            xyz = S.cart2sphere(xyz - torch.tensor([3.0, 3.0, 1.0]))
        h, d = 3, 16
        g = torch.Generator().manual_seed(11)
        q, k, v = (torch.randn(len(xyz), h, d, generator=g) for _ in range(3))
        tq, tk, tv = (0.2 * torch.randn(L, 3, h, d, generator=g) for _ in range(3))
        window, quant = np.array(window, dtype=np.float32), np.array(quant, dtype=np.float32)
        # index structures: the layer's own (precompute_all through the backend) equal the oracle's
        i0, i0o, n_max, i1, i1o, sort_idx = lay.get_indices_params(xyz, b, window, False)
        r0, r0o, rn, r1, r1o, rsort = S.get_indices_params(xyz, b, window)
        assert n_max == rn and torch.equal(i0, r0) and torch.equal(i1, r1) and torch.equal(i0o.long(), r0o) \
            and torch.equal(i1o.long(), r1o) and torch.equal(sort_idx, rsort)
        leaves = [t.clone().requires_grad_(True) for t in (q, k, v, tq, tk, tv)]
        out = lay.sparse_self_attention(leaves[0], leaves[1], leaves[2], xyz, i0.int(), i0o.int(), n_max, i1.int(), i1o.int(), sort_idx,
                                        window, False, pe_type='contextual', rel_query=True, rel_key=True, rel_value=True,
                                        quant_size=quant, quant_grid_length=qgl, relative_pos_query_table=leaves[3],
                                        relative_pos_key_table=leaves[4], relative_pos_value_table=leaves[5],
                                        split_func=None if split_a is None else partial(S.exponential_split, a=split_a))
        ref_leaves = [t.clone().requires_grad_(True) for t in (q, k, v, tq, tk, tv)]
        ref = S.sparse_self_attention(ref_leaves[0], ref_leaves[1], ref_leaves[2], xyz, r0, r0o, rn, r1, r1o, rsort, window, quant,
                                      qgl, ref_leaves[3], ref_leaves[4], ref_leaves[5], split_a)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
        go = torch.randn(out.shape, generator=g)
        out.backward(go)
        ref.backward(go)
        for a, b_ in zip(leaves, ref_leaves):
            assert torch.allclose(a.grad, b_.grad, rtol=1e-4, atol=1e-5 * float(b_.grad.abs().max()) + 1e-6)
