"""The ten `sptr_cuda` functions (third_party/SparseTransformer/src/sptr/pointops_api.cpp:9-20) as C-ABI entries
u2mkd_sptr_<name> (csrc/sptr_ops.hip), called through ctypes with the reference launchers' argument lists and
layouts, against oracle.sptr_ops_ref.  precompute_all: BIT-EXACT against the reference's one known-answer fixture
(test/test_precompute_all.py:9-42, counts = [3,2,6]) and against the oracle on random windows; float ops <= 1e-5
relative (fp32 sums in a different order).  Fixture recipe = the reference's op tests
(test/test_relative_pos_encoding_op_step2.py:8-17: N = 3500, n = 150, hdim = 16, h = 6, L = 31, seed 2)."""
import pytest
import torch

from oracle import sptr_ops_ref as R

pytestmark = pytest.mark.gpu


def _dev(*ts):
    return [t.contiguous().cuda() for t in ts]


def _rel(a, b):
    b = b.double()
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-12))


def _windows(N, n, seed):
    g = torch.Generator().manual_seed(seed)
    cuts = torch.sort(torch.randperm(N - 1, generator=g)[:n - 1] + 1)[0]
    counts = torch.diff(torch.cat([torch.zeros(1, dtype=torch.long), cuts, torch.tensor([N])])).int()
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(-1)]).int()
    sq = torch.cat([counts.new_zeros(1), (counts ** 2).cumsum(-1)]).int()
    return counts, offsets, sq, g


def _precompute(hip, N, n, n_max, counts, offsets, sq):
    M = int(sq[-1])
    c, o, s = _dev(counts, offsets, sq)
    i0o = torch.zeros(N, dtype=torch.int32, device='cuda')
    i1o = torch.zeros(N, dtype=torch.int32, device='cuda')
    i0 = torch.zeros(M, dtype=torch.int32, device='cuda')
    i1 = torch.zeros(M, dtype=torch.int32, device='cuda')
    hip.call('u2mkd_sptr_precompute_all', N, n, n_max, hip.ptr(c), hip.ptr(o), hip.ptr(s), hip.ptr(i0o), hip.ptr(i1o),
             hip.ptr(i0), hip.ptr(i1), hip.stream())
    return i0o, i1o, i0, i1


def test_precompute_all_known_answer(hip):
    """counts = [3,2,6]: the expectations of test/test_precompute_all.py:29-42,65-70, bit for bit."""
    counts = torch.tensor([3, 2, 6], dtype=torch.int32)
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(-1)]).int()
    sq = torch.cat([counts.new_zeros(1), (counts ** 2).cumsum(-1)]).int()
    i0o, i1o, i0, i1 = (t.cpu() for t in _precompute(hip, 11, 3, 6, counts, offsets, sq))
    # the fixture's pure-torch expectations
    v2p = torch.tensor([1, 0, 0, 2, 0, 2, 2, 1, 2, 2, 2]).sort()[0]
    k = 6
    mask = torch.arange(k)[None].expand(3, -1) < counts[:, None]
    to_add = torch.arange(k)[None].expand(3, -1)[mask]
    ctg_index_1_offsets = torch.cat([torch.zeros(1, dtype=torch.long), (counts.long() ** 2).cumsum(-1)])[v2p] + to_add
    index_0_offsets = torch.cat([torch.zeros(1, dtype=torch.long), counts.long()[v2p].cumsum(-1)])
    assert torch.equal(torch.cat([i0o.long(), torch.tensor([49])]), index_0_offsets)   # functional.py:165 appends M
    assert torch.equal(i1o.long(), ctg_index_1_offsets)
    # SURVEY Appendix B-2's vector (== pointops.precompute_index_pairs on this fixture)
    assert i0.tolist() == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 2 + [4] * 2 + sum(([t] * 6 for t in range(5, 11)), [])
    assert i1.tolist() == [0, 1, 2] * 3 + [3, 4] * 2 + list(range(5, 11)) * 6


@pytest.mark.parametrize('N,n,seed', [(3500, 150, 2), (35000, 1500, 1), (700, 3, 5), (5, 5, 0), (1, 1, 0)])
def test_precompute_all_matches_the_oracle_bit_for_bit(hip, N, n, seed):
    counts, offsets, sq, _ = _windows(N, n, seed)
    got = _precompute(hip, N, n, int(counts.max()), counts, offsets, sq)
    ref = R.precompute_all(N, n, int(counts.max()), counts, offsets, sq)
    for a, b in zip(got, ref):
        assert torch.equal(a.cpu(), b)


@pytest.fixture(scope='module')
def fx(hip):
    N, n, h, d, L = 3500, 150, 6, 16, 31
    counts, offsets, sq, g = _windows(N, n, 2)
    i0o, i1o, i0, i1 = R.precompute_all(N, n, int(counts.max()), counts, offsets, sq)
    M = int(sq[-1])
    f = dict(N=N, n=n, M=M, h=h, d=d, L=L, n_max=int(counts.max()), i0=i0, i1=i1, i1o=i1o,
             i0o=torch.cat([i0o, torch.tensor([M], dtype=torch.int32)]))
    f['q'], f['k'], f['v'] = (torch.randn(N, h, d, generator=g) for _ in range(3))
    f['tq'], f['tk'], f['tv'] = (torch.randn(L, 3, h, d, generator=g) for _ in range(3))
    f['rel'] = torch.randint(0, L, (M, 3), generator=g).int()
    f['attn'] = torch.rand(M, h, generator=g)
    f['go_m'] = torch.randn(M, h, generator=g)
    f['go_n'] = torch.randn(N, h, d, generator=g)
    return f


def _t3(x):
    return x.permute(1, 2, 0).contiguous()          # [N,h,d] -> [h,d,N]


def _tt(x):
    return x.permute(2, 3, 1, 0).contiguous()       # [L,3,h,d] -> [h,d,3,L]


def test_scores_forward_three_forms(hip, fx):
    f = fx
    qt, kt, tqt, tkt, relt = _t3(f['q']), _t3(f['k']), _tt(f['tq']), _tt(f['tk']), f['rel'].t().contiguous()
    dq, dk, dtq, dtk, drel, di0, di1, di0o = _dev(qt, kt, tqt, tkt, relt, f['i0'], f['i1'], f['i0o'])
    N, M, h, d, L, nm = f['N'], f['M'], f['h'], f['d'], f['L'], f['n_max']
    out = torch.zeros(h, M, device='cuda')
    hip.call('u2mkd_sptr_attention_step1_forward', N, N, M, h, d, nm, hip.ptr(dq), hip.ptr(dk), hip.ptr(di0),
             hip.ptr(di1), hip.ptr(out), hip.stream())
    s1 = out.clone()
    assert _rel(s1, R.attention_step1_forward(qt, kt, f['i0'], f['i1'])) < 1e-5
    out.zero_()
    hip.call('u2mkd_sptr_dot_prod_with_idx_forward', N, M, h, d, nm, L, hip.ptr(dq), hip.ptr(di0), hip.ptr(di0o),
             hip.ptr(dk), hip.ptr(di1), hip.ptr(dtq), hip.ptr(dtk), hip.ptr(drel), hip.ptr(out), hip.stream())
    s2 = out.clone()
    assert _rel(s2, R.dot_prod_with_idx_forward(qt, f['i0'], kt, f['i1'], tqt, tkt, relt)) < 1e-5
    out.zero_()
    hip.call('u2mkd_sptr_dot_prod_with_idx_all_forward', N, M, h, d, nm, L, hip.ptr(dq), hip.ptr(di0), hip.ptr(di0o),
             hip.ptr(dk), hip.ptr(di1), hip.ptr(dtq), hip.ptr(dtk), hip.ptr(drel), hip.ptr(out), hip.stream())
    assert _rel(out, R.dot_prod_with_idx_all_forward(qt, f['i0'], kt, f['i1'], tqt, tkt, relt)) < 1e-5
    # the reference's own decomposition check (test_relative_pos_encoding_op_step1_all.py:87-89)
    assert _rel(out, (s1 + s2).cpu()) < 1e-5


def test_scores_backward(hip, fx):
    f = fx
    N, M, h, d, L, nm = f['N'], f['M'], f['h'], f['d'], f['L'], f['n_max']
    go, q, k, tq, tk, rel, i0, i1, i0o, i1o = _dev(f['go_m'], f['q'], f['k'], f['tq'], f['tk'], f['rel'], f['i0'],
                                                   f['i1'], f['i0o'], f['i1o'])
    gq, gk = torch.zeros_like(q), torch.zeros_like(k)
    gtq, gtk = torch.zeros_like(tq), torch.zeros_like(tk)
    hip.call('u2mkd_sptr_dot_prod_with_idx_backward', N, M, h, d, nm, L, hip.ptr(go), hip.ptr(q), hip.ptr(i0o),
             hip.ptr(k), hip.ptr(i1o), hip.ptr(i1), hip.ptr(tq), hip.ptr(tk), hip.ptr(rel), hip.ptr(gq), hip.ptr(gk),
             hip.ptr(gtq), hip.ptr(gtk), hip.stream())
    r = R.dot_prod_with_idx_backward(f['go_m'], f['q'], f['i0'], f['k'], f['i1'], f['tq'], f['tk'], f['rel'])
    for a, b in zip((gq, gk, gtq, gtk), r):
        assert _rel(a, b) < 1e-5
    gq2, gk2 = torch.zeros_like(q), torch.zeros_like(k)
    hip.call('u2mkd_sptr_attention_step1_backward', N, M, h, d, nm, hip.ptr(go), hip.ptr(i0), hip.ptr(i0o), hip.ptr(i1),
             hip.ptr(i1o), hip.ptr(q), hip.ptr(k), hip.ptr(gq2), hip.ptr(gk2), hip.stream())
    r2 = R.attention_step1_backward(f['go_m'], f['i0'], f['i1'], f['q'], f['k'])
    assert _rel(gq2, r2[0]) < 1e-5 and _rel(gk2, r2[1]) < 1e-5


def test_values_forward_and_backward(hip, fx):
    f = fx
    N, M, h, d, L, nm = f['N'], f['M'], f['h'], f['d'], f['L'], f['n_max']
    attn, v, tv, rel, i0, i1, i0o, i1o, go = _dev(f['attn'], f['v'], f['tv'], f['rel'], f['i0'], f['i1'], f['i0o'],
                                                  f['i1o'], f['go_n'])
    out = torch.zeros(N, h, d, device='cuda')
    hip.call('u2mkd_sptr_attention_step2_forward', N, M, h, d, nm, hip.ptr(attn), hip.ptr(v), hip.ptr(i0o), hip.ptr(i1),
             hip.ptr(out), hip.stream())
    assert _rel(out, R.attention_step2_forward(f['attn'], f['v'], f['i0'], f['i1'])) < 1e-5
    out.zero_()
    hip.call('u2mkd_sptr_attention_step2_with_rel_pos_value_forward', N, M, h, d, nm, hip.ptr(attn), hip.ptr(v),
             hip.ptr(i0o), hip.ptr(i1), hip.ptr(tv), hip.ptr(rel), hip.ptr(out), hip.stream())
    assert _rel(out, R.attention_step2_with_rel_pos_value_forward(f['attn'], f['v'], f['i0'], f['i1'], f['tv'],
                                                                  f['rel'])) < 1e-5
    # backward: the launcher takes v [h,d,N], table [h,d,3,L], rel_idx [3,M] (sptr/functional.py:387-389)
    vt, tvt, relt = _dev(_t3(f['v']), _tt(f['tv']), f['rel'].t())
    ga, gv, gt = torch.zeros(M, h, device='cuda'), torch.zeros(N, h, d, device='cuda'), torch.zeros_like(tv)
    hip.call('u2mkd_sptr_attention_step2_with_rel_pos_value_backward', N, M, h, d, L, nm, hip.ptr(go), hip.ptr(i0),
             hip.ptr(i0o), hip.ptr(i1), hip.ptr(i1o), hip.ptr(attn), hip.ptr(vt), hip.ptr(tvt), hip.ptr(relt),
             hip.ptr(ga), hip.ptr(gv), hip.ptr(gt), hip.stream())
    r = R.attention_step2_with_rel_pos_value_backward(f['go_n'], f['i0'], f['i1'], f['attn'], _t3(f['v']), _tt(f['tv']),
                                                      f['rel'].t())
    for a, b in zip((ga, gv, gt), r):
        assert _rel(a, b) < 1e-5
    ga.zero_()
    gv.zero_()
    hip.call('u2mkd_sptr_attention_step2_backward', N, M, h, d, nm, hip.ptr(go), hip.ptr(i0), hip.ptr(i0o), hip.ptr(i1),
             hip.ptr(i1o), hip.ptr(attn), hip.ptr(vt), hip.ptr(ga), hip.ptr(gv), hip.stream())
    r = R.attention_step2_backward(f['go_n'], f['i0'], f['i1'], f['attn'], _t3(f['v']))
    assert _rel(ga, r[0]) < 1e-5 and _rel(gv, r[1]) < 1e-5


def test_rejects_what_the_reference_asserts(hip):
    z = torch.zeros(8, device='cuda')
    zi = torch.zeros(8, dtype=torch.int32, device='cuda')
    with pytest.raises(RuntimeError, match='L <= 50'):
        hip.call('u2mkd_sptr_dot_prod_with_idx_forward', 1, 1, 1, 16, 1, 51, hip.ptr(z), hip.ptr(zi), hip.ptr(zi),
                 hip.ptr(z), hip.ptr(zi), hip.ptr(z), hip.ptr(z), hip.ptr(zi), hip.ptr(z), hip.stream())
