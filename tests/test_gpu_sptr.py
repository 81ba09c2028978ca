"""GPU parity of the fused window attention (rows a10-a11) vs oracle.sptr_ref:
window partition identical, forward <= 1e-5, all gradients incl. the three
relative-position tables <= 1e-4 (relative), cubic and spherical branches.
Fixture recipe follows the reference's op tests (third_party/SparseTransformer/test/
test_relative_pos_encoding_op_step2.py:8-17: hdim=16, h=6, L=31-ish tables)."""
import numpy as np
import pytest
import torch

from oracle import sptr_ref as S

pytestmark = pytest.mark.gpu


def _tokens(n, seed, batch=2, sphere=False):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([8.0, 8.0, 2.0])
    b = torch.sort(torch.randint(0, batch, (n,), generator=g))[0]
    if sphere:
        xyz = S.cart2sphere(xyz - torch.tensor([4.0, 4.0, 1.0]))
    return xyz, b


def _rel(a, b):
    a, b = a.double().cpu(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize('sphere,window,quant,h', [
    (False, [0.6, 0.6, 0.6], [0.025, 0.025, 0.025], 3),
    (False, [1.2, 1.2, 1.2], [0.05, 0.05, 0.05], 1),
    (True, [16.0, 16.0, 120.0], [16 / 24, 16 / 24, 5.0], 3),
    (True, [4.0, 4.0, 120.0], [16 / 12, 16 / 12, 5.0], 2),     # aliased quant size
    # heads per branch of the shipped configurations (spvcnn_spformer.py:70-83: C = cs[1..4], h = C / 16 split
    # half / half): cr 1.0 -> up to 8 per branch, cr 2.0 (the teacher, the `_B` student) -> 16 per branch at C = 512
    (False, [2.4, 2.4, 2.4], [0.1, 0.1, 0.1], 8),
    (False, [2.4, 2.4, 2.4], [0.1, 0.1, 0.1], 16),
    (True, [16.0, 16.0, 120.0], [16 / 12, 16 / 12, 5.0], 8),
    (True, [16.0, 16.0, 120.0], [16 / 12, 16 / 12, 5.0], 16),
])
def test_window_attention_fwd_bwd(hip, sphere, window, quant, h):
    from u2mkd_amd import sptr
    n, d, qgl = (3000 if h <= 3 else 1500), 16, 24
    L = 2 * qgl if sphere else 2 * qgl - 1
    a = 0.0125 if sphere else None
    xyz, b = _tokens(n, 5, sphere=sphere)
    g = torch.Generator().manual_seed(9)
    q, k, v = (torch.randn(n, h, d, generator=g) for _ in range(3))
    tq, tk, tv = (0.3 * torch.randn(L, 3, h, d, generator=g) for _ in range(3))
    go = torch.randn(n, h, d, generator=g)

    # oracle (fp32, CPU)
    qr, kr, vr, tqr, tkr, tvr = (t.clone().requires_grad_(True) for t in (q, k, v, tq, tk, tv))
    i0, i0o, n_max, i1, i1o, sort_idx = S.get_indices_params(xyz, b, np.array(window))
    out_r = S.sparse_self_attention(qr, kr, vr, xyz, i0, i0o, n_max, i1, i1o, sort_idx, np.array(window),
                                    np.array(quant), qgl, tqr, tkr, tvr, a)
    out_r.backward(go)

    # HIP
    qd, kd, vd, tqd, tkd, tvd = (t.clone().cuda().requires_grad_(True) for t in (q, k, v, tq, tk, tv))
    plan = sptr.WindowPlan(xyz.cuda(), b.cuda(), np.array(window))
    # same partition into windows (cluster ids compared as a partition, not as raw keys)
    c_ref = S.grid_cluster(xyz, b, np.array(window))
    order = plan.sort_idx.cpu().long()
    ws, wl = plan.wstart.cpu().long(), plan.wlen.cpu().long()
    assert torch.equal(c_ref[order], c_ref[order][ws])                       # one window = one cluster key
    counts = torch.unique(c_ref, return_counts=True)[1]
    assert sorted(wl[ws == torch.arange(n)].tolist()) == sorted(counts.tolist())
    assert plan.n_max == n_max

    out = sptr.window_attention(qd, kd, vd, xyz.cuda(), plan, np.array(quant), qgl, tqd, tkd, tvd, a)
    out.backward(go.cuda())
    assert _rel(out, out_r.detach()) < 1e-5
    for name, x, y in (('dq', qd, qr), ('dk', kd, kr), ('dv', vd, vr), ('dTq', tqd, tqr), ('dTk', tkd, tkr),
                       ('dTv', tvd, tvr)):
        assert _rel(x.grad, y.grad) < 1e-4, name


@pytest.mark.parametrize('sphere,window,quant,h,n', [
    (True, [16.0, 16.0, 120.0], [16 / 24, 16 / 24, 5.0], 3, 3000),       # large spherical windows: many full 16 x 16 tiles
    (True, [4.0, 4.0, 120.0], [16 / 12, 16 / 12, 5.0], 2, 3000),
    (True, [16.0, 16.0, 120.0], [16 / 12, 16 / 12, 5.0], 16, 1500),
    (False, [0.6, 0.6, 0.6], [0.025, 0.025, 0.025], 3, 3000),            # cubic windows of a few tokens: tiles mostly masked
    (False, [2.4, 2.4, 2.4], [0.1, 0.1, 0.1], 8, 1501),                  # (a token count that is not a multiple of 16)
    (True, [16.0, 16.0, 120.0], [16 / 24, 16 / 24, 5.0], 1, 7),          # fewer tokens than one tile
])
def test_tile_form_of_the_forward_equals_the_oracle_and_the_per_pair_kernels(hip, monkeypatch, sphere, window, quant, h, n):
    """csrc/sptr_tiles.hip (U2MKD_SPTR_TILES: 16 x 16 score tiles on v_mfma_f32_16x16x4_f32, relative-position terms from
    per-token strips by look-up, value tables through a per-query histogram) against the CPU oracle (sptr's dataflow,
    <= 1e-5) and against the per-pair kernels of csrc/sptr.hip on the same plan (outputs and log-sum-exp <= 1e-5; the
    backward, which reads the saved log-sum-exp, gives the same gradients <= 1e-5)."""
    from u2mkd_amd import sptr
    from u2mkd_amd.sptr import functional as SF
    d, qgl = 16, 24
    L = 2 * qgl if sphere else 2 * qgl - 1
    a = 0.0125 if sphere else None
    xyz, b = _tokens(n, 5, sphere=sphere)
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(n, h, d, generator=g) for _ in range(3))
    tq, tk, tv = (0.3 * torch.randn(L, 3, h, d, generator=g) for _ in range(3))
    go = torch.randn(n, h, d, generator=g)
    i0, i0o, n_max, i1, i1o, sort_idx = S.get_indices_params(xyz, b, np.array(window))
    ref = S.sparse_self_attention(q, k, v, xyz, i0, i0o, n_max, i1, i1o, sort_idx, np.array(window), np.array(quant), qgl, tq, tk, tv, a)
    res = {}
    for mode in ('0', 'all'):
        monkeypatch.setattr(SF, '_TILES', mode)
        leaves = [t.clone().cuda().requires_grad_(True) for t in (q, k, v, tq, tk, tv)]
        plan = sptr.WindowPlan(xyz.cuda(), b.cuda(), np.array(window))
        out = sptr.window_attention(*leaves[:3], xyz.cuda(), plan, np.array(quant), qgl, *leaves[3:], a)
        out.backward(go.cuda())
        res[mode] = [out.detach()] + [t.grad for t in leaves]
    assert _rel(res['all'][0].cpu(), ref) < 1e-5
    for x, y in zip(res['all'], res['0']):
        # (a gradient that is analytically zero -- single-token windows -- is exactly 0 on one path and rounding noise on the other)
        assert _rel(x, y.cpu()) < 1e-5 or float((x - y).abs().max()) < 1e-5


def test_reference_api_signature(hip):
    """The four names spherical_transformer.py:7 imports, called the way it calls them."""
    from functools import partial
    from u2mkd_amd import sptr
    n, h, d, qgl = 500, 2, 16, 24
    xyz, b = _tokens(n, 2)
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(n, h, d, generator=g) for _ in range(3))
    tabs = [0.2 * torch.randn(47, 3, h, d, generator=g) for _ in range(3)]
    window = sptr.to_3d_numpy([0.6, 0.6, 0.6])
    quant = sptr.to_3d_numpy([0.025, 0.025, 0.025])
    i0, i0o, n_max, i1, i1o, sort_idx = sptr.get_indices_params(xyz.cuda(), b.cuda(), window, False)
    out = sptr.sparse_self_attention(
        query=q.cuda(), key=k.cuda(), value=v.cuda(), xyz=xyz.cuda(), index_0=i0.int(), index_0_offsets=i0o.int(),
        n_max=n_max, index_1=i1.int(), index_1_offsets=i1o.int(), sort_idx=sort_idx, window_size=window,
        shift_win=False, pe_type='contextual', rel_query=True, rel_key=True, rel_value=True, quant_size=quant,
        quant_grid_length=qgl, relative_pos_query_table=tabs[0].cuda(), relative_pos_key_table=tabs[1].cuda(),
        relative_pos_value_table=tabs[2].cuda())
    r = S.get_indices_params(xyz, b, window)
    ref = S.sparse_self_attention(q, k, v, xyz, r[0], r[1], r[2], r[3], r[4], r[5], window, quant, qgl, *tabs)
    assert _rel(out, ref) < 1e-5
    t = sptr.SparseTrTensor(q, torch.cat([b[:, None].float(), xyz], 1), None, None)
    assert t.find_indice_params('x') is None


@pytest.mark.parametrize('window,quant', [([0.3, 0.3, 0.3], [0.0125, 0.0125, 0.0125]),
                                          ([2.4, 2.4, 2.4], [0.1, 0.1, 0.1]),
                                          ([2.0, 2.0, 120.0], [16 / 12, 16 / 12, 5.0])])
def test_quantiser_decisions_equal_on_equal_inputs(hip, window, quant):
    """The hard quantisers of the attention (window id = floor((p - min) / window), in-window coordinate =
    floor(((p - min) mod window) / quant)) are INTEGER decisions on fp32 inputs: fed identical fp32 coordinates, the
    HIP kernels must take exactly the decisions of the CPU restatement -- also for tokens ON a bin edge and one
    ulp either side of it (the adversarial half of this cloud).  What moves a token across an edge between the
    two paths is therefore never the quantiser, only a last-place difference in its INPUT (the atan2-derived
    angles of the spherical branch): the bounded allowance of tests/test_kd_path.py's first fixture."""
    from u2mkd_amd import sptr
    n = 4096
    g = torch.Generator().manual_seed(3)
    w = torch.tensor(window)
    qz = torch.tensor(quant)
    xyz = torch.rand(n, 3, generator=g) * w * 7.3
    lo = xyz.min(0)[0]
    # second half: tokens on window / quantisation edges and 1 ulp around them
    k = torch.randint(1, 7, (n // 2, 3), generator=g).float()
    j = torch.randint(0, 20, (n // 2, 3), generator=g).float()
    edge = lo + torch.where(torch.rand(n // 2, 3, generator=g) < 0.5, k * w, k * w + j * qz)
    ulp = torch.randint(-1, 2, (n // 2, 3), generator=g)
    edge = torch.where(ulp > 0, torch.nextafter(edge, edge + 1), torch.where(ulp < 0, torch.nextafter(edge, edge - 1), edge))
    xyz[n // 2:] = edge
    xyz[0] = lo                                                     # keep the minimum where it was
    b = torch.sort(torch.randint(0, 2, (n,), generator=g))[0]
    plan = sptr.WindowPlan(xyz.cuda(), b.cuda(), np.array(window))
    # window partition
    c_ref = S.grid_cluster(xyz, b, np.array(window))
    order = plan.sort_idx.cpu().long()
    ws = plan.wstart.cpu().long()
    assert torch.equal(c_ref[order], c_ref[order][ws])
    assert int((ws == torch.arange(n)).sum()) == len(torch.unique(c_ref))
    # in-window quantised coordinates (sptr/modules.py:40-41), sorted order
    qc, _, span = plan.quant_coords(xyz.cuda(), np.array(quant), False)
    want = torch.div((xyz - xyz.min(0)[0] + 0.0) % w, qz, rounding_mode='floor').int()
    assert torch.equal(qc.cpu(), want[order])
    assert int(want.max()) < span


@pytest.mark.parametrize('dim,heads', [(32, 2), (128, 8), (48, 3)])
def test_packed_attention_layer_equals_slice_scale_concat_form(hip, monkeypatch, dim, heads):
    """SparseMultiheadSASphereConcat on the packed qkv (row-strided kernels: q scale, head slices of the two branches
    and the concatenation inside the kernels) against the reference's dataflow around the contiguous kernels
    (qkv[:, 0] * scale, per-branch slices, torch.cat): the same arithmetic in the same order, so outputs and every
    gradient (input, qkv / proj parameters, the six tables) are EQUAL."""
    from u2mkd_amd.lidar import sphereformer as SF
    xyz, b = _tokens(3000, 5)
    xyz = (xyz - torch.tensor([4.0, 4.0, 1.0])).cuda()
    b = b.cuda().int()
    torch.manual_seed(dim)
    layer = SF.SparseMultiheadSASphereConcat(dim, heads, np.array([0.6, 0.6, 0.6]), np.array([1.5, 1.5, 80.0]),
                                             np.array([0.025, 0.025, 0.025]), np.array([0.0625, 0.0625, 3.4]), 0.0125).cuda()
    for n, p in layer.named_parameters():
        if 'table' in n:
            torch.nn.init.normal_(p, std=0.2)
    feats = torch.randn(3000, dim, device='cuda')
    g = torch.randn(3000, dim, device='cuda')
    res = {}
    for packed in (True, False):
        monkeypatch.setattr(SF, '_PACKED', packed)
        x = feats.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        y = layer(x, xyz, b)
        y.backward(g)
        res[packed] = [y.detach(), x.grad] + [p.grad.clone() for p in layer.parameters()]
    names = ['out', 'dx'] + [n for n, _ in layer.named_parameters()]
    for n, a, c in zip(names, res[True], res[False]):
        assert torch.equal(a, c), (n, float((a - c).abs().max()))


def test_window_plan_pair_equals_the_two_torch_built_plans(hip):
    """WindowPlan.pair (u2mkd_sptr_plan_prepare: spherical coordinates, bounds and both key arrays in two launches)
    against cart2sphere + two WindowPlan constructions out of torch operators: the spherical coordinates, the bounds,
    the sort permutations and the window ranges are EQUAL, bit for bit (the window a point falls into is a decision)."""
    from u2mkd_amd import sptr
    from u2mkd_amd.lidar.sphereformer import cart2sphere
    g = torch.Generator().manual_seed(5)
    for n, scale in ((54321, 50.0), (6000, 12.0), (77, 3.0), (1, 1.0)):
        xyz = ((torch.rand(n, 3, generator=g) - 0.5) * 2 * scale).cuda()
        xyz[:, 2] *= 0.1
        batch = (torch.rand(n, generator=g) < 0.4).long().cuda()
        wc, ws = np.array([0.6, 0.6, 0.6]), np.array([2.0, 2.0, 120.0])
        sph = cart2sphere(xyz)
        a, b = sptr.WindowPlan(xyz, batch, wc), sptr.WindowPlan(sph, batch, ws)
        p, q, sph2 = sptr.WindowPlan.pair(xyz, batch, wc, ws)
        assert torch.equal(sph, sph2), float((sph - sph2).abs().max())
        for ref, got in ((a, p), (b, q)):
            assert torch.equal(ref.lo, got.lo)
            assert torch.equal(ref.sort_idx, got.sort_idx) and torch.equal(ref.wstart, got.wstart) and torch.equal(ref.wlen, got.wlen)
            assert ref.window_size == got.window_size and ref.n == got.n
