"""GPU parity: every HIP operator behind the C ABI vs oracle.ts_ref on the same
seeded inputs.  Integer / index results bit-exact; fp32 results within the
stated tolerance (north-star: 1e-3 abs on logits; ops are held to 1e-4 rel)."""
import numpy as np
import pytest
import torch

from oracle import ts_ref as R
from u2mkd_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F(hip):
    from u2mkd_amd.torchsparse.nn import functional as F
    return F


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _rel(a, b):
    a = a.double().cpu()
    b = b.double().cpu() if isinstance(b, torch.Tensor) else torch.from_numpy(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _scene(n=3000, batch=2, seed=7):
    b = synth_batch(n, batch, seed)
    return b['coords'], b['feats']


def test_hash_bit_exact(F):
    rng = np.random.default_rng(0)
    c = rng.integers(-500, 2000, (5000, 4)).astype(np.int32)
    c[:, 3] = rng.integers(0, 4, 5000)
    got = F.sphash(_dev(c)).cpu().numpy()
    assert (got == R.sphash(c)).all()
    for size, stride in ((3, 1), (2, 4)):
        off = R.get_kernel_offsets(size, stride)
        got = F.sphash(_dev(c), _dev(off)).cpu().numpy()
        assert got.shape == (len(off), len(c))
        assert (got == R.sphash(c, off)).all()
    assert F.sphash(torch.zeros(0, 4, dtype=torch.int32, device='cuda')).shape == (0,)


def test_hashquery_miss_duplicate_shape(F):
    rng = np.random.default_rng(1)
    ref = rng.integers(0, 1 << 59, 20000).astype(np.int64)
    ref[100] = ref[5]          # duplicate: smallest index wins
    ref[19999] = ref[5]
    q = np.concatenate([ref[rng.integers(0, 20000, 30000)], rng.integers(0, 1 << 59, 10000)]).astype(np.int64)
    q = q.reshape(8, 5000)
    got = F.sphashquery(_dev(q), _dev(ref)).cpu().numpy()
    want = R.sphashquery(q, ref)
    assert got.shape == q.shape and (got == want).all()
    assert (got == -1).sum() > 0
    # empty inputs
    assert F.sphashquery(torch.zeros(0, dtype=torch.int64, device='cuda'), _dev(ref)).numel() == 0
    got = F.sphashquery(_dev(q[0]), torch.zeros(0, dtype=torch.int64, device='cuda'))
    assert (got.cpu().numpy() == -1).all()


def test_count_voxelize_devoxelize(F):
    rng = np.random.default_rng(2)
    n, nv = 20000, 6000
    idx = rng.integers(0, nv, n).astype(np.int32)
    idx[::31] = -1
    counts = F.spcount(_dev(idx), nv)
    assert (counts.cpu().numpy() == R.spcount(idx, nv)).all()
    for c in (4, 32, 96, 3):
        f = torch.randn(n, c)
        fd = f.cuda().requires_grad_(True)
        out = F.spvoxelize(fd, _dev(idx), counts)
        want = R.voxelize_forward(f, idx, counts.cpu())
        assert _rel(out, want) < 1e-5
        g = torch.randn(nv, c)
        out.backward(g.cuda())
        assert _rel(fd.grad, R.voxelize_backward(g, idx, counts.cpu(), n)) < 1e-6
    # devoxelize
    idx8 = rng.integers(-1, nv, (n, 8)).astype(np.int32)
    w8 = torch.rand(n, 8)
    for c in (4, 48, 128, 6):
        f = torch.randn(nv, c)
        fd = f.cuda().requires_grad_(True)
        out = F.spdevoxelize(fd, _dev(idx8), w8.cuda())
        assert _rel(out, R.devoxelize_forward(f, idx8, w8)) < 1e-5
        g = torch.randn(n, c)
        out.backward(g.cuda())
        assert _rel(fd.grad, R.devoxelize_backward(g, idx8, w8, nv)) < 1e-5


def test_ti_weights(F):
    coords, _ = _scene()
    pts = torch.from_numpy(coords).float()
    pts[:, :3] += torch.rand(len(pts), 3) * 0.999
    for s in (1, 2, 8):
        vox = coords.copy()
        vox[:, :3] = vox[:, :3] // s * s
        vox = np.unique(vox, axis=0).astype(np.int32)
        base = torch.cat([torch.floor(pts[:, :3] / s).int() * s, pts[:, 3:].int()], 1).numpy()
        q = R.sphash(base, R.get_kernel_offsets(2, s))
        idx = R.sphashquery(q, R.sphash(vox))
        idx[3, ::5] = -1
        want = R.calc_ti_weights(pts, idx, s)
        got = F.calc_ti_weights(pts.cuda(), _dev(idx), s)
        assert got.shape == want.shape
        assert float((got.cpu() - want).abs().max()) < 2e-6
        w8, i8 = F.ti_weights_n8(pts.cuda(), _dev(idx), s)
        assert (i8.cpu().numpy() == idx.T).all()


def test_downsample_and_kmap_bit_exact(F):
    coords, _ = _scene(4000, 2)
    cd = _dev(coords)
    for ts_, ks, st in ((1, 3, 1), (1, 2, 2), (2, 3, 1), (2, 2, 2), (4, 2, 2)):
        c = coords.copy()
        c[:, :3] = c[:, :3] // ts_ * ts_
        c = np.unique(c[:, [3, 0, 1, 2]], axis=0)[:, [1, 2, 3, 0]].astype(np.int32)
        rng = np.random.default_rng(3)
        c = np.ascontiguousarray(c[rng.permutation(len(c))])
        nbmaps, nbsizes, oc, results = R.build_kmap(c, ts_, ks, st)
        km = F.build_kmap(_dev(c), (ts_,) * 3, (ks,) * 3, (st,) * 3)
        assert (km.out_coords.cpu().numpy() == oc).all()
        assert (km.nbr.cpu().numpy() == results).all()
        got_maps, got_sizes, sizes = km[0], km[1], km[2]
        assert sizes == (len(c), len(oc))
        assert (got_sizes.cpu().numpy() == nbsizes).all()
        assert (got_maps.cpu().numpy() == nbmaps).all()
        if km.nbr_inv is not None:
            inv = np.full((results.shape[0], len(c)), -1, np.int32)
            kk, jj = np.nonzero(results != -1)
            inv[kk, results[kk, jj]] = jj
            assert (km.nbr_inv.cpu().numpy() == inv).all()


CHANNELS = [(4, 32), (32, 32), (64, 64), (48, 16), (128, 96), (192, 128), (16, 48), (8, 20)]


@pytest.mark.parametrize('cin,cout', CHANNELS)
def test_subm_conv_fwd_bwd(F, cin, cout):
    coords, _ = _scene(3000, 2)
    torch.manual_seed(cin * 1000 + cout)
    x = torch.randn(len(coords), cin)
    w = torch.randn(27, cin, cout) / (27 * cin) ** 0.5
    g = torch.randn(len(coords), cout)
    nbmaps, nbsizes, _, _ = R.build_kmap(coords, 1, 3, 1)
    want = R.conv_forward(x, w, nbmaps, nbsizes, (len(coords), len(coords)))
    wgi, wgw = R.conv_backward(x, w, g, nbmaps, nbsizes)
    km = F.build_kmap(_dev(coords), (1, 1, 1), (3, 3, 3), (1, 1, 1))
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = F.ConvolutionFunction.apply(xd, wd, km, False)
    assert _rel(out, want) < 1e-4
    out.backward(g.cuda())
    assert _rel(xd.grad, wgi) < 1e-4
    assert _rel(wd.grad, wgw) < 1e-4


@pytest.mark.parametrize('cin,cout', [(32, 32), (64, 96), (256, 128)])
def test_strided_and_transposed_conv(F, cin, cout):
    coords, _ = _scene(3000, 2)
    torch.manual_seed(5)
    nbmaps, nbsizes, oc, _ = R.build_kmap(coords, 1, 2, 2)
    km = F.build_kmap(_dev(coords), (1, 1, 1), (2, 2, 2), (2, 2, 2))
    sizes = (len(coords), len(oc))
    # down
    x = torch.randn(len(coords), cin)
    w = torch.randn(8, cin, cout) / (8 * cin) ** 0.5
    g = torch.randn(len(oc), cout)
    want = R.conv_forward(x, w, nbmaps, nbsizes, sizes)
    wgi, wgw = R.conv_backward(x, w, g, nbmaps, nbsizes)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = F.ConvolutionFunction.apply(xd, wd, km, False)
    assert _rel(out, want) < 1e-4
    out.backward(g.cuda())
    assert _rel(xd.grad, wgi) < 1e-4 and _rel(wd.grad, wgw) < 1e-4
    # up (transposed): reuse the map with roles swapped
    xc = torch.randn(len(oc), cin)
    gu = torch.randn(len(coords), cout)
    want = R.conv_forward(xc, w, nbmaps, nbsizes, sizes, transposed=True)
    wgi, wgw = R.conv_backward(xc, w, gu, nbmaps, nbsizes, transposed=True)
    xd, wd = xc.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = F.ConvolutionFunction.apply(xd, wd, km, True)
    assert _rel(out, want) < 1e-4
    out.backward(gu.cuda())
    assert _rel(xd.grad, wgi) < 1e-4 and _rel(wd.grad, wgw) < 1e-4


# the last four: layer shapes of the cr = 2.0 models (teacher / `_B` student): cs = [64,64,128,256,512,512,256,192,192],
# e.g. 512 -> 512 residual convs, 768 -> 512 and 384 -> 256 after the skip concatenations, 256 -> 512 at stride 16
@pytest.mark.parametrize('cin,cout', [(32, 32), (64, 64), (96, 128), (20, 12), (512, 512), (768, 512), (256, 512), (384, 256)])
@pytest.mark.parametrize('kind', ['subm', 'down', 'up'])
def test_both_conv_schedules_match_oracle(F, cin, cout, kind):
    """The tile schedule and the pair schedule (+ gather-sum) compute the same contraction; both are
    checked against the oracle: forward, transposed and the input gradient."""
    from u2mkd_amd import _lib as L
    coords, _ = _scene(2500, 2)
    torch.manual_seed(11)
    ks, st_ = (3, 1) if kind == 'subm' else (2, 2)
    nbmaps, nbsizes, oc, _ = R.build_kmap(coords, 1, ks, st_)
    km = F.build_kmap(_dev(coords), (1,) * 3, (ks,) * 3, (st_,) * 3)
    sizes = (len(coords), len(oc))
    transposed = kind == 'up'
    n_in, n_out = (sizes[1], sizes[0]) if transposed else sizes
    x = torch.randn(n_in, cin)
    w = torch.randn(ks ** 3, cin, cout) / (ks ** 3 * cin) ** 0.5
    want = R.conv_forward(x, w, nbmaps, nbsizes, sizes, transposed=transposed)
    ts_, ps = km.schedule(transposed), km.pair_schedule()
    wt = F._transpose_weights(w.cuda())
    o_t = torch.full((n_out, cout), float('nan'), device='cuda')
    o_p = torch.full((n_out, cout), float('nan'), device='cuda')
    ts_.run(x.cuda(), w.cuda(), True, cout, 0, o_t)
    ps.run(x.cuda(), wt, cout, transposed, o_p)
    assert _rel(o_t, want) < 1e-4 and _rel(o_p, want) < 1e-4
    assert float((o_t - o_p).abs().max()) < 1e-4
    # input gradient through both schedules
    g = torch.randn(n_out, cout)
    wgi, _ = R.conv_backward(x, w, g, nbmaps, nbsizes, transposed=transposed)
    gi_p = torch.empty(n_in, cin, device='cuda')
    ps.run(g.cuda(), w.cuda().contiguous(), cin, not transposed, gi_p)
    assert _rel(gi_p, wgi) < 1e-4
    # the bf16x3 pair kernel (csrc/conv_px3.hip, what the wide layers run by default): same gates, both roles,
    # bitwise reproducible; NaN-prefilled scratch shows that every y row the gather-sum reads is written
    if L.load().u2mkd_conv_pairs_x3_supported(cin, cout):
        buf = F._scratch(ps.cap * max(cin, cout) * 4, torch.device('cuda', torch.cuda.current_device()))
        buf[:buf.numel() // 4 * 4].view(torch.float32).fill_(float('nan'))
        o_x = torch.full((n_out, cout), float('nan'), device='cuda')
        ps.run(x.cuda(), F._weight_layout(w.cuda(), True, True), cout, transposed, o_x, fragments=True)
        assert _rel(o_x, want) < 1e-4 and float((o_x - o_p).abs().max()) < 1e-4
        gi_x = torch.full((n_in, cin), float('nan'), device='cuda')
        ps.run(g.cuda(), F._weight_layout(w.cuda(), False, True), cin, not transposed, gi_x, fragments=True)
        assert _rel(gi_x, wgi) < 1e-4
        again = torch.empty_like(gi_x)
        ps.run(g.cuda(), F._weight_layout(w.cuda(), False, True), cin, not transposed, again, fragments=True)
        assert torch.equal(gi_x, again)
    gi_t = torch.empty(n_in, cin, device='cuda')
    if kind == 'subm':
        ts_.run(g.cuda(), w.cuda().contiguous(), False, cin, 1, gi_t)   # mirrored offsets on the same table
    else:
        km.schedule(not transposed).run(g.cuda(), w.cuda().contiguous(), False, cin, 0, gi_t)
    assert _rel(gi_t, wgi) < 1e-4
    # the offset-walking tile kernels agree with the product schedule (tile pairs where instantiated)
    nbr_s, order = ts_.tiles()
    a = torch.empty(n_out, cout, device='cuda')
    L.call('u2mkd_conv_forward_sorted', L.ptr(x.cuda()), n_in, cin, L.ptr(wt), cout, L.ptr(nbr_s), L.ptr(order),
           L.ptr(ts_.tile_order), n_out, ks ** 3, 0, L.ptr(a), L.stream())
    assert float((a - o_t).abs().max()) < 1e-4
    with pytest.raises(RuntimeError):
        L.call('u2mkd_conv_forward_sorted', L.ptr(x.cuda()), n_in, cin, L.ptr(wt), cout, L.ptr(nbr_s), L.ptr(order),
               L.ptr(ts_.tile_order), n_out, ks ** 3, 8, L.ptr(a), L.stream())


def test_pair_schedule_is_the_padded_rulebook(F):
    """u2mkd_pairs_build: the valid slots, in order, are exactly torchsparse's nbmaps (grouped by
    offset, ascending output); groups are padded to 128 (two 64-entry tiles of one offset); pos_in / pos_out invert the list."""
    coords, _ = _scene(3000, 2)
    for ks, st_ in ((3, 1), (2, 2)):
        nbmaps, nbsizes, oc, res = R.build_kmap(coords, 1, ks, st_)
        km = F.build_kmap(_dev(coords), (1,) * 3, (ks,) * 3, (st_,) * 3)
        ps = km.pair_schedule()
        p_pad, n_tiles = ps.meta.cpu().tolist()
        pad = (nbsizes + 127) // 128 * 128
        assert p_pad == int(pad.sum()) and n_tiles == p_pad // 64 and p_pad <= ps.cap
        pi, po = ps.pair_in.cpu().numpy()[:p_pad], ps.pair_out.cpu().numpy()[:p_pad]
        assert ((pi >= 0) == (po >= 0)).all()
        assert (np.stack([pi[pi >= 0], po[po >= 0]], 1) == nbmaps).all()
        assert (ps.tile_k.cpu().numpy()[:n_tiles] == np.repeat(np.arange(len(pad)), pad // 64)).all()
        base = np.cumsum(pad) - pad
        for k in range(len(pad)):
            seg = slice(base[k], base[k] + pad[k])
            assert (pi[seg][:nbsizes[k]] >= 0).all() and (pi[seg][nbsizes[k]:] == -1).all()
        pos_out, pos_in = ps.pos_out.cpu().numpy(), ps.pos_in.cpu().numpy()
        assert ((pos_out >= 0) == (res.T >= 0)).all()
        jj, kk = np.nonzero(pos_out >= 0)
        assert (po[pos_out[jj, kk]] == jj).all() and (pi[pos_out[jj, kk]] == res[kk, jj]).all()
        ii, kk = np.nonzero(pos_in >= 0)
        assert len(ii) == len(nbmaps) and (pi[pos_in[ii, kk]] == ii).all()


@pytest.mark.parametrize('n,cin,cout,bias', [(5000, 32, 256, True), (80000, 256, 128, True), (777, 128, 96, False),
                                             (64, 4, 4, True), (1, 96, 20, True), (130, 48, 64, False),
                                             (333, 64, 512, True), (1, 32, 32, True), (20001, 96, 96, False)])
def test_linear_matches_torch_cpu(F, n, cin, cout, bias):
    """LinearFunction (bf16x3 pair kernel dense mode for widths that are multiples of 32, the f32 pair kernel's
    otherwise, + pair-list wgrad) vs nn.functional.linear in fp32 on the CPU."""
    torch.manual_seed(n + cin)
    x = torch.randn(n, cin)
    w = torch.randn(cout, cin) / cin ** 0.5
    b = torch.randn(cout) if bias else None
    g = torch.randn(n, cout)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    want = torch.nn.functional.linear(xr, wr, br)
    want.backward(g)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    out = F.linear(xd, wd, bd)
    assert out.shape == (n, cout) and out.is_contiguous()
    assert _rel(out, want.detach()) < 1e-5
    out.backward(g.cuda())
    assert _rel(xd.grad, xr.grad) < 1e-5 and _rel(wd.grad, wr.grad) < 1e-4
    if bias:
        assert _rel(bd.grad, br.grad) < 1e-5
    with pytest.raises(RuntimeError):
        F.linear(x, w, b)                      # CPU tensors: no fallback


def test_conv_northstar_size_80k_c64(F):
    """BASELINE.json configs[1] micro-shape: 80k voxels, 64->64, k=3, stride 1."""
    b = synth_batch(80000, 1)
    coords = b['coords']
    torch.manual_seed(0)
    x = torch.randn(len(coords), 64)
    w = torch.randn(27, 64, 64) / (27 * 64) ** 0.5
    g = torch.randn(len(coords), 64)
    nbmaps, nbsizes, _, _ = R.build_kmap(coords, 1, 3, 1)
    want = R.conv_forward(x, w, nbmaps, nbsizes, (len(coords), len(coords)))
    wgi, wgw = R.conv_backward(x, w, g, nbmaps, nbsizes)
    km = F.build_kmap(_dev(coords), (1, 1, 1), (3, 3, 3), (1, 1, 1))
    assert (km[1].cpu().numpy() == nbsizes).all() and (km[0].cpu().numpy() == nbmaps).all()
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = F.ConvolutionFunction.apply(xd, wd, km, False)
    out.backward(g.cuda())
    assert _rel(out, want) < 1e-4 and _rel(xd.grad, wgi) < 1e-4 and _rel(wd.grad, wgw) < 2e-4
    # determinism: a second run is bit-identical (no atomics in the conv path)
    xd2, wd2 = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out2 = F.ConvolutionFunction.apply(xd2, wd2, km, False)
    out2.backward(g.cuda())
    assert torch.equal(out, out2) and torch.equal(xd.grad, xd2.grad) and torch.equal(wd.grad, wd2.grad)


def test_empty_and_ragged(F):
    # a single isolated voxel + an empty batch element
    c = np.array([[5, 5, 5, 0], [100, 7, 3, 2], [101, 7, 3, 2]], dtype=np.int32)
    km = F.build_kmap(_dev(c), (1, 1, 1), (3, 3, 3), (1, 1, 1))
    nbmaps, nbsizes, _, res = R.build_kmap(c, 1, 3, 1)
    assert (km.nbr.cpu().numpy() == res).all()
    x = torch.randn(3, 16)
    w = torch.randn(27, 16, 16)
    out = F.ConvolutionFunction.apply(x.cuda(), w.cuda(), km, False)
    assert _rel(out, R.conv_forward(x, w, nbmaps, nbsizes, (3, 3))) < 1e-5
    with pytest.raises(RuntimeError):
        F.sphash(torch.zeros(3, 4, dtype=torch.int32))  # CPU tensor: no fallback


@pytest.mark.parametrize('n,c', [(5000, 32), (80000, 96), (777, 256), (3000, 4), (1300, 48)])
@pytest.mark.parametrize('relu', [False, True])
def test_batch_norm_matches_torch_cpu(F, n, c, relu):
    """HIP BatchNorm(+ReLU) vs nn.BatchNorm1d (+ReLU) on the CPU in fp64: outputs, input and
    affine gradients, running statistics; train and eval modes."""
    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 2.5 + 3.0        # non-zero mean: exercises the centred variance
    g = torch.randn(n, c)
    ref = torch.nn.BatchNorm1d(c).double()
    ref.weight.data.uniform_(0.5, 1.5)
    ref.bias.data.normal_(0, 0.3)
    ours = torch.nn.BatchNorm1d(c).cuda()
    ours.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    for mode in ('train', 'eval'):
        getattr(ref, mode)()
        getattr(ours, mode)()
        xr = x.double().requires_grad_(True)
        yr = ref(xr)
        if relu:
            yr = torch.relu(yr)
        yr.backward(g.double())
        xo = x.cuda().requires_grad_(True)
        yo = F.batch_norm(xo, ours, relu)
        yo.backward(g.cuda())
        assert _rel(yo, yr) < 2e-6, mode
        # with the fused ReLU an output within fp32 rounding of 0 can take the other branch than
        # the fp64 reference: allow a handful of such elements, hold all others to 2e-5
        err = (xo.grad.double().cpu() - xr.grad).abs() / xr.grad.abs().max()
        assert int((err > 2e-5).sum()) <= (4 if relu else 0), (mode, float(err.max()))
        wtol = 2e-3 if relu else 2e-5   # one flipped element moves a sum over N rows by about |g| / |sum|
        assert _rel(ours.weight.grad, ref.weight.grad) < wtol and _rel(ours.bias.grad, ref.bias.grad) < wtol
        assert _rel(ours.running_mean, ref.running_mean) < 1e-6 and _rel(ours.running_var, ref.running_var) < 1e-5
        assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked)
        ref.zero_grad()
        ours.zero_grad()


@pytest.mark.parametrize('n,c', [(5000, 32), (40000, 96), (777, 256)])
def test_batch_norm_with_fused_residual_and_relu(F, n, c):
    """relu(bn(x) + res) in the BatchNorm passes (the tail of ResidualBlock, build_blocks.py:80-83) vs nn.BatchNorm1d,
    add and ReLU on the CPU in fp64: output, the gradients of x, res, gamma, beta; train and eval modes."""
    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 2.0 + 1.0
    r = torch.randn(n, c)
    g = torch.randn(n, c)
    ref = torch.nn.BatchNorm1d(c).double()
    ref.weight.data.uniform_(0.5, 1.5)
    ref.bias.data.normal_(0, 0.3)
    ours = torch.nn.BatchNorm1d(c).cuda()
    ours.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    for mode in ('train', 'eval'):
        getattr(ref, mode)()
        getattr(ours, mode)()
        xr, rr = x.double().requires_grad_(True), r.double().requires_grad_(True)
        yr = torch.relu(ref(xr) + rr)
        yr.backward(g.double())
        xo, ro = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
        yo = F.batch_norm(xo, ours, True, ro)
        yo.backward(g.cuda())
        assert _rel(yo, yr) < 2e-6, mode
        for got, want, what in ((xo.grad, xr.grad, 'dx'), (ro.grad, rr.grad, 'dres')):
            err = (got.double().cpu() - want).abs() / want.abs().max()
            assert int((err > 2e-5).sum()) <= 4, (mode, what, float(err.max()))      # ReLU flips at fp32 rounding of 0
        assert _rel(ours.weight.grad, ref.weight.grad) < 2e-3 and _rel(ours.bias.grad, ref.bias.grad) < 2e-3
        assert _rel(ours.running_mean, ref.running_mean) < 1e-6 and _rel(ours.running_var, ref.running_var) < 1e-5
        ref.zero_grad()
        ours.zero_grad()
    with pytest.raises(ValueError):
        F.batch_norm(x.cuda(), ours, False, r.cuda())


@pytest.mark.parametrize('kind', ['subm', 'down', 'up'])
@pytest.mark.parametrize('cin,cout', [(32, 64), (96, 48)])
def test_v140_backend_format_entries(F, kind, cin, cout):
    """u2mkd_convolution_forward / _backward take the v1.4.0 rulebook as torchsparse.backend's
    convolution_forward_cuda / convolution_backward_cuda do: (neighbor_map [P,2] on the device,
    neighbor_offset [K] on the HOST, transpose).  The rulebook fed here is the oracle's own (nonzero order),
    not one derived from this library's tables."""
    import ctypes
    from u2mkd_amd import _lib as L
    coords, _ = _scene(2200, 2, seed=5)
    torch.manual_seed(3)
    ks, st_ = (3, 1) if kind == 'subm' else (2, 2)
    nbmaps, nbsizes, oc, _ = R.build_kmap(coords, 1, ks, st_)
    k = ks ** 3
    sizes = (len(coords), len(oc))
    transposed = kind == 'up'
    n_in, n_out = (sizes[1], sizes[0]) if transposed else sizes
    x = torch.randn(n_in, cin)
    w = torch.randn(k, cin, cout) / (k * cin) ** 0.5
    g = torch.randn(n_out, cout)
    want = R.conv_forward(x, w, nbmaps, nbsizes, sizes, transposed=transposed)
    want_gi, want_gw = R.conv_backward(x, w, g, nbmaps, nbsizes, transposed=transposed)
    nb_dev = _dev(np.asarray(nbmaps, dtype=np.int32))
    sizes_host = (ctypes.c_int32 * k)(*[int(v) for v in nbsizes])
    lib = L.load()
    nbytes = lib.u2mkd_convolution_workspace_bytes(n_in, n_out, cin, cout, sizes_host, k)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    xd, wd, gd = x.cuda(), w.cuda(), g.cuda()
    out = torch.full((n_out, cout), float('nan'), device='cuda')
    L.call('u2mkd_convolution_forward', L.ptr(xd), n_in, cin, L.ptr(out), n_out, cout, L.ptr(wd), L.ptr(nb_dev),
           ctypes.addressof(sizes_host), k, int(transposed), L.ptr(ws), nbytes, L.stream())
    assert _rel(out, want) < 1e-4
    gi = torch.full((n_in, cin), float('nan'), device='cuda')
    gw = torch.full((k, cin, cout), float('nan'), device='cuda')
    L.call('u2mkd_convolution_backward', L.ptr(xd), n_in, cin, L.ptr(gi), L.ptr(gd), n_out, cout, L.ptr(wd), L.ptr(gw),
           L.ptr(nb_dev), ctypes.addressof(sizes_host), k, int(transposed), L.ptr(ws), nbytes, L.stream())
    assert _rel(gi, want_gi) < 1e-4 and _rel(gw, want_gw) < 1e-4
    # same numbers as the native (neighbour-table) path, bit for bit: one summation order
    km = F.build_kmap(_dev(coords), (1,) * 3, (ks,) * 3, (st_,) * 3)
    o2 = torch.empty(n_out, cout, device='cuda')
    km.pair_schedule().run(xd, F._transpose_weights(wd), cout, transposed, o2)
    assert torch.equal(out, o2)
    # errors: workspace too small, channel count not a multiple of 4
    with pytest.raises(RuntimeError, match='workspace'):
        L.call('u2mkd_convolution_forward', L.ptr(xd), n_in, cin, L.ptr(out), n_out, cout, L.ptr(wd), L.ptr(nb_dev),
               ctypes.addressof(sizes_host), k, int(transposed), L.ptr(ws), 1024, L.stream())
    with pytest.raises(RuntimeError, match='multiples of 4'):
        L.call('u2mkd_convolution_forward', L.ptr(xd), n_in, 6, L.ptr(out), n_out, cout, L.ptr(wd), L.ptr(nb_dev),
               ctypes.addressof(sizes_host), k, int(transposed), L.ptr(ws), nbytes, L.stream())


@pytest.mark.parametrize('cin,cout', [(5, 32), (3, 17), (32, 17)])
def test_conv3d_accepts_any_channel_count(hip, cin, cout):
    """torchsparse v1.4.0 has no channel-multiple restriction (in_channel = 5 with a time channel, 3, an odd class
    count): the drop-in zero-pads to 16-byte rows inside F.conv3d; forward and both gradients equal the oracle."""
    import u2mkd_amd.torchsparse as ts
    from u2mkd_amd.torchsparse.nn import functional as F
    coords, _ = _scene(1500, 2, seed=2)
    torch.manual_seed(cin + cout)
    x = torch.randn(len(coords), cin)
    w = torch.randn(27, cin, cout) / (27 * cin) ** 0.5
    g = torch.randn(len(coords), cout)
    nbmaps, nbsizes, _, _ = R.build_kmap(coords, 1, 3, 1)
    want = R.conv_forward(x, w, nbmaps, nbsizes, (len(coords), len(coords)))
    wgi, wgw = R.conv_backward(x, w, g, nbmaps, nbsizes)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = F.conv3d(ts.SparseTensor(xd, _dev(coords)), wd, kernel_size=3).F
    assert out.shape == (len(coords), cout) and _rel(out, want) < 1e-4
    out.backward(g.cuda())
    assert xd.grad.shape == x.shape and _rel(xd.grad, wgi) < 1e-4
    assert wd.grad.shape == w.shape and _rel(wd.grad, wgw) < 1e-4


def _tile_schedule_torch(ms_sorted, n, k, split):
    """The torch formulation csrc/schedule.hip replaced: block counts from bit planes, stable argsorts."""
    dev = ms_sorted.device
    t = (n + 63) // 64
    ms = torch.zeros(t * 64, dtype=torch.int32, device=dev)
    ms[:n] = ms_sorted
    bits = (ms.view(t, 4, 16, 1) >> torch.arange(k, dtype=torch.int32, device=dev)) & 1
    c16 = bits.sum(2)
    b4 = ((c16 + 15) >> 4).sum(2)
    b2 = ((c16.view(t, 2, 2, k).sum(2) + 15) >> 4).sum(2)
    b1 = ((c16.sum(1) + 15) >> 4).sum(1)
    tile_order = torch.argsort(b1, descending=True, stable=True).int()
    lg = (b1 > split[0]).int() + (b1 > split[1]).int()
    tid = torch.arange(t, dtype=torch.int32, device=dev)
    rows_left = n - tid * 64
    cand_w, cand_code = [], []
    for l, bl in ((0, b1.view(t, 1)), (1, b2), (2, b4)):
        nsub = 1 << l
        sub = torch.arange(nsub, dtype=torch.int32, device=dev).view(1, nsub)
        live = (lg.view(t, 1) == l) & (sub * (64 >> l) < rows_left.view(t, 1))
        cand_w.append(torch.where(live, bl.int(), -1).reshape(-1))
        cand_code.append(((tid.view(t, 1) << 4) | (sub << 2) | l).reshape(-1))
    w = torch.cat(cand_w)
    srt = torch.argsort(w, descending=True, stable=True)
    return tile_order, torch.cat(cand_code)[srt].int(), int((w >= 0).sum())


@pytest.mark.parametrize('n,stride', [(80000, 1), (30000, 1), (5000, 4), (63, 1), (64, 1), (65, 1), (1, 1)])
def test_tile_schedule_kernels_match_the_torch_formulation(F, n, stride):
    """u2mkd_tile_schedule (two launches) gives the tile order, the work items and the live-item count of the
    stable-argsort formulation bit for bit, at the bench size, on a strided level and on ragged tails; every
    row is covered by exactly one live item."""
    from u2mkd_amd.synth import synth_batch
    coords = synth_batch(n, 1, seed=5)['coords']
    ts = 1
    while ts < stride:
        coords = R.spdownsample(coords, 2, 2, ts)
        ts *= 2
    km = F.build_kmap(_dev(coords), (ts,) * 3, (3,) * 3, (1,) * 3)
    sch = km.schedule(False)
    nn_, k = km.n_out, 27
    mask = torch.zeros(nn_, dtype=torch.int32, device='cuda')
    for kk in range(k):
        mask |= (km.nbr[kk] >= 0).int() << kk
    ms = mask[sch.order.long()]
    assert bool((ms[1:] >= ms[:-1]).all())
    want_order, want_items, want_n = _tile_schedule_torch(ms, nn_, k, F._TILE_SPLIT)
    got_n = int(sch.n_items.item())
    assert got_n == want_n
    assert torch.equal(sch.tile_order, want_order)
    assert torch.equal(sch.items[:got_n], want_items[:got_n])
    it = sch.items[:got_n].long()
    tile, sub, lg = it >> 4, (it >> 2) & 3, it & 3
    rows = 64 >> lg
    first = tile * 64 + sub * rows
    cover = torch.zeros(nn_ + 64, dtype=torch.int32, device='cuda')
    for r in (16, 32, 64):
        sel = first[rows == r]
        idx = (sel.view(-1, 1) + torch.arange(r, device='cuda').view(1, -1)).reshape(-1)
        cover.index_add_(0, idx, torch.ones_like(idx, dtype=torch.int32))
    assert bool((cover[:nn_] == 1).all())


@pytest.mark.parametrize('cin,cout', [(64, 64), (32, 64), (64, 128)])
def test_bf16_storage_conv_group_matches_fp32_kernels_on_rounded_inputs(F, cin, cout):
    """BASELINE.json configs[4] kernels: bf16 feature rows in / out, one bf16 weight plane, fp32 accumulation.
    Against the fp32 kernels (f32 MFMA) on the SAME bf16-rounded inputs the only difference is the final rounding
    of each output to bf16 (2^-9 relative) and the accumulation order; the weight gradient (bf16 rows widened in
    registers, f32 MFMA) is bit-identical to the fp32 kernel on the widened rows."""
    from u2mkd_amd import _lib as L
    from u2mkd_amd.synth import synth_batch
    lib = L.load()
    coords = synth_batch(20000, 1, seed=9)['coords']
    km = F.build_kmap(_dev(coords), (1,) * 3, (3,) * 3, (1,) * 3)
    n, k = km.n_out, 27
    sch = km.schedule(False)
    torch.manual_seed(3)
    xb = torch.randn(n, cin, device='cuda').bfloat16()
    gb = torch.randn(n, cout, device='cuda').bfloat16()
    w = (torch.randn(k, cin, cout, device='cuda') / (k * cin) ** 0.5).bfloat16().float()
    st = L.stream()

    def frags(arith):
        buf = torch.empty(2, lib.u2mkd_weight_fragments_bytes(k, cin, cout, arith), dtype=torch.uint8, device='cuda')
        L.call('u2mkd_weight_fragments', L.ptr(w), k, cin, cout, 2, arith, L.ptr(buf), st)
        return buf
    f3, f1 = frags(3), frags(1)
    assert f3.shape[1] * 2 == f1.shape[1]
    for flip, a_b, ca, cb_, fi in ((0, xb, cin, cout, 0), (1, gb, cout, cin, 1)):      # forward, input gradient
        if not lib.u2mkd_conv_tiles_supported(ca, cb_, k):
            continue
        ob = torch.full((n, cb_), float('nan'), device='cuda').bfloat16()
        L.call('u2mkd_conv_forward_tiles_bf16', L.ptr(a_b), n, ca, L.ptr(f3[fi]), cb_, L.ptr(sch.nbr_s), L.ptr(sch.order),
               L.ptr(sch.items), L.ptr(sch.n_items), n, k, flip, L.ptr(ob), st)
        of = torch.empty(n, cb_, device='cuda')
        a_f = a_b.float()
        L.call('u2mkd_conv_forward_tiles', L.ptr(a_f), n, ca, L.ptr(f1[fi]), cb_, L.ptr(sch.nbr_s), L.ptr(sch.order),
               L.ptr(sch.items), L.ptr(sch.n_items), n, k, flip, 1, L.ptr(of), st)
        err = (ob.float() - of).abs()
        assert bool(torch.isfinite(ob.float()).all())
        assert float((err / (of.abs() + 1e-3 * of.abs().max())).max()) < 2 ** -7
        again = torch.empty_like(ob)
        L.call('u2mkd_conv_forward_tiles_bf16', L.ptr(a_b), n, ca, L.ptr(f3[fi]), cb_, L.ptr(sch.nbr_s), L.ptr(sch.order),
               L.ptr(sch.items), L.ptr(sch.n_items), n, k, flip, L.ptr(again), st)
        assert torch.equal(ob, again)
    pairs, _, plan = km.pairs_plan()
    nbytes = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, cin, cout, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    dwb, dwf = torch.empty(k, cin, cout, device='cuda'), torch.empty(k, cin, cout, device='cuda')
    L.call('u2mkd_conv_wgrad_pairs_bf16', L.ptr(xb), cin, L.ptr(gb), cout, L.ptr(pairs), L.ptr(plan), n, k, 0, L.ptr(ws), nbytes,
           L.ptr(dwb), st)
    xf, gf = xb.float(), gb.float()          # (named: two temporaries would share one freed block)
    L.call('u2mkd_conv_wgrad_pairs', L.ptr(xf), cin, L.ptr(gf), cout, L.ptr(pairs), L.ptr(plan), n, k, 0,
           L.ptr(ws), nbytes, L.ptr(dwf), st)
    assert torch.equal(dwb, dwf)


def test_batched_weight_fragments_equal_the_per_weight_launches(F):
    """u2mkd_weight_fragments_batch (one launch for every trainable weight, behind the optimizer step) writes the same
    bytes as one u2mkd_weight_fragments(transpose = 2) launch per weight: conv kernels [27, cin, cout], a strided
    [8, ..], nn.Linear [1, out, in]; bf16x3 (3 planes) and bf16 storage (1 plane) mixed in one table."""
    from u2mkd_amd import _lib as L
    lib = L.load()
    st = L.stream()
    torch.manual_seed(11)
    shapes = [(27, 64, 64, 2), (27, 32, 96, 2), (8, 64, 128, 2), (1, 256, 128, 2), (27, 64, 64, 3), (1, 96, 32, 3), (27, 128, 256, 2)]
    ws, ref, bufs, rows, first = [], [], [], [], 0
    for k, r, c, arith in shapes:
        w = torch.randn(k, r, c, device='cuda')
        nbytes = lib.u2mkd_weight_fragments_bytes(k, r, c, arith)
        a = torch.empty(2, nbytes, dtype=torch.uint8, device='cuda')
        L.call('u2mkd_weight_fragments', L.ptr(w), k, r, c, 2, arith, L.ptr(a), st)
        b = torch.full((2, nbytes), 0xA5, dtype=torch.uint8, device='cuda')
        rows.append([w.data_ptr(), b.data_ptr(), first, k, r, c, 3 if arith == 2 else 1, 0])
        first += 2 * (k * r * c // 512)
        ws.append(w); ref.append(a); bufs.append(b)
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    L.call('u2mkd_weight_fragments_batch', L.ptr(table), len(rows), first, st)
    for a, b, sh in zip(ref, bufs, shapes):
        assert torch.equal(a, b), sh


def test_optimizer_step_refreshes_every_fragment_image_in_one_launch(F, monkeypatch):
    """The product flow: the first forward + backward lay each weight out once (per-weight launch, registers the image);
    every optimizer step afterwards re-lays ALL registered weights by ONE u2mkd_weight_fragments_batch launch, the next
    step issues no per-weight launch, and its outputs equal a from-scratch evaluation with the updated weights."""
    from u2mkd_amd import _lib as L
    from u2mkd_amd import torchsparse
    import u2mkd_amd.torchsparse.nn as spnn
    coords, feats = _scene(4000, 1, seed=5)
    torch.manual_seed(2)
    net = torch.nn.ModuleList([spnn.Conv3d(32, 64, 3), spnn.Conv3d(64, 64, 3), spnn.Conv3d(64, 32, 3)]).cuda()
    x0 = torchsparse.SparseTensor(torch.randn(len(coords), 32, device='cuda'), _dev(coords))
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    calls = []
    real = L.call

    def counting(name, *a):
        if name.startswith('u2mkd_weight_fragments'):
            calls.append(name)
        return real(name, *a)
    monkeypatch.setattr(L, 'call', counting)

    def step():
        y = x0
        for m in net:
            y = m(y)
        loss = (y.F ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return y.F.detach().clone()
    step()
    assert calls.count('u2mkd_weight_fragments') == 3 and calls.count('u2mkd_weight_fragments_batch') == 1
    del calls[:]
    y2 = step()
    assert calls == ['u2mkd_weight_fragments_batch'], calls
    # the same second forward from scratch: fresh modules holding the weights as they were before the second update
    del calls[:]
    y3 = step()
    assert calls == ['u2mkd_weight_fragments_batch']
    F.invalidate_weight_caches()                 # forces the per-weight path on the current weights
    with torch.no_grad():
        y = x0
        for m in net:
            y = m(y)
    assert calls.count('u2mkd_weight_fragments') == 3
    del calls[:]
    with torch.no_grad():
        z = x0
        for m in net:
            z = m(z)
    assert calls == [] and torch.equal(y.F, z.F)
    assert not torch.equal(y2, y3)
    # batched images == per-weight images: run one more step, then compare a forward on the batch-refreshed images with one
    # on images rebuilt per weight
    step()
    with torch.no_grad():
        a = x0
        for m in net:
            a = m(a)
    F.invalidate_weight_caches()
    with torch.no_grad():
        b = x0
        for m in net:
            b = m(b)
    assert torch.equal(a.F, b.F)


@pytest.mark.gpu
@pytest.mark.parametrize('n,cin,cout', [(9000, 128, 128), (3001, 64, 160), (70, 256, 96), (20000, 96, 96)])
def test_conv_store_takes_the_batch_norm_statistics(n, cin, cout, monkeypatch):
    """spnn.Conv3d -> spnn.BatchNorm (-> ReLU) in training mode: the wide layers' gather-sum writes the BatchNorm's slab
    statistics with its output (u2mkd_pairs_gather_sum_stats) and the BatchNorm starts at its merge step
    (u2mkd_bn_train_forward_from_partial).  Against the same sequence with U2MKD_CONV_BN_STATS off (the BatchNorm's own statistics
    pass): outputs, running statistics, step counter, and every gradient."""
    import torch.nn as nn
    from u2mkd_amd import torchsparse as ts
    from u2mkd_amd.lidar.blocks import FusedSequential
    from u2mkd_amd.torchsparse import nn as spnn
    from u2mkd_amd.torchsparse.nn import functional as F
    from u2mkd_amd import _lib as L
    rng = np.random.default_rng(n)
    pts = np.unique(rng.integers(0, 40, size=(n, 3)), axis=0)
    coords = torch.from_numpy(np.concatenate([pts, np.zeros((len(pts), 1), dtype=np.int64)], 1)).int().cuda()
    feats0 = (torch.randn(len(pts), cin, device='cuda') * 2.0 + 0.7)
    torch.manual_seed(5)
    seq0 = FusedSequential(spnn.Conv3d(cin, cout, kernel_size=3), spnn.BatchNorm(cout), spnn.ReLU(True)).cuda().train()
    state = {k: v.clone() for k, v in seq0.state_dict().items()}
    res = {}
    for on in (True, False):
        monkeypatch.setattr(F, '_CONV_BN_STATS', on)
        seq = FusedSequential(spnn.Conv3d(cin, cout, kernel_size=3), spnn.BatchNorm(cout), spnn.ReLU(True)).cuda().train()
        seq.load_state_dict(state)
        seen = []
        real = L.call
        monkeypatch.setattr(L, 'call', lambda name, *a: (seen.append(name), real(name, *a))[1])
        x = ts.SparseTensor(feats0.clone().requires_grad_(True), coords)
        y = seq(x)
        monkeypatch.setattr(L, 'call', real)
        (y.F * torch.linspace(0.5, 1.5, cout, device='cuda')).sum().backward()
        res[on] = (y.F.detach(), x.F.grad.detach(), seq[0].kernel.grad.detach(), seq[1].weight.grad.detach(), seq[1].bias.grad.detach(),
                   seq[1].running_mean.clone(), seq[1].running_var.clone(), int(seq[1].num_batches_tracked))
        took = 'u2mkd_pairs_gather_sum_stats' in seen and 'u2mkd_bn_train_forward_from_partial' in seen
        assert took == (on and F._pairs_mode(cin, cout, len(pts))), (on, sorted(set(seen)))
    assert res[True][7] == res[False][7] == 1
    for a, b in zip(res[True][:7], res[False][:7]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale
