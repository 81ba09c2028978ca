"""Edge cases on the HIP path: empty and tiny inputs through every operator family
(the reference's own tests cover empty / single-element cases for its ops)."""
import numpy as np
import pytest
import torch

from oracle import ts_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F(hip):
    from u2mkd_amd.torchsparse.nn import functional as F
    return F


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize('n', [0, 1, 2, 65])
@pytest.mark.parametrize('cin,cout', [(32, 32), (96, 128)])
def test_conv_tiny_maps_both_schedules(F, n, cin, cout):
    rng = np.random.default_rng(n)
    c = np.unique(np.concatenate([rng.integers(0, 4, (n, 3)), np.zeros((n, 1), np.int64)], 1), axis=0).astype(np.int32)
    n = len(c)
    km = F.build_kmap(_dev(c).view(-1, 4), (1, 1, 1), (3, 3, 3), (1, 1, 1))
    x = torch.randn(n, cin)
    w = torch.randn(27, cin, cout) / (27 * cin) ** 0.5
    if n:
        nbmaps, nbsizes, _, _ = R.build_kmap(c, 1, 3, 1)
        want = R.conv_forward(x, w, nbmaps, nbsizes, (n, n))
    else:
        want = torch.zeros(0, cout)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = F.ConvolutionFunction.apply(xd, wd, km, False)          # schedule chosen by the channel rule
    assert out.shape == (n, cout)
    if n:
        assert float((out.cpu() - want).abs().max()) < 1e-4
    out.sum().backward()
    assert xd.grad.shape == x.shape and wd.grad.shape == w.shape
    assert torch.isfinite(wd.grad).all() and torch.isfinite(xd.grad).all()
    if n == 0:
        assert float(wd.grad.abs().max()) == 0.0
    # both schedules explicitly
    wt = F._transpose_weights(w.cuda())
    o1 = torch.zeros(n, cout, device='cuda')
    o2 = torch.zeros(n, cout, device='cuda')
    km.schedule(False).run(x.cuda(), w.cuda(), True, cout, 0, o1)
    km.pair_schedule().run(x.cuda(), wt, cout, False, o2)
    if n:
        assert float((o1.cpu() - want).abs().max()) < 1e-4 and float((o2.cpu() - want).abs().max()) < 1e-4


def test_strided_conv_on_single_voxel_and_empty(F):
    for c in (np.zeros((0, 4), np.int32), np.array([[3, 5, 7, 0]], np.int32)):
        km = F.build_kmap(_dev(c).view(-1, 4), (1, 1, 1), (2, 2, 2), (2, 2, 2))
        n = len(c)
        assert km.n_in == n and km.n_out == n
        x = torch.randn(n, 32).cuda().requires_grad_(True)
        w = torch.randn(8, 32, 64).cuda().requires_grad_(True)
        y = F.ConvolutionFunction.apply(x, w, km, False)
        assert y.shape == (n, 64)
        z = F.ConvolutionFunction.apply(y, w.transpose(1, 2).contiguous(), km, True)      # back up
        assert z.shape == (n, 32)
        z.sum().backward()
        assert x.grad.shape == (n, 32) and torch.isfinite(w.grad).all()


def test_linear_bn_point_voxel_tiny(F):
    from u2mkd_amd.lidar.blocks import PointBatchNorm1d
    w = torch.randn(64, 32).cuda().requires_grad_(True)
    b = torch.randn(64).cuda().requires_grad_(True)
    for n in (0, 1, 3):
        x = torch.randn(n, 32).cuda().requires_grad_(True)
        y = F.linear(x, w, b)
        assert y.shape == (n, 64)
        y.sum().backward()
        if n:
            want = torch.nn.functional.linear(x.detach().cpu(), w.detach().cpu(), b.detach().cpu())
            assert float((y.detach().cpu() - want).abs().max()) < 1e-5
    bn = PointBatchNorm1d(64).cuda()
    with pytest.raises(ValueError):
        F.batch_norm(torch.randn(1, 64).cuda(), bn.train())           # nn.BatchNorm1d raises too
    y = F.batch_norm(torch.randn(1, 64).cuda(), bn.eval(), relu=True)  # eval mode: any batch size
    assert y.shape == (1, 64) and float(y.min()) >= 0.0
    # voxelise / devoxelise with no points
    counts = F.spcount(torch.zeros(0, dtype=torch.int32, device='cuda'), 5)
    assert counts.tolist() == [0] * 5
    out = F.spvoxelize(torch.zeros(0, 8, device='cuda'), torch.zeros(0, dtype=torch.int32, device='cuda'), counts)
    assert out.shape == (5, 8) and float(out.abs().max()) == 0.0
    out = F.spdevoxelize(torch.randn(5, 8, device='cuda'), torch.zeros(0, 8, dtype=torch.int32, device='cuda'),
                         torch.zeros(0, 8, device='cuda'))
    assert out.shape == (0, 8)


def test_window_attention_single_token_windows(hip):
    """Every token alone in its window: softmax over one key = 1, out = v + Tv(rel = 0)."""
    from u2mkd_amd import sptr
    n, h, d, qgl = 40, 2, 16, 24
    xyz = (torch.arange(n).float().view(-1, 1) * torch.tensor([[10.0, 0.0, 0.0]])).cuda()      # 10 m apart
    b = torch.zeros(n, dtype=torch.int32).cuda()
    window, quant = np.array([0.6, 0.6, 0.6]), np.array([0.025] * 3)
    plan = sptr.WindowPlan(xyz, b, window)
    assert int(plan.wlen.max()) == 1
    q, k, v = (torch.randn(n, h, d, device='cuda', requires_grad=True) for _ in range(3))
    tq, tk, tv = (torch.randn(2 * qgl - 1, 3, h, d, device='cuda', requires_grad=True) for _ in range(3))
    out = sptr.window_attention(q, k, v, xyz, plan, quant, qgl, tq, tk, tv, None)
    want = v + tv[qgl - 1].sum(0).unsqueeze(0)
    assert float((out - want).abs().max()) < 1e-5
    out.sum().backward()
    # (scores are recomputed in the backward in a different summation order: p = exp(s' - lse) = 1 +- 1e-5)
    assert float((v.grad - 1.0).abs().max()) < 1e-4
    assert float(q.grad.abs().max()) < 1e-4 and float(k.grad.abs().max()) < 1e-4      # softmax over one key
    assert float((tv.grad[qgl - 1] - n).abs().max()) < 1e-2


def test_spvcnn_step_on_a_tiny_scene(hip):
    from u2mkd_amd import lidar, train as T
    from u2mkd_amd.synth import synth_batch
    b = synth_batch(60, 2, 3)
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    model = lidar.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
    run = T.LidarStep(model)
    l0 = float(run(feats, coords, labels))
    l1 = float(run(feats, coords, labels))
    assert np.isfinite(l0) and np.isfinite(l1)


@pytest.mark.parametrize('bad', [(131072, 0, 0, 0), (0, -131073, 0, 0), (0, 0, 0, 512), (0, 0, 0, -1)])
def test_downsample_rejects_coordinates_outside_the_key_range(F, bad):
    """torch.unique(dim=0) of the reference has no coordinate limit; the packed int64 key has (18 bits per axis,
    10 for the batch index).  A row outside it must raise, not merge silently into another voxel; rows at the
    limits themselves are exact."""
    ok = np.array([[131070, -131072, 6, 0], [2, 4, 6, 511], [0, 0, 0, 0], [1, 1, 1, 0]], np.int32)
    out = F.spdownsample(_dev(ok), 2, 2, 1).cpu().numpy()
    want = R.spdownsample(ok, 2, 2, 1)
    assert np.array_equal(out, want)
    with pytest.raises(ValueError):
        F.spdownsample(_dev(np.concatenate([ok, np.array([bad], np.int32)])), 2, 2, 1)
    out = F.spdownsample(_dev(ok), 2, 2, 1).cpu().numpy()          # the flag was cleared
    assert np.array_equal(out, want)


def test_deferred_range_check_raises_once_at_the_end_of_the_context(F):
    """Inside `deferred_range_check()` (point_voxel.prepare_geometry builds the four levels of the encoder in it) the
    flag of every spdownsample call is read once, when the context ends."""
    ok = np.array([[131070, -131072, 6, 0], [2, 4, 6, 511], [0, 0, 0, 0], [1, 1, 1, 0]], np.int32)
    bad = np.concatenate([ok, np.array([[0, 0, 0, 512]], np.int32)])
    with F.deferred_range_check():
        a = F.spdownsample(_dev(ok), 2, 2, 1)
        b = F.spdownsample(a, 2, 2, 2)
    assert np.array_equal(a.cpu().numpy(), R.spdownsample(ok, 2, 2, 1)) and b.shape[0] > 0
    with pytest.raises(ValueError):
        with F.deferred_range_check():
            F.spdownsample(_dev(ok), 2, 2, 1)
            F.spdownsample(_dev(bad), 2, 2, 1)          # no error here ...
            F.spdownsample(_dev(ok), 2, 2, 1)
        # ... but here
    assert np.array_equal(F.spdownsample(_dev(ok), 2, 2, 1).cpu().numpy(), R.spdownsample(ok, 2, 2, 1))   # flag cleared


def test_floor_coords_kernel_equals_the_torch_formula(F):
    """u2mkd_floor_coords vs torch.floor(xyz / s).int() * s | b.int() bit for bit, negative and fractional inputs."""
    from u2mkd_amd.lidar.point_voxel import _floor_coords
    g = torch.Generator().manual_seed(0)
    pc = torch.cat([(torch.rand(5000, 3, generator=g) - 0.3) * 700.0, torch.randint(0, 4, (5000, 1), generator=g).float()], 1)
    pc[:10, :3] = torch.tensor([[-0.0, 8.0, -8.0]] * 10)
    for s in (1, 2, 8, 16):
        want = torch.cat([torch.floor(pc[:, :3] / s).int() * s, pc[:, -1].int().view(-1, 1)], 1)
        assert torch.equal(_floor_coords(pc.cuda(), s).cpu(), want)


@pytest.mark.parametrize('e,nv,drop', [(200000, 60000, 0.1), (480000, 5000, 0.5), (1000, 1, 0.0), (70000, 300000, 0.0), (5, 3, 0.4),
                                       (20000, 4, 0.0), (0, 7, 0.0)])
def test_csr_build_equals_a_stable_argsort(hip, e, nv, drop):
    """u2mkd_csr_build (counting sort + per-segment sort of the entry ids, csrc/csr.hip) groups the entries exactly as
    the stable argsort it replaces: bit-identical live entries and segment offsets, dropped keys (negative or >= nv)
    left out; short (<= 24), long (LDS-ranked) and very long (> 2048: global-memory ranked) segments all occur."""
    from u2mkd_amd.torchsparse.nn import functional as F
    g = torch.Generator().manual_seed(e + nv)
    keys = torch.randint(0, max(nv, 1), (e,), generator=g, dtype=torch.int32)
    if e > 1000:
        keys[: e // 8] = keys[: e // 8] % max(nv // 50, 1)          # a few heavy destinations
    dropped = torch.rand(e, generator=g) < drop
    keys[dropped] = torch.where(torch.rand(int(dropped.sum()), generator=g) < 0.5, -1, nv + 3).int()
    order, seg = F._csr_by_destination(keys.cuda(), nv)
    valid = (keys >= 0) & (keys < nv)
    want_seg = torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(keys[valid].long(), minlength=nv).cumsum(0)])
    assert torch.equal(seg.cpu().long(), want_seg)
    ids = torch.arange(e)[valid]
    want = ids[torch.argsort(keys[valid].long(), stable=True)]
    n_live = int(want_seg[-1])
    assert torch.equal(order.cpu().long()[:n_live], want)
    assert bool((order.cpu()[n_live:] == 0).all())


def test_batched_geometry_pass_equals_the_level_by_level_one_in_two_round_trips(hip, monkeypatch):
    """point_voxel.prepare_geometry_many (the KD step's student + teacher): the same voxel sets, coordinates of every
    level and kernel maps as the level-by-level form (one torch.unique per set), with two host round trips in total and
    no torch.unique at all."""
    import numpy as np
    from u2mkd_amd import torchsparse as ts
    from u2mkd_amd.lidar import point_voxel as PV
    from u2mkd_amd.synth import synth_batch
    from u2mkd_amd.torchsparse.nn import functional as spf
    scenes = [synth_batch(6000, 2, seed=5), synth_batch(9000, 1, seed=6, sweeps=3)]
    mk = lambda b: ts.SparseTensor(torch.from_numpy(b['feats']).cuda(), torch.from_numpy(b['coords']).cuda())
    want = [PV._prepare_geometry_level_by_level(mk(b), 0.05, 0.05) for b in scenes]
    trips = []
    real = spf.wait_counts
    monkeypatch.setattr(spf, 'wait_counts', lambda h, *a: (trips.append(h.n), real(h, *a))[1])
    monkeypatch.setattr(torch, 'unique', lambda *a, **k: (_ for _ in ()).throw(AssertionError('torch.unique in the batched pass')))
    got = PV.prepare_geometry_many([(mk(b), 0.05, 0.05) for b in scenes])
    assert trips == [2, 2 * 5]                       # sizes of the two voxel sets; 4 levels + the range flag per network
    for (zw, xw), (zg, xg) in zip(want, got):
        assert torch.equal(xw.C, xg.C) and torch.equal(xw.F, xg.F) and torch.equal(zw.C, zg.C)
        assert list(xw.cmaps) == list(xg.cmaps) and list(xw.kmaps) == list(xg.kmaps) and len(xw.kmaps) == 9
        for key in xw.cmaps:
            assert torch.equal(xw.cmaps[key], xg.cmaps[key]), key
        for key in xw.kmaps:
            a, b = xw.kmaps[key], xg.kmaps[key]
            assert (a.n_in, a.n_out) == (b.n_in, b.n_out) and torch.equal(a.nbr, b.nbr), key
            assert (a.nbr_inv is None) == (b.nbr_inv is None) and (a.nbr_inv is None or torch.equal(a.nbr_inv, b.nbr_inv))
        assert torch.equal(zw.additional_features['idx_query'][1], zg.additional_features['idx_query'][1])
    # an out-of-range coordinate is still reported (the flag travels with the sizes)
    bad = torch.from_numpy(scenes[0]['coords']).cuda().clone()
    bad[7, 0] = 140000 * 16
    monkeypatch.undo()
    with pytest.raises(ValueError, match='packed key range'):
        spf.DownsamplePyramid(bad, [2, 4]).finish(spf.read_counts(spf.DownsamplePyramid(bad, [2, 4]).counts()))
    ok = spf.DownsamplePyramid(torch.from_numpy(scenes[0]['coords']).cuda(), [2])
    assert ok.finish(spf.read_counts(ok.counts()))[(2, 2, 2)].shape[1] == 4          # the flag was cleared


def test_counts_mailbox_delivers_posted_device_integers(hip):
    """functional.post_counts / wait_counts (csrc/mailbox.hip): device integers (int64 and int32, views with storage offsets)
    posted by one kernel into mapped host memory and polled by the host -- the values of the stream-ordered copy they replace
    (torch.unique's size read in the reference, core/models/utils.py:20), for more posts than the ring has slots, for several
    posts outstanding at once, from a side stream, and next to the A/B path (`U2MKD_COUNTS_MAILBOX=0`'s stack + tolist)."""
    from u2mkd_amd.torchsparse.nn import functional as spf
    base = torch.arange(1000, dtype=torch.int64, device='cuda') * 3
    for r in range(150):                                   # (64 slots: the ring wraps twice)
        ts_ = [base[r + i] if i % 2 else torch.tensor([r * 7 + i], dtype=torch.int32, device='cuda') for i in range(1 + r % 11)]
        want = [int(t) for t in ts_]
        assert spf.read_counts(ts_) == want
    hs = [spf.post_counts([base[i], base[i + 1]]) for i in range(20)]
    assert [spf.wait_counts(h) for h in reversed(hs)] == [[3 * i, 3 * i + 3] for i in reversed(range(20))]
    assert spf.wait_counts(hs[0]) == [0, 3]                # (a second wait returns the kept values)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.cuda._sleep(20_000_000)                      # the producer is still running when the host starts to poll
        x = torch.full((1,), 41, dtype=torch.int64, device='cuda') + 1
        h = spf.post_counts([x])
    assert spf.wait_counts(h) == [42]
    assert spf.read_counts([]) == []
    big = [base[i] for i in range(33)]                     # more than a slot holds: the copy path
    assert spf.read_counts(big) == [3 * i for i in range(33)]
