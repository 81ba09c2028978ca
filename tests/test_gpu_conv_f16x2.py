"""GPU parity of the tile kernel's f16x2 arithmetic (u2mkd_conv_forward_tiles, arith 4): fp32 rows through two fp16 planes per
operand, three partial products, per-row / per-tensor power-of-two scaling.  Held to fp32 GEMM accuracy against a float64
evaluation of torchsparse v1.4.0's gather -> mm -> scatter-add on the same seeded inputs, on data
whose magnitude fp16 could not hold unscaled; bitwise reproducible; the batched fragment refresh writes the same bytes as
the per-weight launch."""
import numpy as np
import pytest
import torch

from u2mkd_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F(hip):
    from u2mkd_amd.torchsparse.nn import functional as F
    return F


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _conv_f64(x, w, nbr, flip):
    """out[j] = sum_k x[nbr[k][j]] @ B_k in float64; forward: B_k = w[k] ([cin, cout]); input gradient (flip): the transposed
    weights of the mirrored offset.  Also returns sum_k |x| @ |B_k| (the scale an error bound is relative to)."""
    k, n = nbr.shape
    xd, wd = x.double(), w.double()
    cols = w.shape[1] if flip else w.shape[2]
    out = torch.zeros(n, cols, dtype=torch.float64, device=x.device)
    mag = torch.zeros_like(out)
    for kk in range(k):
        idx = nbr[kk].long()
        ok = idx >= 0
        rows = xd[idx.clamp(min=0)] * ok[:, None]
        b = wd[k - 1 - kk].t() if flip else wd[kk]
        out += rows @ b
        mag += rows.abs() @ b.abs()
    return out, mag


def _run_tiles(L, lib, sch, x, w, flip, arith, n, k):
    cin, cout = w.shape[1], w.shape[2]
    ca, cb = (cout, cin) if flip else (cin, cout)
    buf = torch.empty(2, lib.u2mkd_weight_fragments_bytes(k, cin, cout, arith), dtype=torch.uint8, device='cuda')
    L.call('u2mkd_weight_fragments', L.ptr(w), k, cin, cout, 2, arith, L.ptr(buf), L.stream())
    out = torch.full((n, cb), float('nan'), device='cuda')
    L.call('u2mkd_conv_forward_tiles', L.ptr(x), n, ca, L.ptr(buf[1 if flip else 0]), cb, L.ptr(sch.nbr_s), L.ptr(sch.order),
           L.ptr(sch.items), L.ptr(sch.n_items), n, k, int(flip), arith, L.ptr(out), L.stream())
    return out


@pytest.mark.parametrize('cin,cout', [(64, 64), (32, 64), (64, 128), (128, 64), (32, 32), (128, 128), (64, 96)])
@pytest.mark.parametrize('spread', ['unit', 'rows', 'tiny', 'huge'])
def test_f16x2_tiles_match_float64_at_fp32_gemm_accuracy(F, cin, cout, spread):
    from u2mkd_amd import _lib as L
    lib = L.load()
    coords = synth_batch(20000, 1, seed=9)['coords']
    km = F.build_kmap(_dev(coords), (1,) * 3, (3,) * 3, (1,) * 3)
    n, k = km.n_out, 27
    sch = km.schedule(False)
    torch.manual_seed(5)
    w = torch.randn(k, cin, cout, device='cuda') / (k * cin) ** 0.5
    for flip in (False, True):
        ca, cb = (cout, cin) if flip else (cin, cout)
        if not lib.u2mkd_conv_tiles_supported(ca, cb, k) or lib.u2mkd_conv_tiles_arith(ca, cb, k) != 4:
            continue
        x = torch.randn(n, ca, device='cuda')
        ww = w
        if spread == 'rows':        # every row its own magnitude, 1e-18 .. 1e18; a tenth of the rows all zero, some with one entry
            x = x * torch.pow(10.0, torch.randint(-18, 19, (n, 1), device='cuda').float())
            x[torch.rand(n, device='cuda') < 0.1] = 0.0
            one = torch.rand(n, device='cuda') < 0.05
            keep = torch.zeros(ca, device='cuda'); keep[3] = 1.0
            x[one] = x[one] * keep
        elif spread == 'tiny':      # gradients late in training; weights after heavy decay
            x = x * 1e-20
            ww = w * 1e-6
        elif spread == 'huge':      # beyond fp16's 65504 on both sides
            x = x * 3e7
            ww = w * 1e9
        got = _run_tiles(L, lib, sch, x, ww, flip, 4, n, k)
        # the table in launch order: nbr_s[k][i] belongs to output row order[i]
        tbl = torch.empty(k, n, dtype=torch.int32, device='cuda')
        tbl[:, sch.order.long()] = sch.nbr_s
        want, mag = _conv_f64(x, ww, tbl, flip)
        assert bool(torch.isfinite(got).all())
        # fp32 GEMM accuracy: |error| <= 2^-20 sum |x||w| (three dropped / rounded terms of 2^-24 each per product, plus the
        # fp32 accumulation of up to 27 x 128 terms)
        err = (got.double() - want).abs()
        bound = mag * 2.0 ** -20 + 1e-300
        assert bool((err <= bound).all()), float((err / bound).max())
        again = _run_tiles(L, lib, sch, x, ww, flip, 4, n, k)
        assert torch.equal(got, again)
        if spread == 'unit':        # and next to bf16x3: the same accuracy class
            x3 = _run_tiles(L, lib, sch, x, ww, flip, 2, n, k)
            e3 = (x3.double() - want).abs()
            assert float(err.max()) <= 4.0 * float(e3.max()) + 1e-12


def test_f16x2_batched_fragments_equal_the_per_weight_launch(F):
    """u2mkd_weight_fragments_batch with f16x2 jobs (planes = 2): per-tensor scale + both planes + both trailers, the same
    bytes as u2mkd_weight_fragments(arith = 4); mixed with bf16x3 and one-plane jobs in one table."""
    from u2mkd_amd import _lib as L
    lib = L.load()
    st = L.stream()
    torch.manual_seed(11)
    shapes = [(27, 64, 64, 4), (27, 32, 96, 2), (8, 64, 128, 4), (27, 128, 128, 4), (27, 64, 64, 3), (27, 32, 32, 4)]
    ws, ref, bufs, rows, first = [], [], [], [], 0
    for i, (k, r, c, arith) in enumerate(shapes):
        w = torch.randn(k, r, c, device='cuda') * 10.0 ** (3 * i - 6)
        nbytes = lib.u2mkd_weight_fragments_bytes(k, r, c, arith)
        a = torch.empty(2, nbytes, dtype=torch.uint8, device='cuda')
        L.call('u2mkd_weight_fragments', L.ptr(w), k, r, c, 2, arith, L.ptr(a), st)
        b = torch.full((2, nbytes), 0xA5, dtype=torch.uint8, device='cuda')
        rows.append([w.data_ptr(), b.data_ptr(), first, k, r, c, {2: 3, 3: 1, 4: 2}[arith], 0])
        first += 2 * (k * r * c // 512)
        ws.append(w); ref.append(a); bufs.append(b)
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    L.call('u2mkd_weight_fragments_batch', L.ptr(table), len(rows), first, st)
    for a, b, sh, w in zip(ref, bufs, shapes, ws):
        assert torch.equal(a, b), sh
        if sh[3] == 4:      # the trailer: {scale, 1 / scale}, the largest |w| scaled into [2^14, 2^15)
            tr = a[0, -16:].view(torch.float32)
            m = float(w.abs().max()) * float(tr[0])
            assert 2.0 ** 14 <= m < 2.0 ** 15, m
            assert float(tr[0]) * float(tr[1]) == 1.0


def test_conv3d_default_runs_f16x2_and_trains(F):
    """spnn.Conv3d on fp32 rows goes through the f16x2 tile kernel by default (u2mkd_conv_tiles_arith = 4 for 64 -> 64) and its
    forward / input gradient agree with the float64 evaluation; the weight's f16x2 image is refreshed by the optimizer hook."""
    from u2mkd_amd import _lib as L
    from u2mkd_amd import torchsparse
    import u2mkd_amd.torchsparse.nn as spnn
    lib = L.load()
    if lib.u2mkd_conv_tiles_arith(64, 64, 27) != 4:
        pytest.skip('U2MKD_CONV_ARITH overrides the default')
    coords = synth_batch(6000, 1, seed=4)['coords']
    torch.manual_seed(8)
    conv = spnn.Conv3d(64, 64, 3).cuda()
    x = torch.randn(len(coords), 64, device='cuda', requires_grad=True)
    y = conv(torchsparse.SparseTensor(x, _dev(coords)))
    g = torch.randn_like(y.F)
    y.F.backward(g)
    assert '_u2mkd_wfrag4' in conv.kernel.__dict__
    km = F.build_kmap(_dev(coords), (1,) * 3, (3,) * 3, (1,) * 3)
    sch = km.schedule(False)
    tbl = torch.empty(27, km.n_out, dtype=torch.int32, device='cuda')
    tbl[:, sch.order.long()] = sch.nbr_s
    want, mag = _conv_f64(x.detach(), conv.kernel.detach(), tbl, False)
    assert bool(((y.F.detach().double() - want).abs() <= mag * 2.0 ** -20).all())
    wantg, magg = _conv_f64(g, conv.kernel.detach(), tbl, True)
    assert bool(((x.grad.double() - wantg).abs() <= magg * 2.0 ** -20).all())
    opt = torch.optim.SGD(conv.parameters(), lr=0.1)
    opt.step()
    y2 = conv(torchsparse.SparseTensor(x.detach(), _dev(coords)))
    want2, mag2 = _conv_f64(x.detach(), conv.kernel.detach(), tbl, False)
    assert bool(((y2.F.detach().double() - want2).abs() <= mag2 * 2.0 ** -20).all())


@pytest.mark.parametrize('cin,cout', [(256, 256), (512, 128), (96, 192), (128, 32), (32, 96)])
@pytest.mark.parametrize('spread', ['unit', 'rows'])
def test_f16x2_pair_schedule_matches_float64(F, cin, cout, spread):
    """The pair-schedule kernel (wide / strided layers) in f16x2 arithmetic: every 32-channel step of a gathered row carries its
    own scale, so magnitudes may also differ ALONG a row; forward (swap = False) and the swapped-role walk (input gradient)."""
    from u2mkd_amd import _lib as L
    lib = L.load()
    if not lib.u2mkd_conv_pairs_f16x2_supported(cin, cout):
        pytest.skip('U2MKD_CONV_ARITH overrides the default')
    coords = synth_batch(12000, 1, seed=3)['coords']
    km = F.build_kmap(_dev(coords), (1,) * 3, (3,) * 3, (1,) * 3)
    n, k = km.n_out, 27
    ps = km.pair_schedule()
    sch = km.schedule(False)
    tbl = torch.empty(k, n, dtype=torch.int32, device='cuda')
    tbl[:, sch.order.long()] = sch.nbr_s
    torch.manual_seed(6)
    w = torch.randn(k, cin, cout, device='cuda') / (k * cin) ** 0.5 * 1e-4
    for flip in (False, True):
        ca, cb = (cout, cin) if flip else (cin, cout)
        x = torch.randn(n, ca, device='cuda')
        if spread == 'rows':      # per row AND per 32-channel segment magnitudes, 1e-12 .. 1e12; zero rows
            x = x * torch.pow(10.0, torch.randint(-12, 13, (n, 1), device='cuda').float())
            seg = torch.pow(10.0, torch.randint(-6, 7, (n, ca // 32), device='cuda').float()).repeat_interleave(32, dim=1)
            x = x * seg
            x[torch.rand(n, device='cuda') < 0.1] = 0.0
        wf = F._weight_layout(w, not flip, True, arith=4)
        got = torch.full((n, cb), float('nan'), device='cuda')
        ps.run(x, wf, cb, flip, got, fragments=2)
        want, mag = _conv_f64(x, w, tbl, flip)
        assert bool(torch.isfinite(got).all())
        err = (got.double() - want).abs()
        bound = mag * 2.0 ** -20 + 1e-300
        assert bool((err <= bound).all()), float((err / bound).max())
        again = torch.empty_like(got)
        ps.run(x, wf, cb, flip, again, fragments=2)
        assert torch.equal(got, again)
        if spread == 'unit':
            w3 = F._weight_layout(w, not flip, True, arith=0)
            x3 = torch.empty_like(got)
            ps.run(x, w3, cb, flip, x3, fragments=True)
            e3 = (x3.double() - want).abs()
            assert float(err.max()) <= 4.0 * float(e3.max()) + 1e-30


@pytest.mark.parametrize('n,cin,cout,bias', [(5000, 32, 256, True), (80000, 256, 128, True), (777, 128, 96, False), (63, 64, 64, True)])
def test_f16x2_linear_matches_float64(F, n, cin, cout, bias):
    """nn.Linear on the dense mode of the pair kernel, f16x2: y = x W^T + b and the input gradient g W against float64."""
    from u2mkd_amd import _lib as L
    if not L.load().u2mkd_conv_pairs_f16x2_supported(cin, cout):
        pytest.skip('U2MKD_CONV_ARITH overrides the default')
    torch.manual_seed(9)
    lin = torch.nn.Linear(cin, cout, bias=bias).cuda()
    x = torch.randn(n, cin, device='cuda') * torch.pow(10.0, torch.randint(-8, 9, (n, 1), device='cuda').float())
    y = F._dense_x3(x, lin.weight, True, lin.bias)
    want = x.double() @ lin.weight.double().t()
    mag = x.double().abs() @ lin.weight.double().abs().t()
    if bias:
        want = want + lin.bias.double()
        mag = mag + lin.bias.double().abs()
    assert bool(((y.double() - want).abs() <= mag * 2.0 ** -20).all())
    g = torch.randn(n, cout, device='cuda')
    dx = F._dense_x3(g, lin.weight, False)
    wantg = g.double() @ lin.weight.double()
    magg = g.double().abs() @ lin.weight.double().abs()
    assert bool(((dx.double() - wantg).abs() <= magg * 2.0 ** -20).all())
    assert '_u2mkd_wfrag4' in lin.weight.__dict__
