"""BF16 STORAGE end to end (BASELINE.json configs[4]: "... bf16, 8 x MI355X"; the reference runs torchsparse's conv /
voxelize / devoxelize in half under amp -- custom_fwd(cast_inputs=torch.half), SURVEY.md Appendix A-6,
core/nusc_trainers.py:157-158,285 -- and nn.BatchNorm1d passes half rows through).

Per operator, under ``torch.autocast('cuda', bfloat16)``: outputs and gradients are bf16 rows, and against the FP32
ORACLE (oracle.ts_ref / torch CPU) evaluated on the same bf16-ROUNDED inputs the error stays within 2^-7 of the
tensor's magnitude -- what is left is one rounding of every stored value (2^-9) plus, where the pair schedule adds bf16
scratch rows, the rounding of each offset's partial product.  Weight / parameter gradients are fp32.
Then the models: SPVCNN's logits under autocast against its own fp32 run, and that rows really travel in bf16."""
import numpy as np
import pytest
import torch

from oracle import ts_ref as R
from u2mkd_amd.synth import synth_batch

pytestmark = pytest.mark.gpu

TOL = 2.0 ** -7


@pytest.fixture(scope='module')
def F(hip):
    from u2mkd_amd.torchsparse.nn import functional as F
    return F


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _err(got, want):
    """max |got - want| relative to the magnitude of want"""
    want = want.double() if isinstance(want, torch.Tensor) else torch.from_numpy(want).double()
    return float((got.double().cpu() - want).abs().max() / (want.abs().max() + 1e-12))


def _r(t):
    """bf16-rounded copy (fp32 values that are exactly representable in bf16)"""
    return t.bfloat16().float()


def _amp():
    return torch.autocast('cuda', dtype=torch.bfloat16)


# (64,64) (96,128): the tile kernel; the rest: the pair schedule's bf16 form, incl. the 32-channel-step tail (96, 160)
@pytest.mark.parametrize('cin,cout', [(64, 64), (96, 128), (128, 128), (256, 192), (96, 96), (160, 64), (512, 256)])
@pytest.mark.parametrize('kind', ['subm', 'down', 'up'])
def test_conv_bf16_rows_against_the_fp32_oracle(F, cin, cout, kind):
    coords = synth_batch(2500, 2, 7)['coords']
    torch.manual_seed(cin + cout)
    ks, st_ = (3, 1) if kind == 'subm' else (2, 2)
    nbmaps, nbsizes, oc, _ = R.build_kmap(coords, 1, ks, st_)
    km = F.build_kmap(_dev(coords), (1,) * 3, (ks,) * 3, (st_,) * 3)
    sizes = (len(coords), len(oc))
    transposed = kind == 'up'
    n_in, n_out = (sizes[1], sizes[0]) if transposed else sizes
    x = _r(torch.randn(n_in, cin))
    w = _r(torch.randn(ks ** 3, cin, cout) / (ks ** 3 * cin) ** 0.5)
    g = _r(torch.randn(n_out, cout))
    want = R.conv_forward(x, w, nbmaps, nbsizes, sizes, transposed=transposed)
    wgi, wgw = R.conv_backward(x, w, g, nbmaps, nbsizes, transposed=transposed)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    with _amp():
        out = F.ConvolutionFunction.apply(xd, wd, km, transposed)
    assert out.dtype == torch.bfloat16 and bool(torch.isfinite(out.float()).all())
    assert _err(out, want) < TOL
    out.backward(g.cuda().bfloat16())
    assert xd.grad.dtype == torch.float32 and wd.grad.dtype == torch.float32      # the dtypes of the leaves
    assert _err(xd.grad, wgi) < TOL
    assert _err(wd.grad, wgw) < TOL
    # deterministic
    with _amp():
        again = F.ConvolutionFunction.apply(xd, wd, km, transposed)
    assert torch.equal(out, again)


def test_conv_small_channel_counts_round_the_fp32_result(F):
    """the 4-channel stem has no bf16 kernel: fp32 rows in, fp32 kernel, ONE rounding of the result"""
    coords = synth_batch(2000, 1, 3)['coords']
    km = F.build_kmap(_dev(coords), (1,) * 3, (3,) * 3, (1,) * 3)
    torch.manual_seed(0)
    x, w = torch.randn(len(coords), 4, device='cuda'), torch.randn(27, 4, 32, device='cuda') * 0.1
    ref = F.ConvolutionFunction.apply(x, w, km, False)
    with _amp():
        out = F.ConvolutionFunction.apply(x, w, km, False)
    assert out.dtype == torch.bfloat16 and torch.equal(out, ref.bfloat16())


@pytest.mark.parametrize('n,cin,cout,bias', [(5000, 32, 256, True), (30000, 256, 128, True), (777, 128, 96, False), (64, 96, 32, True)])
def test_linear_bf16_rows(F, n, cin, cout, bias):
    torch.manual_seed(n)
    x, w = _r(torch.randn(n, cin)), _r(torch.randn(cout, cin) / cin ** 0.5)
    b = torch.randn(cout) if bias else None
    g = _r(torch.randn(n, cout))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(g)
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    with _amp():
        y = F.linear(xd, wd, bd)
    assert y.dtype == torch.bfloat16 and _err(y, yr.detach()) < TOL
    y.backward(g.cuda().bfloat16())
    assert _err(xd.grad, xr.grad) < TOL and _err(wd.grad, wr.grad) < TOL
    if bias:
        assert bd.grad.dtype == torch.float32 and _err(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize('n,c', [(5000, 32), (40000, 96), (777, 256)])
@pytest.mark.parametrize('mode', ['plain', 'relu', 'res'])
def test_batch_norm_bf16_rows(F, n, c, mode):
    torch.manual_seed(c)
    x = _r(torch.randn(n, c) * 2 + 0.5)
    res = _r(torch.randn(n, c)) if mode == 'res' else None
    g = _r(torch.randn(n, c))
    bn_r = torch.nn.BatchNorm1d(c).double()
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5)
        bn_r.bias.uniform_(-0.5, 0.5)
    bn_d = torch.nn.BatchNorm1d(c)
    bn_d.load_state_dict({k: v.float() for k, v in bn_r.state_dict().items()})
    bn_d.cuda()
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if res is not None else None
    yr = bn_r(xr)
    if mode == 'res':
        yr = torch.relu(yr + rr)
    elif mode == 'relu':
        yr = torch.relu(yr)
    yr.backward(g.double())
    xd = x.cuda().bfloat16().requires_grad_(True)
    rd = res.cuda().bfloat16().requires_grad_(True) if res is not None else None
    with _amp():
        y = F.batch_norm(xd, bn_d, mode != 'plain', rd)
    assert y.dtype == torch.bfloat16 and _err(y, yr.detach()) < TOL
    y.backward(g.cuda().bfloat16())
    assert xd.grad.dtype == torch.bfloat16
    # the ReLU mask flips only where |pre-activation| is below the bf16 rounding of y: the masked gradient rows differ
    # there by one element of g; a relative bound over the tensor holds them
    assert _err(xd.grad, xr.grad) < 4 * TOL
    if rd is not None:
        assert _err(rd.grad, rr.grad) < 4 * TOL
    assert bn_d.weight.grad.dtype == torch.float32
    assert _err(bn_d.weight.grad, bn_r.weight.grad) < TOL and _err(bn_d.bias.grad, bn_r.bias.grad) < TOL
    # running statistics from the bf16-rounded rows, in fp32
    assert _err(bn_d.running_mean, bn_r.running_mean) < 1e-5 and _err(bn_d.running_var, bn_r.running_var) < 1e-5
    # eval mode
    bn_d.eval(); bn_r.eval()
    with _amp():
        ye = F.batch_norm(x.cuda().bfloat16(), bn_d, mode != 'plain', rd.detach() if rd is not None else None)
    yre = bn_r(x.double())
    yre = torch.relu(yre + res.double()) if mode == 'res' else (torch.relu(yre) if mode == 'relu' else yre)
    assert ye.dtype == torch.bfloat16 and _err(ye, yre) < TOL


def test_point_voxel_transfers_bf16_rows(F):
    b = synth_batch(6000, 2, 11)
    coords = b['coords']
    c = 64
    torch.manual_seed(1)
    # voxelize: points -> stride-2 voxels
    fl = np.concatenate([coords[:, :3] // 2 * 2, coords[:, 3:]], 1).astype(np.int32)
    uniq, inv = np.unique(fl, axis=0, return_inverse=True)
    idx = torch.from_numpy(inv.astype(np.int32))
    counts = torch.bincount(idx.long(), minlength=len(uniq)).int()
    feats = _r(torch.randn(len(coords), c))
    g = _r(torch.randn(len(uniq), c))
    want = R.voxelize_forward(feats, idx, counts)
    fd = feats.cuda().requires_grad_(True)
    with _amp():
        out = F.spvoxelize(fd, idx.cuda(), counts.cuda())
    assert out.dtype == torch.bfloat16 and _err(out, want) < TOL
    out.backward(g.cuda().bfloat16())
    assert _err(fd.grad, R.voxelize_backward(g, idx, counts, len(coords))) < TOL
    # the 4-channel coordinate means stay fp32 (SphereFormer quantises them into windows)
    with _amp():
        xyz = F.spvoxelize(torch.randn(len(coords), 4, device='cuda'), idx.cuda(), counts.cuda())
    assert xyz.dtype == torch.float32
    # devoxelize: 8-corner gather with random weights, some corners missing
    nv = len(uniq)
    i8 = torch.randint(-1, nv, (len(coords), 8), dtype=torch.int32)
    w8 = torch.rand(len(coords), 8) * (i8 >= 0)
    vf = _r(torch.randn(nv, c))
    gp = _r(torch.randn(len(coords), c))
    vd = vf.cuda().requires_grad_(True)
    with _amp():
        y = F.spdevoxelize(vd, i8.cuda(), w8.cuda())
    assert y.dtype == torch.bfloat16 and _err(y, R.devoxelize_forward(vf, i8, w8)) < TOL
    y.backward(gp.cuda().bfloat16())
    assert _err(vd.grad, R.devoxelize_backward(gp, i8, w8, nv)) < TOL


def _spvcnn(cr, seed=0):
    from u2mkd_amd import lidar
    torch.manual_seed(seed)
    m = lidar.SPVCNN(cr=cr, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
    m.dropout.p = 0.0
    return m


def test_spvcnn_rows_travel_in_bf16_and_logits_stay_close_to_fp32(F, monkeypatch):
    from u2mkd_amd import torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    b = synth_batch(6000, 1, seed=5)
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    m = _spvcnn(1.0)
    ref = m({'lidar': ts.SparseTensor(feats, coords)})['x_vox'].detach()
    seen = {'conv_bf16': 0, 'conv_f32': 0, 'bn_bf16': 0, 'bn_f32': 0}
    real_call = F.L.call

    def spy(name, *a):
        if name.startswith('u2mkd_conv_forward_tiles') or name.startswith('u2mkd_conv_forward_pairs'):
            seen['conv_bf16' if name.endswith('_bf16') else 'conv_f32'] += 1
        if name.startswith('u2mkd_bn_train_forward_res'):
            seen['bn_bf16' if name.endswith('_bf16') else 'bn_f32'] += 1
        return real_call(name, *a)
    monkeypatch.setattr(F.L, 'call', spy)
    with _amp():
        out = m({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
        loss = MixLovaszCrossEntropy(ignore_index=0)(out, labels)
    loss.backward()
    monkeypatch.undo()
    # every conv but the 4-channel stem and every BatchNorm ran on bf16 rows
    assert seen['conv_bf16'] >= 30 and seen['conv_f32'] <= 1, seen
    assert seen['bn_bf16'] >= 40 and seen['bn_f32'] == 0, seen
    assert bool(torch.isfinite(out.float()).all())
    # ~50 bf16 layers deep: the logits keep two significant digits of the fp32 run
    d = (out.float() - ref).abs()
    assert float(d.max()) < 0.15 * float(ref.abs().max()) and float(d.median()) < 0.02 * float(ref.abs().max()), \
        (float(d.max()), float(d.median()), float(ref.abs().max()))
    for n, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()), n
