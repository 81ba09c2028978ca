"""csrc/lovasz.hip (the fused Lovasz-softmax) against the oracle's line-by-line restatement of core/criterions.py:40-101 and
against the package's torch formulation; the fused devoxelise-backward plan against the torch ops it replaced."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from u2mkd_amd import _lib
    _lib.load()
    return _lib


def _case(n, c, seed, ignore_frac=0.2, absent=()):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(n, c, generator=g) * 2
    labels = torch.randint(1, c, (n,), generator=g)
    for a in absent:
        labels[labels == a] = 1 if a != 1 else 2
    labels[torch.rand(n, generator=g) < ignore_frac] = 0
    return logits, labels


@pytest.mark.parametrize('n,c,seed,absent', [(1, 5, 0, ()), (257, 17, 1, (3, 9)), (5000, 17, 2, ()), (80000, 17, 3, (16,)), (300, 2, 4, ())])
def test_fused_lovasz_equals_the_reference_formulation(hip, n, c, seed, absent):
    from oracle import spvcnn_ref as O
    from u2mkd_amd.losses import Lovasz_softmax, lovasz_softmax_flat
    logits, labels = _case(n, c, seed, absent=absent)
    # oracle: the reference's compacting formulation on the CPU (fp32, as the reference runs it)
    x64 = logits.clone().requires_grad_(True)
    valid = labels != 0
    want = O.lovasz_softmax_flat(torch.softmax(x64, 1)[valid], labels[valid]) if bool(valid.any()) else x64.sum() * 0
    want.backward()
    xg = logits.cuda().requires_grad_(True)
    got = Lovasz_softmax(ignore_index=0)(torch.softmax(xg, 1), labels.cuda())
    got.backward()
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-5 * max(1.0, abs(float(want.detach()))), (float(got.detach()), float(want.detach()))
    scale = float(x64.grad.abs().max()) + 1e-12
    assert float((xg.grad.cpu() - x64.grad).abs().max()) <= 1e-4 * scale + 1e-9
    # the package's torch formulation on the same device: same value to rounding, same gradient
    xt = logits.cuda().requires_grad_(True)
    ref = lovasz_softmax_flat(torch.softmax(xt, 1), labels.cuda(), labels.cuda() != 0)
    ref.backward()
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, abs(float(ref.detach())))
    assert float((xg.grad - xt.grad).abs().max()) <= 2e-6 * float(xt.grad.abs().max()) + 1e-10
    # deterministic
    x2 = logits.cuda().requires_grad_(True)
    again = Lovasz_softmax(ignore_index=0)(torch.softmax(x2, 1), labels.cuda())
    again.backward()
    assert torch.equal(again, got) and torch.equal(x2.grad, xg.grad)


def test_fused_lovasz_all_rows_ignored(hip):
    from u2mkd_amd.losses import Lovasz_softmax
    x = torch.randn(100, 17, device='cuda', requires_grad=True)
    loss = Lovasz_softmax(ignore_index=0)(torch.softmax(x, 1), torch.zeros(100, dtype=torch.long, device='cuda'))
    loss.backward()
    assert float(loss.detach()) == 0.0 and float(x.grad.abs().max()) == 0.0


@pytest.mark.parametrize('n,nv,seed', [(0, 10, 0), (1, 1, 1), (3000, 700, 2), (80000, 75000, 3)])
def test_devoxelize_plan_equals_the_torch_ops(hip, n, nv, seed):
    from u2mkd_amd import _lib as L
    from u2mkd_amd.torchsparse.nn import functional as F
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(-1, nv, (n, 8), generator=g, dtype=torch.int32).cuda()
    w = torch.rand(n, 8, generator=g).cuda()
    w[torch.rand(n, 8, generator=g).cuda() < 0.2] = 0.
    w[idx < 0] = 0.
    erow = torch.empty(8 * n, dtype=torch.int32, device='cuda')
    ew = torch.empty(8 * n, dtype=torch.float32, device='cuda')
    seg = torch.empty(nv + 1, dtype=torch.int32, device='cuda')
    ws = torch.empty(max(L.load().u2mkd_devoxelize_plan_workspace_bytes(n, nv), 16), dtype=torch.uint8, device='cuda')
    L.call('u2mkd_devoxelize_plan', L.ptr(idx), L.ptr(w), n, nv, L.ptr(ws), L.ptr(erow), L.ptr(ew), L.ptr(seg), L.stream())
    keys = torch.where(w != 0, idx, -1).view(-1)
    order, seg_ref = F._csr_by_destination(keys, nv)
    live = int(seg_ref[-1])
    assert torch.equal(seg, seg_ref)
    assert torch.equal(erow[:live], (order >> 3)[:live])
    assert torch.equal(ew[:live], w.view(-1)[order.long()][:live])


@pytest.mark.parametrize('n,c,frac_ignored', [(80000, 17, 0.2), (1000, 17, 0.0), (257, 5, 0.5), (3, 17, 0.0), (5000, 33, 0.9)])
def test_fused_cross_entropy_equals_torchs(hip, n, c, frac_ignored):
    """losses._CrossEntropyFunction (csrc/lovasz.hip ce_forward / ce_backward: lse per row, fixed-order sums, one backward pass)
    against nn.CrossEntropyLoss(ignore_index=0) -- core/criterions.py:167-174 -- in float64: value and gradient, ignored rows
    with zero gradient; the MixLovaszCrossEntropy module with and without the fused pass."""
    from u2mkd_amd import losses
    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * 3).cuda()
    y = torch.randint(1, c, (n,), generator=g)
    y[torch.rand(n, generator=g) < frac_ignored] = 0
    y = y.cuda()
    xa = x.clone().requires_grad_(True)
    got = losses._CrossEntropyFunction.apply(xa, y, 0)
    (got * 1.7).backward()
    xd = x.double().clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(xd, y, ignore_index=0)
    (want * 1.7).backward()
    assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
    assert float((xa.grad.double() - xd.grad).abs().max()) <= 1e-6 * float(xd.grad.abs().max()) + 1e-12
    assert float(xa.grad[y == 0].abs().max() if bool((y == 0).any()) else 0.0) == 0.0
    crit = losses.MixLovaszCrossEntropy(ignore_index=0)
    xb, xc = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = crit(xb, y)
    a.backward()
    old, losses._FUSED_CE = losses._FUSED_CE, False
    try:
        b = crit(xc, y)
        b.backward()
    finally:
        losses._FUSED_CE = old
    assert abs(float(a) - float(b)) <= 1e-5 * max(1.0, abs(float(b)))
    assert float((xb.grad - xc.grad).abs().max()) <= 1e-6 * float(xc.grad.abs().max()) + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('n,c,indexed', [(80000, 17, True), (777, 17, False), (65, 5, True), (1, 3, False)])
def test_fused_kl_matches_torch(n, c, indexed, monkeypatch):
    """losses.kl_div_logits (csrc/lovasz.hip: kl_forward / kl_backward, the teacher's rows re-indexed inside the pass) against
    nn.KLDivLoss(reduction='batchmean')(log_softmax(s), softmax(t[index])): value and the gradient of the student's logits."""
    from u2mkd_amd import losses
    torch.manual_seed(n + c)
    s0 = (3.0 * torch.randn(n, c, device='cuda'))
    t = (3.0 * torch.randn(n + 13 if indexed else n, c, device='cuda'))
    t[0, 0] = 80.0                      # a row whose other probabilities underflow to 0 (xlogy's 0 log 0 = 0)
    idx = torch.randint(0, t.shape[0], (n,), device='cuda') if indexed else None
    crit = torch.nn.KLDivLoss(reduction='batchmean')
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(losses, '_FUSED_KL', fused)
        s = s0.clone().requires_grad_(True)
        loss = losses.kl_div_logits(s, t, idx, crit)
        (2.5 * loss).backward()
        out[fused] = (float(loss), s.grad.clone())
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * max(1.0, abs(out[False][0])), (out[True][0], out[False][0])
    assert float((out[True][1] - out[False][1]).abs().max()) <= 1e-6 * max(float(out[False][1].abs().max()), 1e-12) + 1e-9
