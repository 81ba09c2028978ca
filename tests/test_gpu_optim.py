"""optim.FusedSGD (csrc/optim.hip: torch.optim.SGD's update over all parameters of a group in one launch) against
torch.optim.SGD on the device -- the reference's optimizer, core/builder.py:663-669 -- BIT FOR BIT: parameters and momentum
buffers after every step, for the shipped setting (momentum 0.9, weight decay 1e-4, Nesterov), changing learning rates,
parameters that receive no gradient in some steps, odd sizes and misaligned gradient views (a DDP bucket's), the other
flag combinations, two parameter groups, a state_dict round trip in both directions, and the fall-back on what the kernel does
not cover."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(27, 64, 64), (1,), (3,), (4097,), (17, 96), (8192,), (5, 7, 3, 3), (12289,), (64,), (2, 4096)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 10 ** float(torch.randint(-3, 3, (1,), generator=g))).cuda()) for s in SHAPES]


def _grads(ps, seed, skip=(), misalign=False):
    g = torch.Generator().manual_seed(1000 + seed)
    for i, p in enumerate(ps):
        if i in skip:
            p.grad = None
            continue
        v = torch.randn(p.numel() + 1, generator=g).cuda() * 0.3
        p.grad = (v[1:] if misalign else v[:-1].clone()).view_as(p)          # (misalign: a 4-byte offset into a larger buffer)


def _run(opt_cls, kw, steps, groups=False, misalign=False):
    ps = _params(3)
    if groups:
        opt = opt_cls([dict(params=ps[:4], lr=0.24), dict(params=ps[4:], lr=0.024)], lr=0.24, **kw)
    else:
        opt = opt_cls(ps, lr=0.24, **kw)
    out = []
    for k in range(steps):
        for g in opt.param_groups:
            g['lr'] = np.float64(g['lr'] * 0.9 + 1e-3 * k) if k % 2 else float(g['lr'] * 0.9 + 1e-3 * k)      # (LambdaLR hands over numpy floats)
        _grads(ps, k, skip=((1, 5) if k == 0 else (2,) if k == 2 else ()), misalign=misalign)
        opt.step()
        out.append([p.detach().clone() for p in ps] + [opt.state[p]['momentum_buffer'].clone() if 'momentum_buffer' in opt.state.get(p, {})
                                                        else None for p in ps])
        opt.zero_grad()
        assert all(p.grad is None for p in ps)
    return out, opt, ps


@pytest.mark.parametrize('kw', [dict(momentum=0.9, weight_decay=1e-4, nesterov=True), dict(momentum=0.9, weight_decay=0.0, nesterov=False),
                                dict(momentum=0.0, weight_decay=5e-4), dict(momentum=0.5, weight_decay=1e-2, nesterov=True)])
@pytest.mark.parametrize('groups,misalign', [(False, False), (True, False), (False, True)])
def test_fused_sgd_equals_torch_sgd_bit_for_bit(hip, kw, groups, misalign):
    from u2mkd_amd.optim import FusedSGD
    want, _, _ = _run(torch.optim.SGD, kw, 5, groups, misalign)
    got, opt, _ = _run(FusedSGD, kw, 5, groups, misalign)
    assert opt._fused_groups, 'the fused path did not run'
    for k, (a, b) in enumerate(zip(want, got)):
        for i, (x, y) in enumerate(zip(a, b)):
            assert (x is None) == (y is None), (k, i)
            if x is not None:
                assert torch.equal(x, y), (k, i, float((x - y).abs().max()))


def test_fused_sgd_state_dict_round_trips_with_torch_sgd(hip):
    from u2mkd_amd.optim import FusedSGD
    kw = dict(momentum=0.9, weight_decay=1e-4, nesterov=True)
    _, fused, ps_f = _run(FusedSGD, kw, 3)
    _, plain, ps_p = _run(torch.optim.SGD, kw, 3)
    sd_f, sd_p = fused.state_dict(), plain.state_dict()
    assert sd_f['param_groups'] == sd_p['param_groups'] and sd_f['state'].keys() == sd_p['state'].keys()
    for k in sd_p['state']:
        assert torch.equal(sd_f['state'][k]['momentum_buffer'], sd_p['state'][k]['momentum_buffer'])
    # a torch checkpoint into the fused optimizer and the other way round, then two more steps each: same parameters
    ps_a, ps_b = _params(3), _params(3)
    for a, b, src in zip(ps_a, ps_b, ps_p):
        a.data.copy_(src.data); b.data.copy_(src.data)
    oa, ob = FusedSGD(ps_a, lr=0.24, **kw), torch.optim.SGD(ps_b, lr=0.24, **kw)
    oa.load_state_dict(copy.deepcopy(sd_p)); ob.load_state_dict(copy.deepcopy(sd_f))
    for k in range(3, 5):
        for o, ps in ((oa, ps_a), (ob, ps_b)):
            _grads(ps, k)
            o.step()
    for a, b in zip(ps_a, ps_b):
        assert torch.equal(a, b)
    assert all(oa.state[p]['momentum_buffer'].data_ptr() == fb.data_ptr() for p, fb in zip(ps_a, oa._fused_groups[0].bufs))


def test_fused_sgd_falls_back_to_torch_where_the_kernel_does_not_apply(hip):
    from u2mkd_amd.optim import FusedSGD
    ps = _params(5)
    opt = FusedSGD(ps, lr=0.1, momentum=0.9, dampening=0.5)          # dampening: torch's own step
    ref = [p.detach().clone() for p in ps]
    twin = [torch.nn.Parameter(r.clone()) for r in ref]
    plain = torch.optim.SGD(twin, lr=0.1, momentum=0.9, dampening=0.5)
    for k in range(2):
        _grads(ps, k); _grads(twin, k)
        opt.step(); plain.step()
    assert not opt._fused_groups
    for a, b in zip(ps, twin):
        assert torch.equal(a, b)
    cpu = [torch.nn.Parameter(torch.randn(5))]
    o2 = FusedSGD(cpu, lr=0.1, momentum=0.9)
    cpu[0].grad = torch.ones(5)
    o2.step()                                                        # CPU parameters: torch's own step, no error
    assert not o2._fused_groups


def test_global_step_hooks_fire_once_per_fused_step(hip):
    from torch.optim.optimizer import register_optimizer_step_post_hook
    from u2mkd_amd.optim import FusedSGD
    calls = []
    h = register_optimizer_step_post_hook(lambda *a: calls.append(1))
    try:
        ps = _params(7)
        opt = FusedSGD(ps, lr=0.1, momentum=0.9, nesterov=True)
        _grads(ps, 0)
        opt.step()
        o2 = FusedSGD(_params(8), lr=0.1, momentum=0.9, dampening=0.1)
        _grads(o2.param_groups[0]['params'], 0)
        o2.step()
    finally:
        h.remove()
    assert len(calls) == 2
