"""The decoder's bilinear up-sampling (+ skip add) on csrc/pixhead.hip against F.interpolate(mode='bilinear',
align_corners=True) (core/models/image_branch/swiftnet.py: `upsample = lambda x, size: F.interpolate(...)`): the forward
follows torch's fp32 index arithmetic (equal up to the contraction of the four-tap sum into fused multiply-adds), the backward is a gather in a fixed order (torch scatters with float atomics), so
two runs give the same bits."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,size', [((2, 5, 2, 4), (4, 7)), ((3, 16, 8, 14), (16, 28)), ((2, 8, 29, 50), (57, 100)),
                                        ((1, 4, 57, 100), (113, 200)), ((1, 3, 7, 9), (7, 9)), ((2, 4, 1, 1), (3, 5)),
                                        ((1, 2, 16, 28), (57, 100))])
@pytest.mark.parametrize('with_skip', [False, True])
def test_up_bilinear_matches_interpolate(hip, monkeypatch, shape, size, with_skip):
    from u2mkd_amd import camera
    monkeypatch.setattr(camera, '_UP_HIP', True)
    torch.manual_seed(3)
    x = torch.randn(shape, device='cuda', requires_grad=True)
    skip = torch.randn(shape[0], shape[1], *size, device='cuda', requires_grad=True) if with_skip else None
    g = torch.randn(shape[0], shape[1], *size, device='cuda')
    assert camera._up_taps(shape[2], size[0], x.device)[1] is not None
    y = camera._up(x, size, skip)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith('_UpBilinearFunction')
    want = F.interpolate(x, size, mode='bilinear', align_corners=True)
    if with_skip:
        want = want + skip
    err = float((y.detach() - want.detach()).abs().max())
    print('UP-BILINEAR', shape, size, 'forward max abs diff to torch %.2e' % err)
    assert err <= 1e-6 * float(want.detach().abs().max()), err
    grads = torch.autograd.grad(y, [x] + ([skip] if with_skip else []), g)
    xt = x.detach().clone().requires_grad_(True)
    gxt, = torch.autograd.grad(F.interpolate(xt, size, mode='bilinear', align_corners=True), xt, g)
    err = float((grads[0] - gxt).abs().max() / gxt.abs().max().clamp_min(1e-30))
    assert err < 2e-6, err                                  # torch's own (atomic-order dependent) fp32 backward
    x64 = x.detach().double().requires_grad_(True)
    gx64, = torch.autograd.grad(F.interpolate(x64, size, mode='bilinear', align_corners=True), x64, g.double())
    err = float((grads[0].double() - gx64).abs().max() / gx64.abs().max().clamp_min(1e-30))
    assert err < 2e-5, err                                  # fp64 moves the source indices themselves by ~1e-6
    if with_skip:
        assert torch.equal(grads[1], g)
    again = torch.autograd.grad(camera._up(x, size, skip), x, g)[0]
    assert torch.equal(again, grads[0])


def test_up_bilinear_large_factor_falls_back_to_torch(hip, monkeypatch):
    from u2mkd_amd import camera
    monkeypatch.setattr(camera, '_UP_HIP', True)
    x = torch.randn(1, 2, 3, 4, device='cuda', requires_grad=True)
    assert camera._up_taps(3, 64, x.device)[1] is None
    y = camera._up(x, (64, 80))
    assert torch.equal(y, F.interpolate(x, (64, 80), mode='bilinear', align_corners=True)) and 'Upsample' in type(y.grad_fn).__name__


def test_up_bilinear_is_the_path_under_deterministic_algorithms(hip, monkeypatch):
    """torch's bilinear backward raises under torch.use_deterministic_algorithms(True); the decoder then runs on the
    gathering kernels (U2MKD_UP_BILINEAR=auto)."""
    from u2mkd_amd import camera
    monkeypatch.setattr(camera, '_UP_HIP', None)
    x = torch.randn(1, 2, 8, 14, device='cuda', requires_grad=True)
    assert 'UpBilinear' not in type(camera._up(x, (16, 28)).grad_fn).__name__
    torch.use_deterministic_algorithms(True)
    try:
        y = camera._up(x, (16, 28))
        assert 'UpBilinear' in type(y.grad_fn).__name__
        y.sum().backward()
    finally:
        torch.use_deterministic_algorithms(False)
    assert x.grad is not None
