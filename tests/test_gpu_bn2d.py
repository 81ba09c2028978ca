"""BatchNorm2d (+ReLU, +residual) of the camera branch on csrc/bn2d.hip against torch.nn.functional.batch_norm
in fp64 on the CPU: outputs, all gradients, running statistics; and the SwiftNet-18 branch as a whole with the HIP
BatchNorm against the same branch on torch.nn."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, res, bn, relu, training):
    """fp64 CPU reference; returns y, (dx, dres, dgamma, dbeta) for a fixed upstream gradient, and the buffers"""
    x = x.double().cpu().requires_grad_(True)
    res = res.double().cpu().requires_grad_(True) if res is not None else None
    w = bn.weight.detach().double().cpu().requires_grad_(True)
    b = bn.bias.detach().double().cpu().requires_grad_(True)
    rm, rv = bn.running_mean.double().cpu().clone(), bn.running_var.double().cpu().clone()
    y = F.batch_norm(x, rm, rv, w, b, training, bn.momentum, bn.eps)
    if res is not None:
        y = y + res
    clear = (y.detach().abs() > 1e-4) if relu else torch.ones_like(y, dtype=torch.bool)   # away from the ReLU edge
    if relu:
        y = F.relu(y)
    return x, res, w, b, y, rm, rv, clear


@pytest.mark.parametrize('shape', [(6, 64, 45, 80), (2, 8, 7, 9), (3, 128, 12, 20), (2, 16, 180, 320), (1, 4, 1, 2)])
@pytest.mark.parametrize('relu,with_res', [(False, False), (True, False), (True, True), (False, True)])
def test_bn2d_train_matches_fp64_reference(hip, shape, relu, with_res):
    from u2mkd_amd.camera import BatchNorm2d
    torch.manual_seed(sum(shape))
    bn = BatchNorm2d(shape[1], momentum=0.1).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(shape[1]) + 0.5)
        bn.bias.copy_(torch.randn(shape[1]) * 0.3)
        bn.running_mean.copy_(torch.randn(shape[1]))
        bn.running_var.copy_(torch.rand(shape[1]) + 0.5)
    x = (torch.randn(shape) * 2.0 + 3.0).cuda().requires_grad_(True)        # a mean far from 0: cancellation check
    res = torch.randn(shape).cuda().requires_grad_(True) if with_res else None
    g = torch.randn(shape).cuda()
    xr, rr, wr, br, yr, rm, rv, clear = _ref(x.detach(), res.detach() if with_res else None, bn, relu, True)
    y = bn(x, relu, res)
    assert y.shape == x.shape and y.is_contiguous()
    y.backward(g)
    yr.backward(g.double().cpu())
    scale = float(yr.detach().abs().max())
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5 * scale
    # elements within rounding of the ReLU edge may flip: element-wise gradients are compared where the reference
    # is clear of it (a flipped element moves the per-channel sums by 1 / (b * h * w) of their scale)
    assert float(((x.grad.cpu().double() - xr.grad) * clear).abs().max()) < 2e-4 * max(1.0, float(xr.grad.abs().max()))
    assert float((bn.weight.grad.cpu().double() - wr.grad).abs().max()) < 2e-4 * max(1.0, float(wr.grad.abs().max()))
    assert float((bn.bias.grad.cpu().double() - br.grad).abs().max()) < 2e-4 * max(1.0, float(br.grad.abs().max()))
    if with_res:
        assert float(((res.grad.cpu().double() - rr.grad) * clear).abs().max()) < 1e-5 * max(1.0, float(rr.grad.abs().max()))
    assert float((bn.running_mean.cpu().double() - rm).abs().max()) < 1e-5
    assert float((bn.running_var.cpu().double() - rv).abs().max()) < 1e-4 * max(1.0, float(rv.max()))
    assert int(bn.num_batches_tracked) == 1


def test_bn2d_eval_mode_and_fallbacks(hip):
    from u2mkd_amd.camera import BatchNorm2d, bn_act
    torch.manual_seed(0)
    bn = BatchNorm2d(32).cuda()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(32)); bn.running_var.copy_(torch.rand(32) + 0.5)
        bn.weight.copy_(torch.rand(32) + 0.5); bn.bias.copy_(torch.randn(32))
    bn.eval()
    x = torch.randn(2, 32, 10, 12).cuda().requires_grad_(True)
    res = torch.randn(2, 32, 10, 12).cuda()
    xr, _, wr, br, yr, rm, rv, clear = _ref(x.detach(), res, bn, True, False)
    y = bn(x, True, res)
    y.sum().backward()
    yr.sum().backward()
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 1e-5 * float(yr.abs().max())
    assert float(((x.grad.cpu().double() - xr.grad) * clear).abs().max()) < 1e-5 * float(xr.grad.abs().max())
    assert float((bn.weight.grad.cpu().double() - wr.grad).abs().max()) < 1e-4 * float(wr.grad.abs().max())
    assert int(bn.num_batches_tracked) == 0 and torch.equal(bn.running_mean.cpu().double(), rm)
    # the torch.nn path: CPU tensors, autocast, a converted SyncBatchNorm-like module through bn_act
    cpu = BatchNorm2d(32)
    yc = cpu(torch.randn(2, 32, 4, 4), True, torch.randn(2, 32, 4, 4))
    assert yc.shape == (2, 32, 4, 4) and float(yc.min()) >= 0.0
    plain = torch.nn.BatchNorm2d(32).cuda()
    yp = bn_act(plain, x.detach(), relu=True, residual=res)
    assert float(yp.min()) >= 0.0
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ya = bn(x.detach().bfloat16(), True)
    assert ya.shape == x.shape


def test_swiftnet_encoder_with_hip_batchnorm_matches_fp64_reference(hip, monkeypatch):
    """Stem + layer1 + layer2 (10 BatchNorms: fused ReLU, residual adds, a stride-2 down-sample branch) against the
    same modules in fp64 on the CPU.  (The whole net is not a usable oracle for gradients: the pyramid pooling
    BatchNorms see 2 x 1 x 2 values per channel, and fp32 MIOpen and these kernels both sit 3-5 % from fp64 there.)"""
    from u2mkd_amd import camera
    torch.manual_seed(1)
    monkeypatch.setattr(camera, '_HIP_BN2D', True)
    m = camera.SwiftNetRes18().cuda().train()
    m64 = copy.deepcopy(m).double().cpu()
    x = torch.randn(2, 3, 64, 96, device='cuda')

    def encoder(net, inp):
        h, _ = net.forward_resblock(net.forward_stem(inp), net.layer1)
        return net.forward_resblock(h, net.layer2)[0]
    y, y64 = encoder(m, x), encoder(m64, x.double().cpu())
    w = torch.linspace(-1, 1, y.numel()).view_as(y64)
    (y * w.float().cuda()).sum().backward()
    (y64 * w.double()).sum().backward()
    assert float((y.detach().cpu().double() - y64.detach()).abs().max()) < 2e-4 * float(y64.detach().abs().max())
    for (n, p), q in zip(m.named_parameters(), m64.parameters()):
        if q.grad is None:
            assert p.grad is None, n
            continue
        gate = 2e-3 * max(float(q.grad.abs().max()), 1e-2 * float(w.abs().max()))
        assert float((p.grad.cpu().double() - q.grad).abs().max()) < gate, (n, float((p.grad.cpu().double() - q.grad).abs().max()), gate)


def test_swiftnet_forward_with_hip_batchnorm_matches_torch_batchnorm(hip, monkeypatch):
    from u2mkd_amd import camera
    torch.manual_seed(1)
    m_hip = camera.SwiftNetRes18().cuda().train()
    m_ref = copy.deepcopy(m_hip)
    x = torch.randn(2, 3, 128, 192, device='cuda')
    outs = {}
    for name, m, flag in (('hip', m_hip, True), ('torch', m_ref, False)):
        monkeypatch.setattr(camera, '_HIP_BN2D', flag)
        y = m(x, im_size=(128, 192))
        y.square().mean().backward()
        outs[name] = y.detach()
    rel = float((outs['hip'] - outs['torch']).abs().max() / outs['torch'].abs().max())
    assert rel < 1e-4, rel
    for (n, p), q in zip(m_hip.named_parameters(), m_ref.parameters()):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    for (n, b), c in zip(m_hip.named_buffers(), m_ref.buffers()):
        assert float((b.double() - c.double()).abs().max()) < 1e-4 * max(1.0, float(c.double().abs().max())), n
