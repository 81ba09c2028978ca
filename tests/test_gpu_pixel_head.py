"""The pixel head evaluated at the sampled pixels (u2mkd_amd/pixel_head.py, csrc/pixhead.hip) against the dense
evaluation it replaces -- Feature_Fetch(classifier_pix(upsample(x, image size))), the reference's formulation
(swiftnet.py forward_up, spvcnn_swiftnet18_spformer_tsd_full.py classifier_pix, fusion_blocks.py:257-278) written with
torch.nn ops in fp64: logits, every gradient (map, BatchNorm affine, classifier) and the running statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from u2mkd_amd.synth import synth_kd_batch

pytestmark = pytest.mark.gpu


def _inputs(seed, hw, n_vox=900, batch=2):
    b = synth_kd_batch(n_vox, batch, seed=seed, image_hw=hw)
    s = b['student']
    pc = [torch.from_numpy(c).cuda() for c in s['pixel_coordinates']]
    ms = [torch.from_numpy(m).cuda() for m in s['masks']]
    return pc, ms


def _dense_fp64(x, head, pc, ms, size, ib, ncam):
    """The reference formulation in fp64 with torch.nn ops (grid_sample per camera, later cameras overwrite)."""
    u = F.interpolate(x, size, mode='bilinear', align_corners=True)
    bn = head.norm
    y = F.batch_norm(u, None, None, bn.weight.double(), bn.bias.double(), True, 0.0, bn.eps)
    z = F.conv2d(torch.relu(y), head.conv.weight.double())
    z = z.view(ib, ncam, z.shape[1], size[0], size[1])
    out = []
    for b in range(ib):
        n = ms[b].shape[1]
        res = torch.zeros(n, z.shape[2], dtype=torch.float64, device=x.device)
        for cam in range(ncam):
            grid = pc[b][cam].double().view(1, 1, n, 2)
            smp = F.grid_sample(z[b, cam][None], grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0, :, 0].t()
            res = torch.where(ms[b][cam].unsqueeze(1), smp, res)
        out.append(res)
    return torch.cat(out), u


@pytest.mark.parametrize('hw,low', [((64, 112), (32, 56)), ((90, 150), (45, 75)), ((61, 97), (31, 49))])
def test_sampled_pixel_head_equals_the_dense_evaluation(hip, hw, low):
    from u2mkd_amd import camera
    from u2mkd_amd.pixel_head import sampled_head_applies, sampled_pixel_logits
    torch.manual_seed(5)
    ib, ncam, c, classes = 2, 6, 32, 17
    pc, ms = _inputs(9, hw)
    head = camera.BNReluConv(c, classes, k=1).cuda().train()
    with torch.no_grad():
        head.norm.weight.uniform_(0.5, 1.5)
        head.norm.bias.normal_(0, 0.3)
        head.norm.running_mean.normal_(0, 0.2)
        head.norm.running_var.uniform_(0.5, 2.0)
    rm0, rv0 = head.norm.running_mean.clone(), head.norm.running_var.clone()
    x = (torch.randn(ib * ncam, c, *low, device='cuda') * 1.7 + 3.0).requires_grad_(True)     # mean >> 0: the shifted sums matter
    assert sampled_head_applies(x, head)
    got = sampled_pixel_logits(x, head, pc, ms, hw, ib, ncam)
    g = torch.randn_like(got)
    got.backward(g)
    grads = [x.grad.clone(), head.norm.weight.grad.clone(), head.norm.bias.grad.clone(), head.conv.weight.grad.clone()]
    rm1, rv1 = head.norm.running_mean.clone(), head.norm.running_var.clone()

    xd = x.detach().double().requires_grad_(True)
    for p in head.parameters():
        p.grad = None
    want, u = _dense_fp64(xd, head, pc, ms, hw, ib, ncam)
    want.backward(g.double())
    ref = [xd.grad, head.norm.weight.grad, head.norm.bias.grad, head.conv.weight.grad]
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 2e-5 * scale
    for a, b, name in zip(grads, ref, ('map', 'gamma', 'beta', 'classifier')):
        err = float((a.double() - b.double()).abs().max()) / float(b.abs().max())
        assert err < 5e-5, (name, err)
    # running statistics: nn.BatchNorm2d's update with the statistics of the up-sampled map
    m = head.norm.momentum
    mean = u.detach().mean((0, 2, 3))
    var = u.detach().var((0, 2, 3), unbiased=True)
    assert float((rm1.double() - ((1 - m) * rm0.double() + m * mean)).abs().max()) < 1e-5
    assert float((rv1.double() - ((1 - m) * rv0.double() + m * var)).abs().max()) < 1e-5 * float(var.max())
    assert int(head.norm.num_batches_tracked) == 1

    # evaluation mode: running statistics, no dense term
    head.eval()
    with torch.no_grad():
        got_e = sampled_pixel_logits(x.detach(), head, pc, ms, hw, ib, ncam)
        ue = F.interpolate(x.detach().double(), hw, mode='bilinear', align_corners=True)
        ye = F.batch_norm(ue, head.norm.running_mean.double(), head.norm.running_var.double(), head.norm.weight.double(),
                          head.norm.bias.double(), False, 0.0, head.norm.eps)
        ze = F.conv2d(torch.relu(ye), head.conv.weight.double()).view(ib, ncam, classes, *hw)
        from u2mkd_amd.fusion import c2l_gather_torch
        want_e = c2l_gather_torch(ze, [p.double() for p in pc], ms)
    assert float((got_e.double() - want_e).abs().max()) < 2e-5 * float(want_e.abs().max())


def test_sampled_pixel_head_at_the_full_camera_size(hip):
    """BASELINE.json configs[2]'s literal camera size, 6 x 900 x 1600 with the shipped 128-channel decoder map: the
    sampled head against the dense head evaluated by torch in fp32 on the same map (4.4 GB per full-resolution
    tensor) -- logits, the gradient of the map and of the classifier."""
    from u2mkd_amd import camera
    from u2mkd_amd.fusion import c2l_gather_torch
    from u2mkd_amd.pixel_head import sampled_pixel_logits
    torch.manual_seed(2)
    hw, low, ib, ncam, c, classes = (900, 1600), (450, 800), 1, 6, 128, 17
    pc, ms = _inputs(12, hw, n_vox=40000, batch=1)
    head = camera.BNReluConv(c, classes, k=1).cuda().train()
    with torch.no_grad():
        head.norm.weight.uniform_(0.5, 1.5)
        head.norm.bias.normal_(0, 0.3)
    x = (torch.randn(ib * ncam, c, *low, device='cuda') + 0.5).requires_grad_(True)
    got = sampled_pixel_logits(x, head, pc, ms, hw, ib, ncam)
    g = torch.randn_like(got)
    got.backward(g)
    gx, gw = x.grad.clone(), head.conv.weight.grad.clone()
    x.grad = None
    for p in head.parameters():
        p.grad = None
    u = F.interpolate(x, hw, mode='bilinear', align_corners=True)
    y = F.batch_norm(u, None, None, head.norm.weight, head.norm.bias, True, 0.0, head.norm.eps)
    # (the 1x1 classifier as a matrix product over the pixels: MIOpen's weight gradient of this convolution is WRONG at
    # this size -- its input is 4.4 GB, beyond 32-bit byte offsets; tools/dbg_miopen_large.py)
    z = (torch.relu(y).permute(0, 2, 3, 1).reshape(-1, c) @ head.conv.weight.view(classes, c).t())
    z = z.view(ib * ncam, hw[0], hw[1], classes).permute(0, 3, 1, 2).reshape(ib, ncam, classes, *hw)
    want = c2l_gather_torch(z, pc, ms)
    want.backward(g)
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())
    assert float((gx - x.grad).norm() / x.grad.norm()) < 1e-4
    assert float((gw - head.conv.weight.grad).norm() / head.conv.weight.grad.norm()) < 1e-4


def test_sampled_pixel_head_with_a_sample_no_camera_sees(hip):
    """One sample of the batch is seen by no camera (all masks False) and one camera of the other sample sees nothing:
    those points read zeros in the dense formulation (Feature_Fetch's `zeros` padding); the sampled head must agree in
    logits and gradients (their samples carry zero weight, so they add nothing to the BatchNorm backward sums)."""
    from u2mkd_amd import camera
    from u2mkd_amd.pixel_head import sampled_pixel_logits
    torch.manual_seed(11)
    hw, low, ib, ncam, c, classes = (64, 112), (32, 56), 2, 6, 16, 17
    pc, ms = _inputs(4, hw)
    ms[1][:] = False
    ms[0][3] = False
    head = camera.BNReluConv(c, classes, k=1).cuda().train()
    x = (torch.randn(ib * ncam, c, *low, device='cuda') + 1.0).requires_grad_(True)
    got = sampled_pixel_logits(x, head, pc, ms, hw, ib, ncam)
    n0 = ms[0].shape[1]
    assert float(got[n0:].abs().max()) == 0.0
    g = torch.randn_like(got)
    got.backward(g)
    grads = [x.grad.clone(), head.norm.weight.grad.clone(), head.norm.bias.grad.clone(), head.conv.weight.grad.clone()]
    xd = x.detach().double().requires_grad_(True)
    for p in head.parameters():
        p.grad = None
    want, _ = _dense_fp64(xd, head, pc, ms, hw, ib, ncam)
    want.backward(g.double())
    assert float((got.double() - want).abs().max()) < 2e-5 * float(want.abs().max())
    for a, b, name in zip(grads, [xd.grad, head.norm.weight.grad, head.norm.bias.grad, head.conv.weight.grad], ('map', 'gamma', 'beta', 'classifier')):
        assert float((a.double() - b.double()).abs().max()) < 5e-5 * float(b.abs().max()), name


def test_student_uses_the_sampled_head_and_matches_the_dense_path(hip, monkeypatch):
    """Whole student forward / backward with the sampled head against the same model with it switched off (same
    weights, same batch; BatchNorm buffers put back in between)."""
    from oracle.spvcnn_ref import fill_state_by_name
    from u2mkd_amd import kd, lidar, pixel_head, torchsparse as ts
    b = synth_kd_batch(1200, 2, seed=78, image_hw=(64, 112))
    s = b['student']
    sp = {k: v for k, v in lidar.spformer_kwargs(drop_path_rate=0.0).items() if k not in ('cr', 'in_channel', 'num_classes')}
    model = fill_state_by_name(kd.TSDFull(cr=1.0, cr_t=1.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp),
                               conv2d_he=True).cuda().train()
    ms_ = model.model_s
    ms_.dropout.p = 0.0
    state = {k: v.clone() for k, v in ms_.state_dict().items()}
    used = []
    real = pixel_head.sampled_pixel_logits
    monkeypatch.setattr(kd, 'sampled_pixel_logits', lambda *a: (used.append(1), real(*a))[1])
    outs = []
    for enabled in (False, True, False):          # (the first pass also warms MIOpen's per-shape solver choice up)
        monkeypatch.setattr(pixel_head, '_ENABLED', enabled)
        ms_.load_state_dict(state)
        ms_.zero_grad(set_to_none=True)
        pc = [torch.from_numpy(c).cuda() for c in s['pixel_coordinates']]
        ms = [torch.from_numpy(m).cuda() for m in s['masks']]
        stu = {'lidar': ts.SparseTensor(torch.from_numpy(s['feats']).cuda(), torch.from_numpy(s['coords']).cuda()),
               'images': ((torch.from_numpy(s['images']) / 255.0 - 0.45) / 0.225).permute(0, 1, 4, 2, 3).contiguous().cuda(),
               'pixel_coordinates': pc, 'masks': ms, 'fov_mask': torch.from_numpy(s['fov_mask']).cuda()}
        out = ms_(stu)
        out['x_pix'].square().mean().backward()
        outs.append((out['x_pix'].detach(), ms_.classifier_pix.conv.weight.grad.clone(), ms_.classifier_pix.norm.weight.grad.clone(),
                     ms_.pix_branch.upsample[2].blend_conv.conv.weight.grad.clone(), ms_.pix_branch.conv1.weight.grad.clone(),
                     ms_.classifier_pix.norm.running_var.clone()))
    assert len(used) == 1
    # two dense passes calibrate what run-to-run differences of the library convolutions do to each quantity (the stem's
    # gradient passes through the whole camera branch: 2e-4 .. 2e-3 between two identical dense passes); the sampled pass
    # must sit within that band (one pair is a noisy estimate of it: factor 10) or 1e-4 -- a wrong term would be O(0.1)
    for a, bb, cc, name in zip(outs[1], outs[2], outs[0], ('x_pix', 'classifier', 'gamma', 'last blend', 'stem conv', 'running_var')):
        err = float((a - bb).norm() / bb.norm())
        floor = float((cc - bb).norm() / bb.norm())
        print('PIXHEAD', name, 'sampled vs dense %.2e   dense vs dense %.2e' % (err, floor))
        # (the stem's gradient: two identical dense passes differ by 2e-4 .. 2e-3 -- one pair is a poor estimate of that band)
        assert err < max(5e-3 if name == 'stem conv' else 1e-4, 10.0 * floor), (name, err, floor)


@pytest.mark.parametrize('shape', [(2, 5, 37, 52), (3, 8, 64, 112), (1, 3, 7, 9), (1, 2, 2, 2), (1, 4, 9, 16), (2, 2, 5, 4)])
def test_maxpool_3x3_s2_equals_torch_including_ties(hip, shape):
    """camera.MaxPool3x3s2 (one byte per output instead of an int64 index, gathering backward) against nn.MaxPool2d(3, 2, 1):
    outputs and input gradients EQUAL, also on maps full of ties (the zeros a ReLU leaves: the first maximum in row-major
    window order must win, as in torch), odd sizes and windows clipped by the border."""
    from u2mkd_amd import camera
    torch.manual_seed(sum(shape))
    pool, ref = camera.MaxPool3x3s2(), torch.nn.MaxPool2d(3, 2, 1)
    for relu in (False, True):
        x0 = torch.randn(*shape, device='cuda')
        if relu:
            x0 = torch.relu(x0 - 0.8)                         # mostly zeros: all-tie windows
        xa, xb = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        ya, yb = pool(xa), ref(xb)
        assert ya.shape == yb.shape and torch.equal(ya, yb)
        g = torch.randn_like(ya)
        ya.backward(g)
        yb.backward(g)
        assert torch.equal(xa.grad, xb.grad), float((xa.grad - xb.grad).abs().max())
