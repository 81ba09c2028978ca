"""The HIP SyncBatchNorm path at world size 2: local slab statistics -> all_gather of [2C+1] -> Chan merge ->
normalise (+ReLU); backward: local sums -> all_reduce of [2C] -> apply with the global count.

RCCL refuses two ranks on one device and the GPU box has one, so the two ranks are two fresh child processes on
cuda:0 joined by a GLOO group: the kernels are the product's, only the two tiny collectives travel through host
memory (functional._gather_rows / _sum_over_ranks).  UNEQUAL rows per rank.  Oracle: one process, nn.BatchNorm1d in
fp64 over the concatenated rows with the loss summed over the ranks -- y and dx of a rank are its slice, dgamma /
dbeta are the rank's own sums (DDP averages them afterwards), the running statistics use the global count."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
import torch, torch.distributed as dist
from u2mkd_amd.lidar.point_voxel import PointSyncBatchNorm1d
from u2mkd_amd.torchsparse.nn import functional as F
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
torch.cuda.set_device(0)
C, rows = 64, (1500, 377)
g = torch.Generator().manual_seed(7)
xs = [torch.randn(n, C, generator=g) * 2.0 + 1.5 for n in rows]
ws = [torch.randn(n, C, generator=g) for n in rows]
bn = PointSyncBatchNorm1d(C, momentum=0.1).cuda().train()
with torch.no_grad():
    bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
res = {}
for relu in (False, True):
    x = xs[rank].cuda().requires_grad_(True)
    bn.zero_grad(set_to_none=True)
    assert F._sync_group(bn) is not None and F._sync_group(bn)[1] == 2
    y = F.batch_norm(x, bn, relu)
    (y * ws[rank].cuda()).sum().backward()
    res['relu%d' % relu] = {'y': y.detach().cpu(), 'dx': x.grad.cpu(), 'dgamma': bn.weight.grad.cpu(), 'dbeta': bn.bias.grad.cpu()}
res['running_mean'], res['running_var'] = bn.running_mean.cpu(), bn.running_var.cpu()
res['tracked'] = int(bn.num_batches_tracked)
# the tail of a ResidualBlock: relu(bn(x) + r) inside the synchronising pieces
rs = [torch.randn(n, C, generator=g) for n in rows]
x = xs[rank].cuda().requires_grad_(True)
r = rs[rank].cuda().requires_grad_(True)
bn.zero_grad(set_to_none=True)
y = F.batch_norm(x, bn, True, r)
(y * ws[rank].cuda()).sum().backward()
res['res'] = {'y': y.detach().cpu(), 'dx': x.grad.cpu(), 'dres': r.grad.cpu(), 'dgamma': bn.weight.grad.cpu(), 'dbeta': bn.bias.grad.cpu()}
torch.save(res, out)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(child, outs, env, timeout, extra=()):
    """Two ranks of ``child`` on a free rendezvous port; a port that another process took between the probe and the bind
    (EADDRINUSE: seen once in ~50 suite runs) gets a second and a third try on a new one."""
    for attempt in range(3):
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, '-c', child, ROOT, str(r), str(len(outs)), str(port), outs[r], *extra], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(len(outs))]
        errs = []
        for p in procs:
            try:
                _, err = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            errs.append(err)
        if all(p.returncode == 0 for p in procs):
            return
        if attempt < 2 and any('EADDRINUSE' in e or 'address already in use' in e for e in errs):
            continue
        for p, err in zip(procs, errs):
            assert p.returncode == 0, err[-3000:]


@pytest.mark.timeout(600)
def test_hip_sync_batchnorm_two_ranks_equals_batchnorm_over_all_rows(hip, tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    outs = [str(tmp_path / f'r{r}.pt') for r in range(2)]
    _run_ranks(_CHILD, outs, env, 400)
    got = [torch.load(o) for o in outs]
    # ---- oracle: one process, all rows, fp64
    C, rows = 64, (1500, 377)
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(n, C, generator=g) * 2.0 + 1.5 for n in rows]
    ws = [torch.randn(n, C, generator=g) for n in rows]
    gamma = (torch.rand(C, generator=g) + 0.5).double()
    beta = (torch.randn(C, generator=g) * 0.2).double()
    for relu in (False, True):
        x = torch.cat(xs).double().requires_grad_(True)
        bn = torch.nn.BatchNorm1d(C, momentum=0.1).double().train()
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta)
        y = bn(x)
        pre = y
        if relu:
            y = torch.relu(y)
        (y * torch.cat(ws).double()).sum().backward()
        clear = (pre.detach().abs() > 1e-4) if relu else torch.ones_like(pre, dtype=torch.bool)
        xhat = (x.detach() - x.detach().mean(0)) / torch.sqrt(x.detach().var(0, unbiased=False) + bn.eps)
        dyp = torch.cat(ws).double() * ((pre.detach() > 0) if relu else 1.0)
        lo = 0
        for r, n in enumerate(rows):
            sl = slice(lo, lo + n)
            lo += n
            k = got[r]['relu%d' % relu]
            assert float((k['y'].double() - y.detach()[sl]).abs().max()) < 2e-5 * float(y.detach().abs().max()), (relu, r)
            assert float(((k['dx'].double() - x.grad[sl]) * clear[sl]).abs().max()) < 2e-4 * max(1.0, float(x.grad.abs().max())), (relu, r)
            want_db, want_dg = dyp[sl].sum(0), (dyp[sl] * xhat[sl]).sum(0)             # the rank's own sums
            assert float((k['dbeta'].double() - want_db).abs().max()) < 2e-4 * max(1.0, float(want_db.abs().max())), (relu, r)
            assert float((k['dgamma'].double() - want_dg).abs().max()) < 2e-4 * max(1.0, float(want_dg.abs().max())), (relu, r)
    # the residual form: relu(bn(x) + r), y / dx / dres per rank, the rank's own dgamma / dbeta
    rs = [torch.randn(n, C, generator=g) for n in rows]
    x = torch.cat(xs).double().requires_grad_(True)
    rr = torch.cat(rs).double().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C, momentum=0.1).double().train()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    pre = bn(x) + rr
    y = torch.relu(pre)
    (y * torch.cat(ws).double()).sum().backward()
    clear = pre.detach().abs() > 1e-4
    xhat = (x.detach() - x.detach().mean(0)) / torch.sqrt(x.detach().var(0, unbiased=False) + bn.eps)
    dyp = torch.cat(ws).double() * (pre.detach() > 0)
    lo = 0
    for r, n in enumerate(rows):
        sl = slice(lo, lo + n)
        lo += n
        k = got[r]['res']
        assert float((k['y'].double() - y.detach()[sl]).abs().max()) < 2e-5 * float(y.detach().abs().max()), r
        assert float(((k['dx'].double() - x.grad[sl]) * clear[sl]).abs().max()) < 2e-4 * max(1.0, float(x.grad.abs().max())), r
        assert float(((k['dres'].double() - rr.grad[sl]) * clear[sl]).abs().max()) < 2e-5 * max(1.0, float(rr.grad.abs().max())), r
        want_db, want_dg = dyp[sl].sum(0), (dyp[sl] * xhat[sl]).sum(0)
        assert float((k['dbeta'].double() - want_db).abs().max()) < 2e-4 * max(1.0, float(want_db.abs().max())), r
        assert float((k['dgamma'].double() - want_dg).abs().max()) < 2e-4 * max(1.0, float(want_dg.abs().max())), r
    # running statistics after the two passes (global mean, unbiased global variance), equal on both ranks
    ref = torch.nn.BatchNorm1d(C, momentum=0.1).double().train()
    for _ in range(2):
        ref(torch.cat(xs).double())
    for r in range(2):
        assert got[r]['tracked'] == 2
        assert float((got[r]['running_mean'].double() - ref.running_mean).abs().max()) < 1e-5
        assert float((got[r]['running_var'].double() - ref.running_var).abs().max()) < 1e-4
    assert torch.equal(got[0]['running_mean'], got[1]['running_mean']) and torch.equal(got[0]['running_var'], got[1]['running_var'])


_CHILD_DDP = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
import torch
from u2mkd_amd import distributed as D, lidar, torchsparse as ts
from u2mkd_amd.losses import MixLovaszCrossEntropy
from u2mkd_amd.synth import synth_batch
D.init_from_env('gloo')
torch.manual_seed(0)
model = lidar.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
model.dropout.p = 0.0
net = D.wrap_model(model, sync_bn=True)
assert isinstance(net, D.BucketedGradientAverage)
assert any(isinstance(m, lidar.SparseSyncBatchNorm) for m in net.modules())
b = synth_batch(1500 + 300 * rank, 1, seed=D.scene_seed(100))          # unequal scenes
feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
loss = MixLovaszCrossEntropy(ignore_index=0)(net({'lidar': ts.SparseTensor(feats, coords)})['x_vox'], labels)
loss.backward()
torch.cuda.synchronize()
torch.save({'loss': float(loss), 'grads': {n: p.grad.cpu() for n, p in net.module.named_parameters()}}, out)
D.shutdown()
'''


@pytest.mark.timeout(900)
def test_two_rank_ddp_step_equals_one_process_on_both_scenes(hip, tmp_path):
    """DDP (gradient buckets all-reduced during the backward, the weight-gradient side stream underneath) +
    SparseSyncBatchNorm / PointSyncBatchNorm1d on the HIP path, two ranks with their own scenes: the averaged gradients
    equal those of ONE process holding both scenes in one batch with the loss (L_0 + L_1) / 2 -- BatchNorm over all
    voxels of the batch is what SyncBatchNorm computes over the ranks."""
    import numpy as np
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    from u2mkd_amd.synth import synth_batch
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'U2MKD_FORCE_DDP', 'U2MKD_FORCE_SYNC_BN')}
    outs = [str(tmp_path / f'd{r}.pt') for r in range(2)]
    _run_ranks(_CHILD_DDP, outs, env, 600)
    got = [torch.load(o) for o in outs]
    for n in got[0]['grads']:
        assert torch.equal(got[0]['grads'][n], got[1]['grads'][n]), n               # all-reduced: the same on both ranks
    # ---- one process, both scenes in one batch
    torch.manual_seed(0)
    model = lidar.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
    model.dropout.p = 0.0
    scenes = [synth_batch(1500 + 300 * r, 1, seed=100 + r) for r in range(2)]
    for r, s in enumerate(scenes):
        s['coords'][:, 3] = r
    feats = torch.from_numpy(np.concatenate([s['feats'] for s in scenes])).cuda()
    coords = torch.from_numpy(np.concatenate([s['coords'] for s in scenes])).cuda()
    labels = [torch.from_numpy(s['labels']).cuda() for s in scenes]
    out = model({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
    n0 = scenes[0]['feats'].shape[0]
    crit = MixLovaszCrossEntropy(ignore_index=0)
    l0, l1 = crit(out[:n0], labels[0]), crit(out[n0:], labels[1])
    ((l0 + l1) / 2).backward()
    l0v, l1v = float(l0.detach()), float(l1.detach())
    assert abs(got[0]['loss'] - l0v) < 1e-4 * abs(l0v) and abs(got[1]['loss'] - l1v) < 1e-4 * abs(l1v)
    errs = []
    gmax = max(float(p.grad.norm()) for p in model.parameters())
    for n, p in model.named_parameters():
        g = got[0]['grads'][n].cuda()
        # (the bias of a Linear in front of a BatchNorm has a gradient of exactly zero up to rounding: floor the scale)
        errs.append(float((g - p.grad).norm() / max(float(p.grad.norm()), 1e-4 * gmax)))
    errs = np.array(errs)
    print('DDP-2 vs one process: L2-relative gradient distance median %.2e max %.2e' % (np.median(errs), errs.max()))
    # fp32 summation order differs (per-rank slabs vs one batch) and ReLU inputs within rounding of zero may flip
    assert np.median(errs) < 1e-3 and errs.max() < 3e-2, (np.median(errs), errs.max())


_CHILD_2D = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
import torch, torch.distributed as dist
from u2mkd_amd.camera import BatchNorm2d, SyncBatchNorm2d, bn_act
from u2mkd_amd.lidar.point_voxel import SparseSyncBatchNorm
from u2mkd_amd.torchsparse.nn import functional as F
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
torch.cuda.set_device(0)
C, shapes = 24, ((3, 20, 36), (2, 20, 36))            # UNEQUAL images per rank
g = torch.Generator().manual_seed(11)
xs = [torch.randn(b, C, h, w, generator=g) * 1.5 + 0.7 for b, h, w in shapes]
rs = [torch.randn(b, C, h, w, generator=g) for b, h, w in shapes]
ws = [torch.randn(b, C, h, w, generator=g) for b, h, w in shapes]
bn = BatchNorm2d(C, momentum=0.1)
with torch.no_grad():
    bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
bn = SparseSyncBatchNorm.convert_sync_batchnorm(bn).cuda().train()
assert isinstance(bn, SyncBatchNorm2d) and F._sync_group(bn)[1] == 2
res = {}
for mode in ('plain', 'relu', 'res'):
    x = xs[rank].cuda().requires_grad_(True)
    r = rs[rank].cuda().requires_grad_(True) if mode == 'res' else None
    bn.zero_grad(set_to_none=True)
    y = bn_act(bn, x, relu=mode != 'plain', residual=r)
    (y * ws[rank].cuda()).sum().backward()
    res[mode] = {'y': y.detach().cpu(), 'dx': x.grad.cpu(), 'dgamma': bn.weight.grad.cpu(), 'dbeta': bn.bias.grad.cpu(),
                 'dres': r.grad.cpu() if r is not None else None}
res['running_mean'], res['running_var'], res['tracked'] = bn.running_mean.cpu(), bn.running_var.cpu(), int(bn.num_batches_tracked)
torch.save(res, out)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_hip_sync_batchnorm2d_two_ranks_equals_batchnorm2d_over_all_images(hip, tmp_path):
    """The camera branch's BatchNorm2d under DDP (SyncBatchNorm2d on the csrc/bn2d.hip pieces, ReLU / residual fused):
    two processes with 3 and 2 images against ONE fp64 nn.BatchNorm2d over all 5 -- y, dx, the residual gradient, the
    per-rank dgamma / dbeta, the running statistics with the global count."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    outs = [str(tmp_path / f'q{r}.pt') for r in range(2)]
    _run_ranks(_CHILD_2D, outs, env, 400)
    got = [torch.load(o) for o in outs]
    C, shapes = 24, ((3, 20, 36), (2, 20, 36))
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(b, C, h, w, generator=g) * 1.5 + 0.7 for b, h, w in shapes]
    rs = [torch.randn(b, C, h, w, generator=g) for b, h, w in shapes]
    ws = [torch.randn(b, C, h, w, generator=g) for b, h, w in shapes]
    gamma = (torch.rand(C, generator=g) + 0.5).double()
    beta = (torch.randn(C, generator=g) * 0.2).double()
    nb = [s[0] for s in shapes]
    for mode in ('plain', 'relu', 'res'):
        x = torch.cat(xs).double().requires_grad_(True)
        r = torch.cat(rs).double().requires_grad_(True) if mode == 'res' else None
        bn = torch.nn.BatchNorm2d(C, momentum=0.1).double().train()
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta)
        pre = bn(x) + (r if r is not None else 0.0)
        y = torch.relu(pre) if mode != 'plain' else pre
        w_all = torch.cat(ws).double()
        (y * w_all).sum().backward()
        clear = (pre.detach().abs() > 1e-4) if mode != 'plain' else torch.ones_like(pre, dtype=torch.bool)
        xd = x.detach()
        xhat = (xd - xd.mean((0, 2, 3), keepdim=True)) / torch.sqrt(xd.var((0, 2, 3), unbiased=False, keepdim=True) + bn.eps)
        dyp = w_all * ((pre.detach() > 0) if mode != 'plain' else 1.0)
        lo = 0
        for rk, n in enumerate(nb):
            sl = slice(lo, lo + n)
            lo += n
            k = got[rk][mode]
            assert float((k['y'].double() - y.detach()[sl]).abs().max()) < 2e-5 * float(y.detach().abs().max()), (mode, rk)
            assert float(((k['dx'].double() - x.grad[sl]) * clear[sl]).abs().max()) < 2e-4 * max(1.0, float(x.grad.abs().max())), (mode, rk)
            if r is not None:
                assert float(((k['dres'].double() - r.grad[sl]) * clear[sl]).abs().max()) < 2e-5 * max(1.0, float(r.grad.abs().max()))
            want_db, want_dg = dyp[sl].sum((0, 2, 3)), (dyp[sl] * xhat[sl]).sum((0, 2, 3))
            assert float((k['dbeta'].double() - want_db).abs().max()) < 2e-4 * max(1.0, float(want_db.abs().max())), (mode, rk)
            assert float((k['dgamma'].double() - want_dg).abs().max()) < 2e-4 * max(1.0, float(want_dg.abs().max())), (mode, rk)
    ref = torch.nn.BatchNorm2d(C, momentum=0.1).double().train()
    for _ in range(3):
        ref(torch.cat(xs).double())
    for rk in range(2):
        assert got[rk]['tracked'] == 3
        assert float((got[rk]['running_mean'].double() - ref.running_mean).abs().max()) < 1e-5
        assert float((got[rk]['running_var'].double() - ref.running_var).abs().max()) < 1e-4


_CHILD_HEAD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
import torch, torch.distributed as dist
from u2mkd_amd import camera
from u2mkd_amd.lidar.point_voxel import SparseSyncBatchNorm
from u2mkd_amd.pixel_head import sampled_head_applies, sampled_pixel_logits
from u2mkd_amd.synth import synth_kd_batch
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
torch.cuda.set_device(0)
hw, low, c, classes, ncam = (64, 112), (32, 56), 32, 17, 6
b = synth_kd_batch(900, 2, seed=9, image_hw=hw)['student']
pc = [torch.from_numpy(b['pixel_coordinates'][rank]).cuda()]          # rank r holds sample r
ms = [torch.from_numpy(b['masks'][rank]).cuda()]
g = torch.Generator().manual_seed(21)
head = camera.BNReluConv(c, classes, k=1)
with torch.no_grad():
    head.norm.weight.copy_(torch.rand(c, generator=g) + 0.5); head.norm.bias.copy_(torch.randn(c, generator=g) * 0.3)
    head.conv.weight.copy_(torch.randn(classes, c, 1, 1, generator=g) * 0.2)
head = SparseSyncBatchNorm.convert_sync_batchnorm(head).cuda().train()
x_all = torch.randn(2 * ncam, c, *low, generator=g) * 1.7 + 3.0
gy_all = [torch.randn(m.shape[1], classes, generator=g) for m in b['masks']]
x = x_all[rank * ncam:(rank + 1) * ncam].cuda().requires_grad_(True)
assert isinstance(head.norm, camera.SyncBatchNorm2d) and sampled_head_applies(x, head)
y = sampled_pixel_logits(x, head, pc, ms, hw, 1, ncam)
y.backward(gy_all[rank].cuda())
torch.save({'y': y.detach().cpu(), 'dx': x.grad.cpu(), 'dgamma': head.norm.weight.grad.cpu(), 'dbeta': head.norm.bias.grad.cpu(),
            'dw': head.conv.weight.grad.cpu(), 'running_mean': head.norm.running_mean.cpu(), 'running_var': head.norm.running_var.cpu()}, out)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_sampled_pixel_head_two_ranks_equals_the_dense_head_over_both_samples(hip, tmp_path):
    """The sampled pixel head (u2mkd_amd/pixel_head.py) under SyncBatchNorm: two processes with one sample (6 cameras)
    each against ONE dense fp64 evaluation over both samples -- logits and map gradient per rank, the BatchNorm affine
    and classifier gradients summed over the ranks, running statistics over all 12 up-sampled maps."""
    import torch.nn.functional as F
    from test_gpu_pixel_head import _dense_fp64
    from u2mkd_amd import camera
    from u2mkd_amd.synth import synth_kd_batch
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    outs = [str(tmp_path / f'h{r}.pt') for r in range(2)]
    _run_ranks(_CHILD_HEAD, outs, env, 400)
    got = [torch.load(o) for o in outs]
    hw, low, c, classes, ncam = (64, 112), (32, 56), 32, 17, 6
    b = synth_kd_batch(900, 2, seed=9, image_hw=hw)['student']
    pc = [torch.from_numpy(p).cuda() for p in b['pixel_coordinates']]
    ms = [torch.from_numpy(m).cuda() for m in b['masks']]
    g = torch.Generator().manual_seed(21)
    head = camera.BNReluConv(c, classes, k=1)
    with torch.no_grad():
        head.norm.weight.copy_(torch.rand(c, generator=g) + 0.5)
        head.norm.bias.copy_(torch.randn(c, generator=g) * 0.3)
        head.conv.weight.copy_(torch.randn(classes, c, 1, 1, generator=g) * 0.2)
    head = head.cuda().train()
    x_all = torch.randn(2 * ncam, c, *low, generator=g) * 1.7 + 3.0
    gy_all = [torch.randn(m.shape[1], classes, generator=g) for m in b['masks']]
    xd = x_all.cuda().double().requires_grad_(True)
    want, u = _dense_fp64(xd, head, pc, ms, hw, 2, ncam)
    want.backward(torch.cat(gy_all).cuda().double())
    n0 = ms[0].shape[1]
    scale = float(want.abs().max())
    for r, sl, isl in ((0, slice(0, n0), slice(0, ncam)), (1, slice(n0, None), slice(ncam, None))):
        assert float((got[r]['y'].double() - want.detach()[sl].cpu()).abs().max()) < 2e-5 * scale, r
        assert float((got[r]['dx'].double() - xd.grad[isl].cpu()).abs().max()) < 5e-5 * float(xd.grad.abs().max()), r
    for key, ref in (('dgamma', head.norm.weight.grad), ('dbeta', head.norm.bias.grad), ('dw', head.conv.weight.grad)):
        tot = got[0][key].double() + got[1][key].double()
        assert float((tot - ref.cpu().double()).abs().max()) < 5e-5 * float(ref.abs().max()), key
    m = 0.1
    mean, var = u.detach().mean((0, 2, 3)).cpu(), u.detach().var((0, 2, 3), unbiased=True).cpu()
    for r in range(2):
        assert float((got[r]['running_mean'].double() - m * mean).abs().max()) < 1e-5
        assert float((got[r]['running_var'].double() - ((1 - m) + m * var)).abs().max()) < 1e-5 * float(var.max())
