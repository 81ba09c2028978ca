"""GPU parity of the whole LiDAR network (row a9): u2mkd_amd.lidar.SPVCNN on
HIP kernels vs the CPU oracle restatement, identical state dict and cloud.
North-star gate: per-point logits within 1e-3 abs (fp32); kernel grads 1e-3 rel."""
import numpy as np
import pytest
import torch

from oracle import spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def _run_three(n_vox, batch, cr, seed, monkeypatch):
    """The same network, state dict and cloud three times: CPU oracle in fp64 (the arbiter), CPU oracle in fp32 (the
    reference arithmetic: logits / loss are compared with it) and the HIP model; every ReLU recorded on the arbiter
    and on the HIP run (tests/grad_arbiter.py)."""
    import grad_arbiter as GA
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    b = synth_batch(n_vox, batch, seed)
    feats, coords, labels = (torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))
    kw = dict(cr=cr, in_channel=4, num_classes=17, pres=0.05, vres=0.05)

    crit = MixLovaszCrossEntropy(ignore_index=0)

    def run_cpu(dtype, record, masks=None):
        m = O.fill_state_by_name(O.SPVCNN(**kw)).train().to(dtype)
        m.dropout.p = 0.0            # random masks differ across implementations (SURVEY 8d "Weights")
        if masks is not None:        # the arbiter evaluated on the HIP run's side of every ReLU (grad_arbiter.force_relu_masks)
            GA.force_relu_masks(m, masks)
        rec, remove = GA.record_oracle_relus(m) if record else ({}, lambda: None)
        out = m({'lidar': ots.SparseTensor(feats.to(dtype), coords)})['x_vox']
        remove()
        # (the oracle's Lovasz restatement computes in .float() like core/criterions.py; the fp64 arbiter takes the
        # dtype-generic form of the same loss)
        loss = O.mix_lovasz_cross_entropy(out, labels) if dtype == torch.float32 else crit(out, labels)
        loss.backward()
        return m, out.detach(), loss.detach(), rec
    m64, _, _, rec64 = run_cpu(torch.float64, True)
    m32, out32, loss32, _ = run_cpu(torch.float32, False)
    mg = lidar.SPVCNN(**kw)
    mg.load_state_dict(m32.state_dict())     # same keys by construction
    mg.cuda().train()
    mg.dropout.p = 0.0
    rec_hip = GA.record_hip_relus(mg, monkeypatch)
    out = mg({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox']
    loss = crit(out, labels.cuda())
    loss.backward()
    return m64, m32, out32, loss32, mg, out, loss, rec64, rec_hip, lambda masks: run_cpu(torch.float64, False, masks)[0]


@pytest.mark.parametrize('n_vox,batch,cr', [(3000, 2, 0.5), (30000, 1, 0.5), (6000, 1, 1.0)])
def test_spvcnn_logits_and_grads(hip, monkeypatch, n_vox, batch, cr):
    import grad_arbiter as GA
    m64, m32, out_ref, loss_ref, model, out, loss, rec64, rec_hip, rerun64 = _run_three(n_vox, batch, cr, 11, monkeypatch)
    err = float((out.detach().cpu() - out_ref).abs().max())
    assert err < 1e-3, f'logit max abs err {err}'
    assert abs(float(loss) - float(loss_ref)) < 1e-3
    # Gradient gate (1e-3 relative per conv kernel), against the fp64 arbiter.  The network is only
    # piecewise smooth: a ReLU input within fp32 rounding of 0 flips between two fp32 evaluation orders, and one
    # flipped element can carry a visible share of a layer's gradient -- such a scene passes only with the flipped
    # elements found and named (grad_arbiter.assert_grads_within_fp64_gate); smooth operators are held to 1e-4 in
    # test_gpu_torchsparse_ops.py.
    GA.assert_grads_within_fp64_gate('scene 11 (%d voxels x %d, cr %.1f)' % (n_vox, batch, cr), model, m64, m32, rec64, rec_hip, rerun64)
    ref_grads = dict(m32.named_parameters())
    for name, p in model.named_parameters():
        if float(ref_grads[name].grad.abs().max()) < 1e-7:
            # structurally zero gradient (a Linear bias feeding a train-mode BatchNorm)
            assert float(p.grad.abs().max()) < 1e-6, name


def test_kernel_grads_are_as_close_to_fp64_as_the_fp32_reference_is(hip, monkeypatch):
    """The same gate over three more clouds.  What a miss looked like when it happened (scene 11, when nn.Linear moved
    from f32 MFMA to bf16x3 arithmetic): every activation and every upstream gradient of the two runs agreed to 5e-7,
    ONE mask element of vox_ups.2.1.0 differed and carried 0.5 % of that layer's gradient norm -- all 40 kernels below
    it moved by 2e-3 while both nn.Linear variants were within 1.5e-7 of fp64.  The arbiter now finds and names such
    elements itself; a scene misses the strict bound only with them, and never by more than a flip's size."""
    import grad_arbiter as GA
    strict = []
    for seed in (12, 13, 14):
        m64, m32, _, _, mg, _, _, rec64, rec_hip, rerun64 = _run_three(3000, 2, 0.5, seed, monkeypatch)
        strict.append(GA.assert_grads_within_fp64_gate('scene %d' % seed, mg, m64, m32, rec64, rec_hip, rerun64))
        monkeypatch.undo()
    print('GRAD-FP64 strict on', strict)
    # a scene that is not strict has passed the causal form (fp64 re-evaluated with the HIP masks, 1e-3); flips are rare
    # events, so most scenes must be strict outright
    assert sum(strict) >= 2, strict


def test_ddp_syncbn_path_on_gpu_single_rank(hip):
    """The N>1 code path (NCCL process group, SyncBatchNorm conversion, DDP wrap, bucketed all-reduce
    hooks) executed with world_size 1 on the GPU: the HIP autograd ops must work under DDP."""
    import os
    import socket
    import torch.distributed as dist
    from u2mkd_amd import distributed as D, lidar, train as T
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        b = synth_batch(2500, 2, 5)
        feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
        model = lidar.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
        sync = lidar.SparseSyncBatchNorm.convert_sync_batchnorm(model)
        assert any(isinstance(m, lidar.SparseSyncBatchNorm) for m in sync.modules())
        net = torch.nn.parallel.DistributedDataParallel(sync, device_ids=[0], gradient_as_bucket_view=True)
        opt = T.make_optimizer(net.parameters())
        from u2mkd_amd import torchsparse as ts
        from u2mkd_amd.losses import MixLovaszCrossEntropy
        crit = MixLovaszCrossEntropy(ignore_index=0)
        os.environ['U2MKD_FORCE_SYNC_BN'] = '1'     # take the synchronising BatchNorm path although world_size == 1
        losses = []
        for _ in range(3):
            out = net({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
            loss = crit(out, labels)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert all(np.isfinite(losses)) and losses[-1] < losses[0] + 1.0
        assert D.max_over_ranks(1.5) == 1.5
        _check_hip_sync_bn_pieces(lidar)
    finally:
        os.environ.pop('U2MKD_FORCE_SYNC_BN', None)
        dist.destroy_process_group()


def _check_hip_sync_bn_pieces(lidar):
    """The SyncBatchNorm path (local stats | all_gather | merge | apply; local sums | all_reduce | apply)
    forced on at world_size 1: must reproduce the fused single-GPU BatchNorm(+ReLU) and its running stats."""
    import os
    from u2mkd_amd.lidar.blocks import PointBatchNorm1d
    from u2mkd_amd.lidar.point_voxel import PointSyncBatchNorm1d
    from u2mkd_amd.torchsparse.nn import functional as spf
    torch.manual_seed(3)
    for n, c, relu in ((5000, 32, True), (777, 96, False), (2, 64, True)):
        ref = PointBatchNorm1d(c).cuda().train()
        with torch.no_grad():
            ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-0.5, 0.5)
        syn = PointSyncBatchNorm1d(c).cuda().train()
        syn.load_state_dict(ref.state_dict())
        x = torch.randn(n, c, device='cuda') * 2 + 1
        g = torch.randn(n, c, device='cuda')
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya = spf.batch_norm(xa, ref, relu)
        os.environ['U2MKD_FORCE_SYNC_BN'] = '1'
        yb = spf.batch_norm(xb, syn, relu)
        os.environ.pop('U2MKD_FORCE_SYNC_BN')
        ya.backward(g); yb.backward(g)
        assert float((ya - yb).abs().max()) < 1e-5
        assert float((xa.grad - xb.grad).abs().max()) < 1e-5
        assert float((ref.weight.grad - syn.weight.grad).abs().max()) < 1e-3 * max(1.0, float(ref.weight.grad.abs().max()))
        assert float((ref.bias.grad - syn.bias.grad).abs().max()) < 1e-3 * max(1.0, float(ref.bias.grad.abs().max()))
        assert float((ref.running_mean - syn.running_mean).abs().max()) < 1e-6
        assert float((ref.running_var - syn.running_var).abs().max()) < 1e-5
        assert int(syn.num_batches_tracked) == 1


@pytest.mark.parametrize('n0,n1,c,relu', [(3000, 1700, 32, True), (777, 5, 96, False), (64, 4000, 256, True)])
def test_sync_bn_pieces_emulate_two_ranks_on_one_gpu(hip, n0, n1, c, relu):
    """SyncBatchNorm at world_size 2, emulated on ONE GPU: a batch is cut into two UNEQUAL per-rank parts, every
    part goes through the per-rank pieces (u2mkd_bn_local_stats -> [all_gather = stacking the two [2C+1] rows] ->
    u2mkd_bn_merge_stats(world = 2) -> u2mkd_bn_apply;  u2mkd_bn_backward_local -> [all_reduce = the sum of the two
    [2C] rows] -> u2mkd_bn_backward_apply) and the result must equal torch's BatchNorm1d over the whole batch in
    fp64: y, dx, the per-rank dgamma / dbeta (DDP averages them afterwards: their SUM is the full-batch gradient),
    the running statistics with the GLOBAL count (torch.nn.SyncBatchNorm semantics, core/models/utils.py:138-220)."""
    from u2mkd_amd import _lib as L
    torch.manual_seed(n0 + c)
    eps, mom = 1e-5, 0.1
    x = torch.randn(n0 + n1, c, device='cuda') * 1.7 + 0.6
    dy = torch.randn(n0 + n1, c, device='cuda')
    gamma = torch.rand(c, device='cuda') + 0.5
    beta = torch.rand(c, device='cuda') - 0.5
    # reference: fp64 BatchNorm1d over the concatenated batch
    ref = torch.nn.BatchNorm1d(c, eps=eps, momentum=mom).double().cuda().train()
    with torch.no_grad():
        ref.weight.copy_(gamma.double()); ref.bias.copy_(beta.double())
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.double())
    # the two "ranks"
    parts = [(x[:n0].contiguous(), dy[:n0].contiguous()), (x[n0:].contiguous(), dy[n0:].contiguous())]
    lib, st = L.load(), L.stream()
    stats = torch.empty(2, 2 * c + 1, device='cuda')
    for r, (xp, _) in enumerate(parts):
        n = xp.shape[0]
        partial = torch.empty(max(lib.u2mkd_bn_num_slabs(n), 1) * 2 * c, device='cuda')
        L.call('u2mkd_bn_local_stats', L.ptr(xp), n, c, L.ptr(partial), L.ptr(stats[r]), st)
    assert stats[:, 2 * c].tolist() == [float(n0), float(n1)]
    run_mean, run_var = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    mean, invstd, total = torch.empty(c, device='cuda'), torch.empty(c, device='cuda'), torch.empty(1, device='cuda')
    L.call('u2mkd_bn_merge_stats', L.ptr(stats), 2, c, eps, mom, L.ptr(run_mean), L.ptr(run_var), L.ptr(mean), L.ptr(invstd),
           L.ptr(total), st)
    assert float(total) == n0 + n1
    assert float((run_mean.double() - ref.running_mean).abs().max()) < 1e-6
    assert float((run_var.double() - ref.running_var).abs().max()) < 1e-5
    ys, sums = [], torch.empty(2, 2 * c, device='cuda')
    for r, (xp, dyp) in enumerate(parts):
        n = xp.shape[0]
        y = torch.empty_like(xp)
        L.call('u2mkd_bn_apply', L.ptr(xp), n, c, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), int(relu), L.ptr(y), st)
        ys.append(y)
        partial = torch.empty(max(lib.u2mkd_bn_num_slabs(n), 1) * 2 * c, device='cuda')
        L.call('u2mkd_bn_backward_local', L.ptr(dyp), L.ptr(xp), n, c, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta),
               int(relu), L.ptr(partial), L.ptr(sums[r]), st)
    assert float((torch.cat(ys).double() - yr).abs().max()) < 1e-5
    reduced = sums.sum(0).contiguous()                      # the all_reduce
    dxs = []
    for xp, dyp in parts:
        n = xp.shape[0]
        dx = torch.empty_like(xp)
        L.call('u2mkd_bn_backward_apply', L.ptr(dyp), L.ptr(xp), n, c, L.ptr(total), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
               L.ptr(beta), int(relu), L.ptr(reduced), L.ptr(dx), st)
        dxs.append(dx)
    scale = max(1.0, float(xr.grad.abs().max()))
    assert float((torch.cat(dxs).double() - xr.grad).abs().max()) < 1e-4 * scale
    # sums rows = [dbeta, dgamma] per rank; the full-batch parameter gradients are their sums
    gs = max(1.0, float(ref.weight.grad.abs().max()))
    assert float((reduced[:c].double() - ref.bias.grad).abs().max()) < 1e-4 * gs
    assert float((reduced[c:].double() - ref.weight.grad).abs().max()) < 1e-4 * gs


def test_teacher_only_step_masks_the_loss_to_key_frame_voxels(hip):
    """Row f4 -- the stage-1 trainer (core/spformer_trainer.py:58-94) on a multi-sweep scene: the network sees the
    voxels of all sweeps, the loss only the key-frame ones (`outputs['x_vox'][keyframe_mask]`).  The product step
    must give the loss and the gradients of the manual computation on the same weights, and voxels of the other
    sweeps must not contribute to the classifier's bias gradient."""
    from u2mkd_amd import lidar, torchsparse as ts, train as T
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    b = synth_batch(6000, 1, seed=9, sweeps=3)
    kf = torch.from_numpy(b['keyframe']).cuda()
    assert 0 < int(kf.sum()) < kf.numel()                   # genuinely multi-sweep
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    torch.manual_seed(0)
    kw = dict(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    model = lidar.SPVCNN(**kw).cuda().train()
    model.dropout.p = 0.0
    twin = lidar.SPVCNN(**kw).cuda().train()
    twin.dropout.p = 0.0
    twin.load_state_dict(model.state_dict())
    # manual: forward, mask, loss, backward
    out = twin({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
    want = MixLovaszCrossEntropy(ignore_index=0)(out[kf], labels[kf])
    want.backward()
    runner = T.LidarStep(model, num_epochs=1, batch_size=1)
    w0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    got = runner(feats, coords, labels, keyframe_mask=kf)
    assert abs(float(got) - float(want)) < 1e-5
    # the step was plain SGD-nesterov from zero momentum: p1 = p0 - lr * (1 + momentum) * (g + wd * p0)
    lr, mom, wd = 0.24, 0.9, 1.0e-4
    for n, p in model.named_parameters():
        g = dict(twin.named_parameters())[n].grad
        step = (w0[n] - p.detach()) / (lr * (1 + mom)) - wd * w0[n]
        assert float((step - g).abs().max()) <= 2e-4 * max(1.0, float(g.abs().max())), n
    # and the loss without the mask is a different number (the mask is not a no-op on this scene)
    full = MixLovaszCrossEntropy(ignore_index=0)(out.detach(), labels)
    assert abs(float(full) - float(want)) > 1e-4


def test_lidar_step_prefetch_loop_ending_without_prefetch_matches_the_plain_loop(hip, monkeypatch):
    """train.LidarStep(prefetch=next) queues the next batch's geometry on a side stream; the LAST call of an epoch has
    prefetch=None and must still order its forward behind that stream (the wait sat inside the staged branch).
    Losses of the prefetching loop == the plain loop's, step for step (SPVCNN on the library's kernels is bit-reproducible),
    in the staged and in the one-piece mode."""
    import torch
    from u2mkd_amd import lidar, torchsparse as ts, train as T
    from u2mkd_amd.synth import synth_batch
    kw = dict(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    bs = [synth_batch(9000 + 1000 * i, 1, seed=60 + i) for i in range(3)]
    bs = [tuple(torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels')) for b in bs]

    def loop(prefetch, staged='1'):
        monkeypatch.setenv('U2MKD_STAGED_GEOMETRY', staged)
        torch.manual_seed(0)
        model = lidar.SPVCNN(**kw).cuda().train()
        model.dropout.p = 0.0
        run = T.LidarStep(model, num_epochs=1, batch_size=1)
        cur = [t.clone() for t in bs[0]]
        out = []
        for i in range(3):
            nxt = [t.clone() for t in bs[i + 1]] if i + 1 < 3 else None
            out.append(float(run(*cur, prefetch=(nxt[0], nxt[1]) if (prefetch and nxt) else None)))
            cur = nxt
        return out
    want = loop(False)
    for staged in ('1', '0'):
        got = loop(True, staged)
        assert got == want, (staged, got, want)


def test_eval_batchnorm_folded_into_the_convolution_equals_the_two_pass_form(hip):
    """Inference (frozen teacher / evaluation): spnn.Conv3d -> eval-mode BatchNorm (-> ReLU | + residual -> ReLU) runs as one
    convolution whose store applies scale / shift / residual / ReLU (functional.conv_eval_affine, build_blocks.py:25-31,59-71);
    against the two-pass formulation (U2MKD_FOLD_EVAL_BN=0's path) on both conv schedules, strided and transposed convs."""
    import torch
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.lidar import blocks
    from u2mkd_amd.synth import synth_batch
    b = synth_batch(12000, 2, seed=17)
    feats, coords = torch.from_numpy(b['feats']).cuda(), torch.from_numpy(b['coords']).cuda()
    torch.manual_seed(5)
    model = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=2.0)).cuda().eval()
    for m in model.modules():          # running statistics away from their initial (0, 1)
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.7, 1.3)
            m.bias.data.normal_(0, 0.1)
    calls = []
    orig = blocks.spf.conv_eval_affine

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(out is not None)
        return out
    blocks.spf.conv_eval_affine = spy
    try:
        with torch.no_grad():
            folded = model({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
            n_folded = sum(calls)
            blocks._FOLD_EVAL_BN = False
            plain = model({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
    finally:
        blocks._FOLD_EVAL_BN = True
        blocks.spf.conv_eval_affine = orig
    assert n_folded >= 30, (n_folded, len(calls))          # most of the 49 conv + BatchNorm pairs have a folded form
    err = float((folded - plain).abs().max())
    print('FOLDED-BN %d of %d pairs folded, max |logit difference| %.2e (range %.1f)' % (n_folded, len(calls), err, float(plain.abs().max())))
    assert err < 2e-5 * max(1.0, float(plain.abs().max()))
    # training mode / gradients enabled: never folded
    calls.clear()
    blocks.spf.conv_eval_affine = spy
    try:
        model.train()
        model({'lidar': ts.SparseTensor(feats, coords)})
    finally:
        blocks.spf.conv_eval_affine = orig
    assert not calls
