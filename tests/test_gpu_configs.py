"""The two BASELINE.json configurations no other GPU test steps through as a whole:
configs[3] -- the `_B` student (cr 2.0) + cr_t 2.0 teacher, batch 2 per GPU, DDP + SyncBatchNorm;
configs[4] -- multi-sweep teacher input (keyframe masks) under bf16 autocast.
Small scenes and images: these check the wiring of the step (shapes, re-index, streams, collectives, autocast),
parity of the operators is held elsewhere (test_kd_path.py goldens at cr 2.0, test_gpu_torchsparse_ops.py)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _runner(cr, cr_t, amp=False):
    from u2mkd_amd import kd as KD, lidar, train as T
    torch.manual_seed(0)
    sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
    model = KD.TSDFull(cr=cr, cr_t=cr_t, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
    run = T.KDStep(model, num_epochs=50, batch_size=2, amp=amp)
    run.train_mode()
    for m in model.modules():                      # deterministic comparison runs: no dropout / drop-path masks
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0
    return run


def test_configs3_cr2_student_batch2_under_ddp_and_syncbn(hip):
    """_B.yaml (cr 2.0 / cr_t 2.0), two scenes per GPU, the N>1 code path forced on one GPU (NCCL group of one rank,
    DDP wrap, SyncBatchNorm conversion of the student): the first step's loss equals the plain single-process
    step's, every student parameter receives a gradient, the teacher none, the loss stays finite over three steps."""
    import torch.distributed as dist
    from u2mkd_amd import train as T
    from u2mkd_amd.synth import synth_kd_batch
    d = T.kd_batch_to_device(synth_kd_batch(2500, 2, seed=21, image_hw=(64, 112)))
    plain = _runner(2.0, 2.0)
    want = float(plain(d))
    del plain
    torch.cuda.empty_cache()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    os.environ['U2MKD_FORCE_DDP'] = '1'
    os.environ['U2MKD_FORCE_SYNC_BN'] = '1'
    try:
        run = _runner(2.0, 2.0)
        from u2mkd_amd import distributed as D
        assert isinstance(run.net, D.BucketedGradientAverage)
        assert any(type(m).__name__.endswith('SyncBatchNorm') or 'Sync' in type(m).__name__ for m in run.model.model_s.modules())
        losses = [float(run(d)) for _ in range(3)]
        assert abs(losses[0] - want) < 2e-3 * abs(want), (losses[0], want)
        # (lr 0.24 on a tiny scene with random labels: the loss is not monotone; it has to stay finite and bounded)
        assert all(np.isfinite(losses)) and max(losses) < 10 * losses[0]
        for n, p in run.model.model_s.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        assert all(p.grad is None for p in run.model.model_t.parameters())
    finally:
        os.environ.pop('U2MKD_FORCE_DDP', None)
        os.environ.pop('U2MKD_FORCE_SYNC_BN', None)
        dist.destroy_process_group()


def test_configs4_multisweep_teacher_under_bf16_autocast(hip):
    """Three aggregated sweeps for the teacher (keyframe_mask_full over all points, the student on the key frame),
    the step under bf16 autocast as `amp_enabled` does in the reference trainer: the first loss is within bf16
    noise of the fp32 step's, it stays finite, master weights and gradients stay fp32."""
    from u2mkd_amd import train as T
    from u2mkd_amd.synth import synth_kd_batch
    nb = synth_kd_batch(7000, 1, seed=33, image_hw=(64, 112), sweeps=3)
    assert nb['teacher']['num_pts'][0] > nb['teacher']['keyframe_mask_full'].sum() > 0
    d = T.kd_batch_to_device(nb)
    assert d['keyframe_mask_full'] is not None
    f32 = _runner(1.0, 2.0)
    want = float(f32(d))
    del f32
    torch.cuda.empty_cache()
    run = _runner(1.0, 2.0, amp='bf16')
    losses = [float(run(d)) for _ in range(3)]
    assert abs(losses[0] - want) < 0.05 * abs(want), (losses[0], want)
    assert all(np.isfinite(losses)) and max(losses) < 10 * losses[0]
    for n, p in run.model.model_s.named_parameters():
        assert p.dtype == torch.float32 and p.grad is not None and p.grad.dtype == torch.float32, n


def test_prefetched_geometry_makes_the_forward_sync_free_and_changes_nothing(hip):
    """train.KDStep(prefetch=next batch): the voxel sets and kernel maps of batch k+1 are built between the forward and
    the backward of step k (kd.TSDFull.prepare).  (1) same losses as the plain loop, step for step; (2) a forward that
    is handed the prepared geometry contains NO host synchronisation (torch's sync debug mode raises on one)."""
    from u2mkd_amd import train as T
    from u2mkd_amd.synth import synth_kd_batch
    batches = [T.kd_batch_to_device(synth_kd_batch(2500, 1, seed=40 + i, image_hw=(64, 112))) for i in range(3)]
    plain = _runner(1.0, 1.0)
    want = [float(plain(T.fresh_batch(b))) for b in batches]
    del plain
    torch.cuda.empty_cache()
    run = _runner(1.0, 1.0)
    got, cur = [], T.fresh_batch(batches[0])
    for i in range(3):
        nxt = T.fresh_batch(batches[i + 1]) if i + 1 < 3 else None
        got.append(float(run(cur, prefetch=nxt)))
        cur = nxt
    # the first step is exactly comparable (same weights); later ones carry lr-0.24 updates whose rounding noise
    # (MIOpen's camera branch is not bit-reproducible) the tiny scene amplifies -- as in test_gpu_graphs.py
    assert np.isclose(got[0], want[0], rtol=2e-3) and np.allclose(got, want, rtol=5e-2), (got, want)
    # sync-free forward on prepared geometry
    d = T.fresh_batch(batches[1])
    in_mod = run.model.prepare(run._in_mod(d))
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        out = run.net(in_mod)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert bool(torch.isfinite(out['stu']['x_vox']).all())


def test_plans_built_a_step_ahead_change_nothing(hip, monkeypatch):
    """U2MKD_PREFETCH_PLANS=1: the student's batch-only index structures (point <-> voxel maps with their scatter / gather
    plans, kernel-map schedules, window plans, point <-> pixel plans) built with the next batch's geometry instead of at first
    use.  (1) the forward of a batch prepared that way finds them all: no plan-building entry is called inside it from the
    second batch on; (2) the student's logits of the same batch with the same weights are what the lazy path computes."""
    from u2mkd_amd import _lib as L, kd as KD, train as T
    from u2mkd_amd.synth import synth_kd_batch
    from u2mkd_amd.torchsparse.nn import functional as F
    batches = [T.kd_batch_to_device(synth_kd_batch(2500 + 300 * i, 1, seed=60 + i, image_hw=(64, 112))) for i in range(2)]
    run = _runner(1.0, 1.0)
    run.model.eval()                                   # (no dropout / DropPath: the two forwards below are comparable)
    outs = {}
    # (u2mkd_c2l_plan is not in the list: the sampled pixel head builds its full-resolution plan on the camera stream)
    builders = ('u2mkd_csr_build', 'u2mkd_devoxelize_plan', 'u2mkd_sptr_plan_prepare', 'u2mkd_l2c_finish', 'u2mkd_l2c_keys',
                'u2mkd_tile_schedule', 'u2mkd_pairs_build', 'u2mkd_hash', 'u2mkd_kernel_hash', 'u2mkd_sptr_quant_coords')
    for ahead in (False, True):
        monkeypatch.setattr(F, '_PREFETCH_PLANS', ahead)
        with torch.no_grad():
            for i, b in enumerate(batches):        # (the first batch teaches the network's uses of its kernel maps and map sizes)
                in_mod = run.model.prepare(run._in_mod(T.fresh_batch(b)))
                seen = []
                real = L.call
                monkeypatch.setattr(L, 'call', lambda name, *a: (seen.append(name), real(name, *a))[1])
                out = run.model.model_s(in_mod['student'])
                monkeypatch.setattr(L, 'call', real)
                outs[(ahead, i)] = out['x_vox'].clone()
                if ahead and i == 1:
                    assert not [n for n in seen if n in builders], sorted(set(n for n in seen if n in builders))
                if not ahead:
                    assert any(n in builders for n in seen)
    for i in range(2):
        assert outs[(True, i)].shape == outs[(False, i)].shape
        assert float((outs[(True, i)] - outs[(False, i)]).abs().max()) <= 1e-4 * float(outs[(False, i)].abs().max()), i


def test_geometry_in_slices_gives_the_step_the_same_geometry(hip, monkeypatch):
    """train.KDStep with ``prefetch=``: the next batch's geometry queued in slices between the phases of the current step
    (U2MKD_STAGED_GEOMETRY=1, the single-rank default) against the one-piece form behind the backward (=0): the same voxel
    sets and coordinates for every batch, the frozen teacher's logits bit for bit, the first loss within rounding (the student's
    float atomics)."""
    from u2mkd_amd import train as T
    from u2mkd_amd.synth import synth_kd_batch
    from u2mkd_amd.torchsparse.nn import functional as F
    batches = [T.kd_batch_to_device(synth_kd_batch(3000 + 500 * i, 1, seed=31 + i, image_hw=(64, 112))) for i in range(3)]

    def run(staged):
        monkeypatch.setenv('U2MKD_STAGED_GEOMETRY', '1' if staged else '0')
        runner = _runner(1.0, 2.0)
        watch = T.TeacherWatch(runner.model.model_t)
        losses, geo = [], []
        cur = T.fresh_batch(batches[0])
        for i in range(4):
            nxt = T.fresh_batch(batches[(i + 1) % 3])
            watch.key = i
            losses.append(float(runner(cur, prefetch=nxt)))
            g = runner._queued[1]                         # the prepared in_mod of the next batch
            z, x0 = g['student']['_geometry']
            tz, tx0 = g['teacher']['_geometry']
            geo.append((x0.C.clone(), tx0.C.clone(), z.C.clone()))
            cur = nxt
        torch.cuda.synchronize()
        return losses, geo, [t for _, t in watch.log]
    a, b = run(False), run(True)
    assert abs(a[0][0] - b[0][0]) <= 1e-5 * abs(a[0][0]), (a[0], b[0])
    for ga, gb in zip(a[1], b[1]):
        for ta, tb in zip(ga, gb):
            assert torch.equal(ta, tb)
    assert torch.equal(a[2][0], b[2][0])               # the first step's teacher logits (later steps follow different students)


def test_teacher_issued_one_step_ahead_gives_the_same_teacher_and_the_same_first_loss(hip, monkeypatch):
    """kd.teacher_ahead (U2MKD_TEACHER_AHEAD: the frozen teacher's forward of batch k + 1 queued behind step k's backward, on
    the teacher's stream, picked up by TSDFull.forward of that batch): every batch's teacher logits bit for bit those of the
    in-line order (the teacher is frozen: eval-mode BatchNorm, no gradient), each computed exactly once, the first loss equal."""
    from u2mkd_amd import kd as KD, train as T
    from u2mkd_amd.synth import synth_kd_batch
    batches = [T.kd_batch_to_device(synth_kd_batch(3000 + 400 * i, 1, seed=71 + i, image_hw=(64, 112))) for i in range(3)]

    def run(ahead):
        monkeypatch.setattr(KD, '_TEACHER_AHEAD', ahead)
        monkeypatch.setenv('U2MKD_STAGED_GEOMETRY', '1')
        runner = _runner(1.0, 2.0)
        watch = T.TeacherWatch(runner.model.model_t)
        losses = []
        cur = dict(T.fresh_batch(batches[0]), _key=0)
        for i in range(4):
            nxt = dict(T.fresh_batch(batches[(i + 1) % 3]), _key=(i + 1) % 3)
            losses.append(float(runner(cur, prefetch=nxt)))
            cur = nxt
        torch.cuda.synchronize()
        return losses, list(watch.log)
    (la, ta), (lb, tb) = run(0), run(1)
    assert abs(la[0] - lb[0]) <= 1e-5 * abs(la[0]), (la, lb)
    # in-line: batches 0, 1, 2, 0; ahead: 0 (in line), then 1, 2, 0 and the queued-but-unused 1 of the last call
    assert [k for k, _ in ta] == [0, 1, 2, 0] and [k for k, _ in tb] == [0, 1, 2, 0, 1]
    for (ka, a), (kb, b) in zip(ta, tb):
        assert ka == kb and torch.equal(a, b), ka
    monkeypatch.setattr(KD, '_TEACHER_AHEAD', 2)
    lc = run(2)[0]
    assert abs(la[0] - lc[0]) <= 1e-5 * abs(la[0])
