"""Rows f2 / f4 of SURVEY.md 8f on the GPU, against the oracle: the eval branch of the two trainers (voxel logits ->
per-point predictions -> MeanIoU) through the HIP models.

* KD trainer (core/nusc_trainers.py:367-418): `train.KDStep.evaluate` against a golden made by the reference's own
  model class in eval mode (`tests/golden/make_golden.py kd_eval`: logits of the voxel head, the pixel head and the
  teacher, and the three per-point prediction vectors of the reference loop);
* teacher trainer on a multi-sweep scene (core/spformer_trainer.py:95-117, `keyframe_mask_full`):
  `train.LidarStep.evaluate` against the CPU oracle model evaluated here, the reference loop restated in
  tests/test_evaluate.py and its MeanIoU arithmetic (core/callbacks.py:118-160)."""
import os

import numpy as np
import pytest
import torch

from oracle import spformer_ref as R, spvcnn_ref as O, torchsparse_cpu as ots
from test_evaluate import _reference_miou, _reference_predictions
from u2mkd_amd.synth import synth_batch, synth_eval_feed, synth_kd_batch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MARGIN = 2e-3       # an arg-max may differ only where the golden's two largest logits are closer than twice the 1e-3 gate


def _same_predictions(got, want, logits_want):
    """Predictions equal wherever the golden logits decide by more than MARGIN (rows of identical logits, e.g. the
    all-zero rows of points no camera sees, arg-max to the same first index on both sides)."""
    srt = np.sort(logits_want, 1)
    decided = (srt[:, -1] - srt[:, -2] >= MARGIN) | (srt[:, -1] == srt[:, 0])
    diff = got != want
    assert not np.any(diff & decided), (int((diff & decided).sum()), 'predictions differ on decided rows')
    return int(diff.sum())


def test_kd_eval_branch_matches_reference_golden(hip):
    from u2mkd_amd import kd, lidar, train as T
    from u2mkd_amd.evaluate import MeanIoU
    gold = np.load(os.path.join(G, 'kd_eval_cr10_3000.npz'))
    seed = int(gold['seed'])
    b = synth_kd_batch(1500, 2, seed=seed, image_hw=(64, 112))
    b['student']['images'] = ((b['student']['images'] / 255.0 - 0.45) / 0.225).astype(np.float32)
    f = {k: torch.from_numpy(v).cuda() for k, v in synth_eval_feed(b, seed).items()}
    sp = {k: v for k, v in lidar.spformer_kwargs(drop_path_rate=0.0).items() if k not in ('cr', 'in_channel', 'num_classes')}
    model = O.fill_state_by_name(kd.TSDFull(cr=1.0, cr_t=1.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp,
                                            debug_val=True), conv2d_he=True).cuda()
    runner = T.KDStep(model)
    model.eval()
    d = T.kd_batch_to_device(b)
    with pytest.raises(AssertionError):
        model.train()
        runner.evaluate(d, f['s_inverse_map'], f['s_inverse_batch'], f['targets_mapped'], f['label_fov'])
    model.eval()
    with torch.no_grad():
        out = model(runner._in_mod(d))
    for got, key in ((out['stu']['x_vox'], 'x_vox'), (out['stu']['x_pix'], 'x_pix'), (out['t']['x_vox'], 'x_vox_t')):
        err = float((got.cpu() - torch.from_numpy(gold[key])).abs().max())
        print('EVAL-PARITY', key, 'max abs err %.2e' % err, '(logit range %.1f)' % float(np.abs(gold[key]).max()))
        assert err < 1e-3, (key, err)
    ret = runner.evaluate(d, f['s_inverse_map'], f['s_inverse_batch'], f['targets_mapped'], f['label_fov'],
                          f['t_inverse_batch'], f['targets_mapped_t'])
    inv_s, ib_s = f['s_inverse_map'].cpu().numpy(), f['s_inverse_batch'].cpu().numpy()
    sb = b['student']['coords'][:, -1]
    tb = b['teacher']['coords'][:, -1]
    rows_s = np.concatenate([np.nonzero(sb == i)[0][inv_s[ib_s == i]] for i in range(2)])
    rows_t = np.concatenate([np.nonzero(tb == i)[0][b['teacher']['inverse_map'][f['t_inverse_batch'].cpu().numpy() == i]]
                             for i in range(2)])
    n_diff = 0
    for key, lk, rows in (('outputs_vox', 'x_vox', rows_s), ('outputs_pix', 'x_pix', rows_s), ('outputs_vox_t', 'x_vox_t', rows_t)):
        n_diff += _same_predictions(ret[key].cpu().numpy(), gold[key], gold[lk][rows])
    print('EVAL-PARITY predictions that differ on undecided rows:', n_diff)
    assert torch.equal(ret['targets'].cpu(), f['targets_mapped'].cpu()) and torch.equal(ret['targets_fov'].cpu(), f['label_fov'].cpu())
    assert torch.equal(ret['targets_t'].cpu(), f['targets_mapped_t'].cpu())
    for out_key, tgt_key in (('outputs_vox', 'targets'), ('outputs_pix', 'targets_fov'), ('outputs_vox_t', 'targets_t')):
        m = MeanIoU(17, 0, out_key, tgt_key)
        m.after_step(ret)
        got, _ = m.after_epoch()
        want, _ = _reference_miou([(gold[out_key], ret[tgt_key].cpu().numpy())], 17, 0)
        assert abs(got - want) <= (1e-12 if n_diff == 0 else 2e-3), (out_key, got, want)


def test_teacher_eval_branch_on_a_multisweep_scene_matches_the_oracle(hip):
    from u2mkd_amd import lidar, train as T
    from u2mkd_amd.evaluate import MeanIoU
    ms = np.load(os.path.join(G, 'teacher_multisweep_cr10_6000.npz'))
    seed = int(ms['seed'])                                   # the multi-sweep scene known to sit off the quantiser edges
    b = synth_batch(3000, 2, seed=seed, sweeps=3)
    feats, coords = torch.from_numpy(b['feats']), torch.from_numpy(b['coords'])
    ref = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))).eval()
    with torch.no_grad():
        want_logits = ref({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
    rng = np.random.default_rng(seed)
    vb = coords[:, -1]
    nv = [int((vb == i).sum()) for i in range(2)]
    npts = [int(1.4 * n) for n in nv]
    inv = torch.cat([torch.from_numpy(rng.integers(0, n, p)) for n, p in zip(nv, npts)])
    ib = torch.cat([torch.full((p,), i) for i, p in enumerate(npts)])
    perm = torch.from_numpy(rng.permutation(len(ib)))       # the feed need not be grouped by scene
    inv, ib = inv[perm], ib[perm]
    labels = torch.from_numpy(rng.integers(0, 17, len(ib)))
    kfm = torch.from_numpy(rng.random(len(ib)) < 0.4)        # keyframe_mask_full of the multi-sweep loader
    order = torch.argsort(ib, stable=True)
    want_pred = _reference_predictions(want_logits, vb, inv, ib, kfm[order])
    want_t = labels[order][kfm[order]]

    model = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=1.0, drop_path_rate=0.0))
    model.load_state_dict(ref.state_dict())
    model.cuda()
    runner = T.LidarStep(model)
    model.eval()
    ret = runner.evaluate(feats.cuda(), coords.cuda(), inv.cuda(), ib.cuda(), labels.cuda(), kfm.cuda())
    with torch.no_grad():
        from u2mkd_amd import torchsparse as ts
        got_logits = model({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox'].cpu()
    err = float((got_logits - want_logits).abs().max())
    print('EVAL-PARITY teacher multi-sweep eval logits max abs err %.2e' % err)
    assert err < 1e-3
    assert torch.equal(ret['targets'].cpu(), want_t)
    # rows the predictions were read from, in the reference's order
    rows = torch.cat([(vb == i).nonzero().squeeze(1)[inv[ib == i]] for i in range(2)])[kfm[order]]
    n_diff = _same_predictions(ret['outputs_vox'].cpu().numpy(), want_pred.numpy(), want_logits[rows].numpy())
    m = MeanIoU(17, 0, 'outputs_vox', 'targets')
    m.after_step(ret)
    got, _ = m.after_epoch()
    want, _ = _reference_miou([(want_pred.numpy(), want_t.numpy())], 17, 0)
    assert abs(got - want) <= (1e-12 if n_diff == 0 else 2e-3), (got, want)
