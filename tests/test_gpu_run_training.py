"""run_training.py end to end on the GPU: the two stages of the reference's pipeline (README: train_spformer.py, then
train_lc_nusc_tsd_full.py with `model.teacher_pretrain` = the first stage's best checkpoint), from configuration files in
the reference's layout -- once on synthetic scenes, once on the synthetic on-disk nuScenes tree of the loader's tests."""
import os

import pytest
import torch
import yaml

from test_builder import ROOT, SPFORMER, TRAIN_DEFAULT, TSD
from test_nuscenes_loader import tree  # noqa: F401  (module-scoped fixture: the on-disk tables, sweeps, images)

pytestmark = pytest.mark.gpu


def _configs(tmp_path, **dataset):
    d = tmp_path / 'configs' / 'nuscenes' / 'train'
    d.mkdir(parents=True)
    root = dict(ROOT, num_epochs=1, batch_size=2, workers_per_gpu=0)
    root['dataset'] = dict(root['dataset'], **dataset)
    small = {'num_epochs': 1, 'batch_size': 2}
    teacher = dict(SPFORMER, **small)
    teacher['model'] = dict(SPFORMER['model'], cr=0.5)
    kd = dict(TSD, **small)
    kd['model'] = dict(TSD['model'], cr=0.5, cr_t=0.5, imagenet_pretrain=None)
    kd['dataset'] = {'name': 'lc_semantic_nusc_tsd_full', 'flip': True, 'im_drop': 3, 'multisweeps': {'num_sweeps': 2, 'only_past': False}}
    teacher['dataset'] = {'name': 'semantic_nusc', 'multisweeps': {'num_sweeps': 2, 'only_past': False}}
    for path, content in ((tmp_path / 'configs' / 'nuscenes' / 'default.yaml', root), (d / 'default.yaml', TRAIN_DEFAULT),
                          (d / 'spformer.yaml', teacher), (d / 'tsd.yaml', kd)):
        path.write_text(yaml.safe_dump(content))
    return str(d)


def test_two_stage_pipeline_on_synthetic_scenes(hip, tmp_path, capsys):
    import run_training
    cfgs = _configs(tmp_path, im_cr=0.08)
    run1 = str(tmp_path / 'teacher')
    h = run_training.main([os.path.join(cfgs, 'spformer.yaml'), '--run-dir', run1, '--non-dist', '--synthetic', '2500', '--max-iters', '2'])
    assert len(h) == 1 and h[0]['loss'] == h[0]['loss'] and 0.0 <= h[0]['iou/val/vox'] <= 1.0
    best = os.path.join(run1, 'checkpoints', 'max-iou-val-vox.pt')
    assert sorted(os.listdir(os.path.join(run1, 'checkpoints'))) == ['max-iou-val-vox.pt', 'step-2.pt']
    ck = torch.load(best, map_location='cpu', weights_only=False)
    assert {'model', 'scaler', 'optimizer', 'scheduler', 'epoch', 'global_step'} <= set(ck) and ck['global_step'] == 2
    assert any(k.endswith('.kernel') for k in ck['model'])               # the reference's conv weight key

    run2 = str(tmp_path / 'kd')
    h = run_training.main([os.path.join(cfgs, 'tsd.yaml'), '--run-dir', run2, '--non-dist', '--synthetic', '2500', '--max-iters', '2',
                           '--model.in_channel_t', '4', '--model.teacher_pretrain', best, '--optimizer.lr', '0.01'])
    out = capsys.readouterr().out
    assert 'weights: teacher_pretrain_weight' in out
    assert {'iou-vox/val', 'iou-pix/val', 'iou-vox-t/val'} <= set(h[0]) and h[0]['loss'] == h[0]['loss']
    assert sorted(os.listdir(os.path.join(run2, 'checkpoints'))) == ['max-iou-pix-val.pt', 'max-iou-vox-val.pt', 'step-2.pt']
    # the teacher inside the KD checkpoint is the stage-1 network, untouched by the KD steps (frozen)
    kd_ck = torch.load(os.path.join(run2, 'checkpoints', 'step-2.pt'), map_location='cpu', weights_only=False)['model']
    for k, v in ck['model'].items():
        if v.dtype.is_floating_point and 'num_batches_tracked' not in k:
            assert torch.equal(kd_ck['model_t.' + k], v), k
    # resuming from a trainer checkpoint takes precedence over the teacher's file (core/nusc_trainers.py:173-201)
    run_training.main([os.path.join(cfgs, 'tsd.yaml'), '--run-dir', str(tmp_path / 'kd2'), '--non-dist', '--synthetic', '2500',
                       '--max-iters', '1', '--model.in_channel_t', '4', '--model.teacher_pretrain', best,
                       '--weight-path', os.path.join(run2, 'checkpoints', 'step-2.pt')])
    assert 'weights: weight_path' in capsys.readouterr().out


def test_kd_trainer_on_the_on_disk_tree(hip, tmp_path, tree):  # noqa: F811
    import run_training
    root, ver = tree
    cfgs = _configs(tmp_path, root=root, version=ver, im_cr=0.08)
    cwd = os.getcwd()
    os.chdir(tmp_path)                      # no ./data/nuscenes/*_official.npy here: both splits hold every sample
    try:
        h = run_training.main([os.path.join(cfgs, 'tsd.yaml'), '--run-dir', str(tmp_path / 'run'), '--non-dist',
                               '--model.in_channel_t', '4', '--optimizer.lr', '0.01'])
    finally:
        os.chdir(cwd)
    assert len(h) == 1 and h[0]['loss'] == h[0]['loss']
    assert all(0.0 <= h[0][k] <= 1.0 for k in ('iou-vox/val', 'iou-pix/val', 'iou-vox-t/val'))
    assert os.path.isfile(tmp_path / 'run' / 'checkpoints' / 'step-1.pt') and os.path.isfile(tmp_path / 'run' / 'history.json')
