"""run_training.py end to end on the GPU: the two stages of the reference's pipeline (README: train_spformer.py, then
train_lc_nusc_tsd_full.py with `model.teacher_pretrain` = the first stage's best checkpoint), from configuration files in
the reference's layout -- once on synthetic scenes, once on the synthetic on-disk nuScenes tree of the loader's tests."""
import json
import os
import socket
import subprocess
import sys

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


def test_two_ranks_on_one_gpu_train_the_kd_student(hip, tmp_path):
    """The launcher contract of the reference (torchrun, one rank per GPU) with two ranks: SyncBatchNorm conversion of
    the student + DDP inside train.KDStep, a DistributedSampler shard per rank, the all-reduced MeanIoU, rank 0 alone
    writing the checkpoints.  RCCL refuses two ranks on one device, so the group is gloo (`--backend gloo`): the kernels
    are the product's, the collectives travel through host memory."""
    cfgs = _configs(tmp_path, im_cr=0.08)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    run = str(tmp_path / 'kd_ddp')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(repo, 'run_training.py'), os.path.join(cfgs, 'tsd.yaml'), '--run-dir', run,
           '--backend', 'gloo', '--synthetic', '2000', '--max-iters', '2', '--model.in_channel_t', '4', '--optimizer.lr', '0.01']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    p = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert 'world 2' in p.stdout
    hist = json.load(open(os.path.join(run, 'history.json')))
    assert len(hist) == 1 and hist[0]['loss'] == hist[0]['loss'] and 0.0 <= hist[0]['iou-vox/val'] <= 1.0
    ck = torch.load(os.path.join(run, 'checkpoints', 'step-2.pt'), map_location='cpu', weights_only=False)
    assert all(k.startswith('module.') for k in ck['model'])                   # saved from the DDP wrapper, as the reference does
    assert any('model_s' in k and k.endswith('.kernel') for k in ck['model'])


def test_teacher_trainer_on_the_on_disk_tree_with_multisweep_masks(hip, tmp_path, tree):  # noqa: F811
    """Stage 1 on the loader's output: the teacher half of every sample (3 aggregated sweeps -> `keyframe_mask` for the
    loss, `keyframe_mask_full` for the evaluation, core/spformer_trainer.py:64-68,95-117)."""
    import run_training
    root, ver = tree
    cfgs = _configs(tmp_path, root=root, version=ver, im_cr=0.08)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        h = run_training.main([os.path.join(cfgs, 'spformer.yaml'), '--run-dir', str(tmp_path / 'teacher'), '--non-dist',
                               '--optimizer.lr', '0.01'])
    finally:
        os.chdir(cwd)
    assert len(h) == 1 and h[0]['loss'] == h[0]['loss'] and 0.0 <= h[0]['iou/val/vox'] <= 1.0
    assert sorted(os.listdir(tmp_path / 'teacher' / 'checkpoints')) == ['max-iou-val-vox.pt', 'step-1.pt']


def test_without_non_dist_a_weight_path_resumes_the_whole_trainer(hip, tmp_path, capsys):
    """core/nusc_trainers.py:174-177: in the (default) distributed mode `--weight-path` restores model, optimizer, LR
    schedule and loss scaler; the run continues with the epoch after the checkpoint's."""
    import run_training
    cfgs = _configs(tmp_path, im_cr=0.08)
    common = ['--synthetic', '2000', '--max-iters', '2', '--model.in_channel_t', '4', '--optimizer.lr', '0.01']
    run1 = str(tmp_path / 'a')
    run_training.main([os.path.join(cfgs, 'tsd.yaml'), '--run-dir', run1] + common)
    ck = os.path.join(run1, 'checkpoints', 'step-2.pt')
    saved = torch.load(ck, map_location='cpu', weights_only=False)
    assert saved['epoch'] == 1 and saved['scheduler']['last_epoch'] == 2
    capsys.readouterr()
    h = run_training.main([os.path.join(cfgs, 'tsd.yaml'), '--run-dir', str(tmp_path / 'b'), '--weight-path', ck, '--num_epochs', '2'] + common)
    out = capsys.readouterr().out
    assert 'resumed the trainer state at epoch 2, step 2' in out and [r['epoch'] for r in h] == [2]
    again = torch.load(os.path.join(tmp_path, 'b', 'checkpoints', 'step-4.pt'), map_location='cpu', weights_only=False)
    assert again['epoch'] == 2 and again['global_step'] == 4 and again['scheduler']['last_epoch'] == 4
    mom = [s for s in again['optimizer']['state'].values() if 'momentum_buffer' in s]
    assert mom and all(torch.isfinite(s['momentum_buffer']).all() for s in mom)
