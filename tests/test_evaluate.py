"""Row f3: the eval branch (voxel logits -> per-point predictions) and MeanIoU against direct restatements of the
reference loops (core/nusc_trainers.py:367-389, core/callbacks.py:118-160).  CPU; the 2-rank sum over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from u2mkd_amd.evaluate import MeanIoU, voxel_logits_to_point_predictions

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_predictions(logits, vb, inv, ib, kfm=None):
    outs = []
    for idx in range(int(ib.max()) + 1):                     # nusc_trainers.py:375-388
        cur_scene_pts = (vb == idx).numpy()
        cur_inv = inv[ib == idx].numpy()
        outs.append(logits[cur_scene_pts][cur_inv].argmax(1))
    o = torch.cat(outs, 0)
    return o[kfm] if kfm is not None else o


def test_point_predictions_equal_reference_loop():
    g = torch.Generator().manual_seed(1)
    nv = [50, 0, 73]                                          # a scene without voxels in between
    vb = torch.cat([torch.full((n,), i) for i, n in enumerate(nv)])
    vb = vb[torch.randperm(len(vb), generator=g)]             # voxels of the scenes interleaved
    logits = torch.randn(len(vb), 17, generator=g)
    npts = [120, 0, 200]
    ib = torch.cat([torch.full((n,), i) for i, n in enumerate(npts)])
    inv = torch.cat([torch.randint(0, max(v, 1), (n,), generator=g) for v, n in zip(nv, npts)])
    got = voxel_logits_to_point_predictions(logits, vb, inv, ib)
    assert torch.equal(got, _reference_predictions(logits, vb, inv, ib))
    kfm = torch.rand(len(ib), generator=g) < 0.7
    assert torch.equal(voxel_logits_to_point_predictions(logits, vb, inv, ib, kfm), _reference_predictions(logits, vb, inv, ib, kfm))


def _reference_miou(steps, num_classes, ignore):
    seen, corr, pos = np.zeros(num_classes), np.zeros(num_classes), np.zeros(num_classes)
    for o, t in steps:                                        # callbacks.py:118-137
        o, t = o[t != ignore], t[t != ignore]
        for i in range(num_classes):
            seen[i] += np.sum(t == i); corr[i] += np.sum((t == i) & (o == t)); pos[i] += np.sum(o == i)
    ious = []
    for i in range(num_classes):                              # callbacks.py:146-156
        if seen[i] == 0:
            if i == ignore:
                continue
            ious.append(1)
        else:
            ious.append(corr[i] / (seen[i] + pos[i] - corr[i]))
    return float(np.mean(ious)), ious


def _steps(seed, n=4):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        t = rng.integers(0, 17, 5000)
        t[t == 9] = 3                                          # class 9 never seen -> counts as IoU 1
        o = np.where(rng.random(5000) < 0.6, t, rng.integers(0, 17, 5000))
        out.append((o, t))
    return out


def test_mean_iou_equals_reference_arithmetic():
    steps = _steps(0)
    m = MeanIoU(17, 0)
    for o, t in steps:
        m.after_step({'outputs': torch.from_numpy(o), 'targets': torch.from_numpy(t)})
    miou, ious = m.after_epoch()
    want, want_ious = _reference_miou(steps, 17, 0)
    assert miou == pytest.approx(want, abs=1e-12) and len(ious) == 16
    assert np.allclose(ious, want_ious, atol=1e-12) and ious[8] == 1       # class 9 (index 8 after dropping ignore)
    m.before_epoch()
    assert m.after_epoch()[0] == 1.0                                       # nothing seen: every kept class counts 1


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from u2mkd_amd import distributed as D
    D.init_from_env('gloo')
    m = MeanIoU(17, 0)
    for o, t in _steps(10 + rank):
        m.after_step({'outputs': torch.from_numpy(o), 'targets': torch.from_numpy(t)})
    miou, _ = m.after_epoch()
    torch.save(miou, os.path.join(out_dir, f'm{rank}.pt'))
    D.shutdown()


@pytest.mark.timeout(300)
def test_mean_iou_sums_over_ranks(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want, _ = _reference_miou(_steps(10) + _steps(11), 17, 0)
    for r in range(2):
        assert torch.load(os.path.join(tmp_path, f'm{r}.pt')) == pytest.approx(want, abs=1e-12)
