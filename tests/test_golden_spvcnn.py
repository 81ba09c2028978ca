"""Golden vectors made by the reference's own SPVCNN / MixLovaszCrossEntropy
classes (tests/golden/make_golden.py, run in the build container where
/root/reference exists).  CPU: pins the oracle restatement bit-for-bit and the
product model's state-dict keys.  GPU: pins the HIP model within 1e-3."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd.synth import synth_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KW = dict(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05)


def _inputs():
    b = synth_batch(4000, 1, seed=21)
    return tuple(torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))


def test_oracle_model_matches_reference_class():
    gold = np.load(os.path.join(G, 'spvcnn_cr05_4000.npz'))
    feats, coords, labels = _inputs()
    m = O.fill_state_by_name(O.SPVCNN(**KW)).train()
    m.dropout.p = 0.0
    out = m({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
    # same algorithm and operation order as the reference class; only the BLAS thread partition may differ
    assert np.abs(out.detach().numpy() - gold['logits']).max() < 5e-5
    loss = O.mix_lovasz_cross_entropy(out, labels)
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-6
    loss.backward()
    g = dict(m.named_parameters())
    for name, key, sl in (('stem.0.kernel', 'grad_stem0', slice(None)), ('classifier_vox.0.weight', 'grad_cls', slice(None)),
                          ('vox_ups.3.1.1.net.3.kernel', 'grad_up3', 13)):
        a, b = g[name].grad.numpy()[sl], gold[key]
        assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max(), name


def test_state_dict_keys_match_reference():
    from u2mkd_amd import lidar
    with open(os.path.join(G, 'spvcnn_cr05_keys.json')) as f:
        keys = json.load(f)
    sd = lidar.SPVCNN(**KW).state_dict()
    assert list(sd.keys()) == list(keys.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == keys
    assert list(O.SPVCNN(**KW).state_dict().keys()) == list(keys.keys())


def test_losses_match_reference_criterion():
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    gold = np.load(os.path.join(G, 'lovasz_ce.npz'))
    y = torch.from_numpy(gold['y'])
    for fn in (MixLovaszCrossEntropy(ignore_index=0), lambda a, b: O.mix_lovasz_cross_entropy(a, b, 0)):
        x = torch.from_numpy(gold['x']).clone().requires_grad_(True)
        loss = fn(x, y)
        loss.backward()
        assert abs(float(loss.detach()) - float(gold['loss'])) < 2e-6
        assert float((x.grad - torch.from_numpy(gold['grad'])).abs().max()) < 1e-7


@pytest.mark.gpu
def test_hip_model_matches_reference_golden(hip):
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    gold = np.load(os.path.join(G, 'spvcnn_cr05_4000.npz'))
    feats, coords, labels = _inputs()
    ref = O.fill_state_by_name(O.SPVCNN(**KW))
    model = lidar.SPVCNN(**KW)
    model.load_state_dict(ref.state_dict())
    model.cuda().train()
    model.dropout.p = 0.0
    out = model({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox']
    err = float((out.detach().cpu() - torch.from_numpy(gold['logits'])).abs().max())
    assert err < 1e-3, err
    loss = MixLovaszCrossEntropy(ignore_index=0)(out, labels.cuda())
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-3
