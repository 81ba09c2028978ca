"""Golden vectors made by the reference's own SPVCNN_SPFORMER / SphereFormer modules
(core/models/nuscenes/spvcnn_spformer.py, core/models/sphereformer/spherical_transformer.py,
imported in the build container over the CPU oracle operators; tests/golden/make_golden.py).
CPU: the oracle restatement must reproduce them to fp32 round-off (5e-5; including the
quant_size_sphere aliasing, which changes logits by O(1) when wrong).  GPU: the HIP model must match within 1e-3."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import spformer_ref as R
from oracle import spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd.synth import synth_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
BLK = 'transformer_blocks.1.attn.'


# L2-relative bound of the sampled parameter gradients against the fp32 CPU golden.  Measured on MI355X: 2.2e-3 .. 5.3e-3.  The
# convolution-only network is held to 1e-3 against an fp64 arbiter (tests/grad_arbiter.py); the attention path has none:
# the reference's sptr operators force fp32 (spherical_transformer.py:221-244), so the CPU golden itself carries fp32
# rounding of the same size as the HIP evaluation's -- two fp32 evaluations are compared here, at a bound 2x the
# worst measurement.
GRAD_GATE = 1.2e-2


def _inputs():
    b = synth_batch(2000, 2, seed=33)
    return tuple(torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))


def test_oracle_spformer_matches_reference_class():
    gold = np.load(os.path.join(G, 'spformer_cr10_4000.npz'))
    feats, coords, labels = _inputs()
    m = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))).train()
    m.dropout.p = 0.0
    out = m({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
    # same algorithm, same operation order; only the BLAS thread partition of the dense layers may differ
    assert np.abs(out.detach().numpy() - gold['logits']).max() < 5e-5
    loss = O.mix_lovasz_cross_entropy(out, labels)
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-6
    loss.backward()
    g = dict(m.named_parameters())
    # gradients: CPU index_add_ accumulation order varies run to run -> not bit-stable; 1e-4 of the max
    for name, key in (('relative_pos_query_table', 'grad_tq'), ('relative_pos_value_table_sphere', 'grad_tv_sphere'),
                      ('qkv.weight', 'grad_qkv')):
        a, b = g[BLK + name].grad.numpy(), gold[key]
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max(), name


def test_spformer_state_dict_keys_match_reference():
    from u2mkd_amd import lidar
    with open(os.path.join(G, 'spformer_cr10_keys.json')) as f:
        keys = json.load(f)
    m = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=1.0))
    sd = m.state_dict()
    assert list(sd.keys()) == list(keys.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == keys
    # forward-time hyper-parameters incl. the aliased spherical quant size
    for i, blk in enumerate(m.transformer_blocks):
        a = blk.attn
        assert np.allclose(a.window_size, 0.3 * 2 ** i) and np.allclose(a.quant_size, 0.0125 * 2 ** i)
        assert np.allclose(a.window_size_sphere, [2 * 2 ** i, 2 * 2 ** i, 120])
        assert np.allclose(a.quant_size_sphere, [16 / 12, 16 / 12, 5.0])
        assert a.quant_grid_length == 24 and a.quant_grid_length_sphere == 24


@pytest.mark.gpu
def test_hip_spformer_matches_reference_golden(hip):
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    gold = np.load(os.path.join(G, 'spformer_cr10_4000.npz'))
    feats, coords, labels = _inputs()
    ref = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0)))
    model = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=1.0, drop_path_rate=0.0))
    model.load_state_dict(ref.state_dict())
    model.cuda().train()
    model.dropout.p = 0.0
    out = model({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox']
    err = float((out.detach().cpu() - torch.from_numpy(gold['logits'])).abs().max())
    assert err < 1e-3, err
    loss = MixLovaszCrossEntropy(ignore_index=0)(out, labels.cuda())
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-3
    loss.backward()
    g = dict(model.named_parameters())
    for name, key in ((BLK + 'relative_pos_query_table', 'grad_tq'),
                      (BLK + 'relative_pos_value_table_sphere', 'grad_tv_sphere'), (BLK + 'qkv.weight', 'grad_qkv')):
        a, b = g[name].grad.cpu().double(), torch.from_numpy(gold[key]).double()
        rel = float((a - b).norm() / b.norm())
        print('SPFORMER-GRAD', name, '%.3e' % rel)
        assert rel < GRAD_GATE, name
