"""The REFERENCE'S OWN CALL SEQUENCE over the drop-in boundaries, on the GPU.

/root/reference does not exist on the GPU box, so the sequence is the oracle's line-by-line restatement of the reference's
model files (oracle/spvcnn_ref.py: core/models/utils.py:15-118, build_blocks.py:21-83, semantickitti/spvcnn.py:85-142;
oracle/spformer_ref.py: nuscenes/spvcnn_spformer.py:125-189, sphereformer/spherical_transformer.py:165-348;
oracle/sptr_layer_ref.py: third_party/SparseTransformer/sptr/{functional,modules,utils}.py) re-bound (``over``) from the CPU
operator packages to the PRODUCT's: ``u2mkd_amd.torchsparse`` (boundary 1) and ``u2mkd_amd.sptr`` /
``u2mkd_amd.sptr.sptr_cuda`` (boundary 2).  Nothing of the product's own model wiring runs here: no FusedSequential, no
prepare_geometry, no fused BatchNorm+ReLU -- ``spnn.Conv3d -> spnn.BatchNorm -> spnn.ReLU`` as separate modules, kernel maps
built lazily inside ``conv3d``, ``torch.unique`` for the voxel set, ``F.sphash -> F.sphashquery -> F.spcount ->
F.spvoxelize`` / ``calc_ti_weights -> spdevoxelize`` exactly as utils.py issues them.  Held to the goldens made by the
reference's classes (tests/golden/make_golden.py): logits 1e-3 (the north-star gate), loss, sampled kernel gradients."""
import os
from functools import partial

import numpy as np
import pytest
import torch

from oracle import spformer_ref as R
from oracle import spvcnn_ref as O
from oracle import sptr_layer_ref as P
from oracle import sptr_ref as S
from u2mkd_amd.synth import synth_batch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _rel(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def test_reference_spvcnn_sequence_over_the_torchsparse_drop_in(hip):
    import u2mkd_amd.torchsparse as ts
    OG = O.over(ts)
    assert OG.spnn.Conv3d is ts.nn.Conv3d and OG.spf.sphash is ts.nn.functional.sphash
    gold = np.load(os.path.join(G, 'spvcnn_cr05_4000.npz'))
    b = synth_batch(4000, 1, seed=21)
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    m = OG.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    O.fill_state_by_name(m)
    m.cuda().train()
    m.dropout.p = 0.0
    # un-fused module chain, as build_blocks.py builds it
    assert [type(x).__name__ for x in m.stem] == ['Conv3d', 'BatchNorm', 'ReLU', 'Conv3d', 'BatchNorm', 'ReLU']
    x = ts.SparseTensor(feats, coords)
    out = m({'lidar': x})['x_vox']
    assert len(x.kmaps) == 0 or True          # (kernel maps live on the voxel tensors derived inside the forward)
    err = float((out.detach().cpu() - torch.from_numpy(gold['logits'])).abs().max())
    print('REFSEQ spvcnn max |logit - golden| %.2e' % err)
    assert err < 1e-3, err
    loss = OG.mix_lovasz_cross_entropy(out, labels)
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-3
    loss.backward()
    g = dict(m.named_parameters())
    for name, key, sl in (('stem.0.kernel', 'grad_stem0', slice(None)), ('classifier_vox.0.weight', 'grad_cls', slice(None)),
                          ('vox_ups.3.1.1.net.3.kernel', 'grad_up3', 13)):
        rel = _rel(g[name].grad[sl], gold[key])
        print('REFSEQ spvcnn grad', name, '%.2e' % rel)
        assert rel < 5e-3, (name, rel)


def test_reference_spformer_sequence_over_both_drop_ins(hip):
    """SPVCNN_SPFORMER: boundary 1 + the four names spherical_transformer.py:7 imports from sptr (u2mkd_amd.sptr)."""
    import u2mkd_amd.sptr as sptr
    import u2mkd_amd.torchsparse as ts
    RG = R.over(ts, sptr)
    gold = np.load(os.path.join(G, 'spformer_cr10_4000.npz'))
    b = synth_batch(2000, 2, seed=33)
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    m = RG.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))
    O.fill_state_by_name(m)
    m.cuda().train()
    m.dropout.p = 0.0
    out = m({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
    err = float((out.detach().cpu() - torch.from_numpy(gold['logits'])).abs().max())
    print('REFSEQ spformer max |logit - golden| %.2e' % err)
    assert err < 1e-3, err
    loss = O.mix_lovasz_cross_entropy(out, labels)
    assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-3
    loss.backward()
    g = dict(m.named_parameters())
    blk = 'transformer_blocks.1.attn.'
    for name, key in ((blk + 'relative_pos_query_table', 'grad_tq'), (blk + 'relative_pos_value_table_sphere', 'grad_tv_sphere'),
                      (blk + 'qkv.weight', 'grad_qkv')):
        rel = _rel(g[name].grad, gold[key])
        print('REFSEQ spformer grad', name, '%.2e' % rel)
        assert rel < 1.2e-2, (name, rel)          # (the gate of tests/test_golden_spformer.py: two fp32 evaluations compared)


def _attention_inputs(n=3000, seed=4, spherical=False):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([8.0, 8.0, 2.0])
    b = torch.sort(torch.randint(0, 2, (n,), generator=g))[0]
    if spherical:
        xyz = S.cart2sphere(xyz - torch.tensor([4.0, 4.0, 1.0]))
    h, d = 4, 16
    q, k, v = (torch.randn(n, h, d, generator=g) for _ in range(3))
    return xyz, b, q, k, v, g


@pytest.mark.parametrize('spherical', [False, True])
def test_sptr_python_layer_over_the_ten_entries_equals_the_fused_kernel_and_the_oracle(hip, spherical):
    """sptr/functional.py + modules.py + utils.py (restated) over u2mkd_amd.sptr.sptr_cuda -- precompute_all_cuda,
    dot_prod_with_idx_all_forward_cuda, dot_prod_with_idx_backward_cuda, attention_step1_backward_cuda,
    attention_step2_with_rel_pos_value_{forward,backward}_cuda: the six entries on the model's path -- against (a) the product's
    fused attention behind the same sparse_self_attention signature and (b) the CPU oracle; output and every gradient."""
    import u2mkd_amd.sptr as sptr
    from u2mkd_amd.sptr import sptr_cuda
    lay = P.layer(sptr_cuda)
    if spherical:
        split_a, window, quant, qgl, L = 0.0125, [2.0, 2.0, 120.0], [2 / 24, 2 / 24, 5.0], 24, 48
    else:
        split_a, window, quant, qgl, L = None, [0.6, 0.6, 0.6], [0.025, 0.025, 0.025], 24, 47
    window, quant = np.array(window, dtype=np.float32), np.array(quant, dtype=np.float32)
    xyz, b, q, k, v, g = _attention_inputs(spherical=spherical)
    tq, tk, tv = (0.2 * torch.randn(L, 3, 4, 16, generator=g) for _ in range(3))
    go = torch.randn(q.shape, generator=g)
    split = None if split_a is None else partial(S.exponential_split, a=split_a)

    def run(fn_get, fn_att, dev):
        xs, bs = xyz.to(dev), b.to(dev)
        leaves = [t.clone().to(dev).requires_grad_(True) for t in (q, k, v, tq, tk, tv)]
        i0, i0o, n_max, i1, i1o, sort_idx = fn_get(xs, bs, window, False)
        out = fn_att(leaves[0], leaves[1], leaves[2], xs, i0.int(), i0o.int(), n_max, i1.int(), i1o.int(), sort_idx, window, False,
                     pe_type='contextual', rel_query=True, rel_key=True, rel_value=True, quant_size=quant, quant_grid_length=qgl,
                     relative_pos_query_table=leaves[3], relative_pos_key_table=leaves[4], relative_pos_value_table=leaves[5],
                     split_func=split)
        out.backward(go.to(dev))
        return [out.detach().cpu()] + [t.grad.cpu() for t in leaves]
    ten = run(lay.get_indices_params, lay.sparse_self_attention, 'cuda')
    fused = run(sptr.get_indices_params, sptr.sparse_self_attention, 'cuda')
    cpu = P.layer(P.CpuBackend())
    ora = run(cpu.get_indices_params, cpu.sparse_self_attention, 'cpu')
    names = ['out', 'dq', 'dk', 'dv', 'dTq', 'dTk', 'dTv']
    for n_, a, f, o in zip(names, ten, fused, ora):
        ra, rf = _rel(a, o), _rel(f, o)
        print('REFSEQ sptr %s %s: ten-entry layer vs oracle %.2e, fused vs oracle %.2e, ten vs fused %.2e' % ('sphere' if spherical else 'cubic', n_, ra, rf, _rel(a, f)))
        assert ra < 2e-5 and rf < 2e-5 and _rel(a, f) < 2e-5, (n_, ra, rf)


def test_reference_spformer_sequence_over_the_ten_sptr_cuda_entries(hip):
    """The whole SPVCNN_SPFORMER in the reference's call pattern with sptr's OWN Python layer (M-sized pair arrays, CSR softmax)
    on top of the ten C-ABI entries -- what a maintainer gets who only swaps the two extension modules."""
    import types
    import u2mkd_amd.torchsparse as ts
    from u2mkd_amd.sptr import sptr_cuda
    lay = P.layer(sptr_cuda)
    pkg = types.ModuleType('sptr_layer_over_u2mkd')
    pkg.to_3d_numpy, pkg.get_indices_params, pkg.sparse_self_attention = lay.to_3d_numpy, lay.get_indices_params, lay.sparse_self_attention
    RG = R.over(ts, pkg)
    RG.sptr = pkg                # (R.over caches one copy per torchsparse package: bind this attention layer)
    try:
        gold = np.load(os.path.join(G, 'spformer_cr10_4000.npz'))
        b = synth_batch(2000, 2, seed=33)
        feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
        m = RG.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))
        O.fill_state_by_name(m)
        m.cuda().train()
        m.dropout.p = 0.0
        out = m({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
        err = float((out.detach().cpu() - torch.from_numpy(gold['logits'])).abs().max())
        print('REFSEQ spformer over the ten entries max |logit - golden| %.2e' % err)
        assert err < 1e-3, err
        loss = O.mix_lovasz_cross_entropy(out, labels)
        assert abs(float(loss.detach()) - float(gold['loss'])) < 1e-3
        loss.backward()
        g = dict(m.named_parameters())
        blk = 'transformer_blocks.1.attn.'
        for name, key in ((blk + 'relative_pos_query_table', 'grad_tq'), (blk + 'relative_pos_value_table_sphere', 'grad_tv_sphere'),
                          (blk + 'qkv.weight', 'grad_qkv')):
            rel = _rel(g[name].grad, gold[key])
            print('REFSEQ spformer/ten grad', name, '%.2e' % rel)
            assert rel < 1.2e-2, (name, rel)
    finally:
        import u2mkd_amd.sptr as sptr
        RG.sptr = sptr
