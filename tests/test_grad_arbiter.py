"""Host-side check of the fp64 gradient arbiter's ReLU bookkeeping (tests/grad_arbiter.py) on the CPU oracle."""
import torch

import grad_arbiter as GA
from oracle import spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd.synth import synth_batch


def test_flipped_relu_elements_are_found_and_named():
    b = synth_batch(500, 1, 5)
    feats, coords = torch.from_numpy(b['feats']), torch.from_numpy(b['coords'])
    kw = dict(cr=0.25, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    recs = []
    for dtype in (torch.float64, torch.float32):
        m = O.fill_state_by_name(O.SPVCNN(**kw)).train().to(dtype)
        m.dropout.p = 0.0
        rec, remove = GA.record_oracle_relus(m)
        m({'lidar': ots.SparseTensor(feats.to(dtype), coords)})
        remove()
        recs.append(rec)
    rec64, rec32 = recs
    # every ReLU of the network is on record: 2 stem + 4 down x (1 + 2 x 2) + 4 up x (1 + 2 x 2) + 3 point MLPs
    assert len(rec64) == 2 + 4 * (1 + 2 * 2) + 4 * (1 + 2 * 2) + 3 and set(rec64) == set(rec32)
    assert 'vox_downs.1.1.relu' in rec64 and 'stem.2' in rec64 and 'point_transforms.0.2' in rec64
    masks = {k: v > 0 for k, v in rec32.items()}
    flips, n = GA.flipped_relu_elements(rec64, masks)
    assert all(abs(f[3]) < 1e-4 for f in flips)          # fp32 vs fp64: only rounding-sized pre-activations can differ
    x = rec64['vox_downs.2.1.relu']
    r, c = [int(v) for v in (x.abs() > 0.5).nonzero()[0]]
    masks['vox_downs.2.1.relu'][r, c] = not bool(masks['vox_downs.2.1.relu'][r, c])
    flips2, n2 = GA.flipped_relu_elements(rec64, masks, limit=10 ** 6)
    assert n2 == n + 1 and ('vox_downs.2.1.relu', r, c, float(x[r, c])) in flips2


def test_forced_masks_reproduce_a_run_and_move_with_a_flip():
    """force_relu_masks: the oracle evaluated with ITS OWN recorded masks gives the same logits and gradients as the free
    run (y = x * (x > 0) is what ReLU computes); with one decided element flipped, the gradients move -- the causal
    escape of assert_grads_within_fp64_gate compares against exactly this re-evaluation."""
    b = synth_batch(400, 1, 6)
    feats, coords = torch.from_numpy(b['feats']).double(), torch.from_numpy(b['coords'])
    kw = dict(cr=0.25, in_channel=4, num_classes=17, pres=0.05, vres=0.05)

    def run(masks=None):
        m = O.fill_state_by_name(O.SPVCNN(**kw)).train().double()
        m.dropout.p = 0.0
        if masks is not None:
            GA.force_relu_masks(m, masks)
        rec, remove = GA.record_oracle_relus(m)
        out = m({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
        remove()
        (out ** 2).mean().backward()
        return m, out.detach(), rec
    m0, out0, rec0 = run()
    masks = {k: v > 0 for k, v in rec0.items()}
    m1, out1, _ = run(masks)
    assert float((out0 - out1).abs().max()) < 1e-12
    g0, g1 = dict(m0.named_parameters()), dict(m1.named_parameters())
    for k in g0:
        assert float((g0[k].grad - g1[k].grad).abs().max()) <= 1e-12 * (1 + float(g0[k].grad.abs().max())), k
    x = rec0['vox_downs.1.1.relu']
    r, c = [int(v) for v in (x > 0.3).nonzero()[0]]
    masks['vox_downs.1.1.relu'][r, c] = False
    m2, out2, _ = run(masks)
    assert float((out0 - out2).abs().max()) > 1e-6
    g2 = dict(m2.named_parameters())
    assert float((g0['stem.0.kernel'].grad - g2['stem.0.kernel'].grad).abs().max()) > 0
