"""hipGraph replay of the static-shape (camera side) pieces of the KD step (u2mkd_amd/graphs.py): a replayed piece
returns what the eager piece returns -- outputs, input / parameter gradients, BatchNorm buffers -- including under
gradient accumulation, and the whole KD step takes the same losses with and without capture."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(8, 16, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(16)
        self.conv2 = nn.Conv2d(16, 8, 1)

    def forward(self, x, y):
        h = torch.relu(self.bn(self.conv(x)))
        return self.conv2(h) + y, h


def _pair():
    from u2mkd_amd import graphs
    torch.manual_seed(3)
    eager = _Block().cuda()
    replay = copy.deepcopy(eager)
    piece = graphs.StaticPiece('block', lambda x, y: replay(x, y), [replay])
    return eager, replay, piece


def _inputs(seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(4, 8, 24, 40, device='cuda', generator=g)                    # no gradient (an image)
    y = torch.randn(4, 8, 24, 40, device='cuda', generator=g).requires_grad_()     # an activation from upstream
    return x, y


def _loss(outs):
    a, h = outs
    return (a * a).mean() + h.sum() * 1e-3


def _same(a, b, what):
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (what, float((a - b).abs().max()))


def test_replayed_piece_equals_eager_piece_over_steps(hip, monkeypatch):
    from u2mkd_amd import graphs
    monkeypatch.setattr(graphs, '_ENABLED', True)
    eager, replay, piece = _pair()
    opt_e = torch.optim.SGD(eager.parameters(), lr=0.05, momentum=0.9)
    opt_r = torch.optim.SGD(replay.parameters(), lr=0.05, momentum=0.9)
    for step in range(4):
        (xe, ye), (xr, yr) = _inputs(step), _inputs(step)
        le, lr_ = _loss(eager(xe, ye)), _loss(piece(xr, yr))
        opt_e.zero_grad(); opt_r.zero_grad()                 # the reference's order: forward, zero_grad, backward
        le.backward(); lr_.backward()
        _same(le, lr_, 'loss step %d' % step)
        _same(ye.grad, yr.grad, 'input gradient')
        for (n, pe), pr in zip(eager.named_parameters(), replay.parameters()):
            _same(pe.grad, pr.grad, n)
        opt_e.step(); opt_r.step()
    rec = list(piece._records.values())
    assert len(rec) == 1 and rec[0] is not None, 'the piece was not captured'
    for (n, be), br in zip(eager.named_buffers(), replay.buffers()):            # warm-up passes left no trace
        _same(be.float(), br.float(), n)
    assert int(replay.bn.num_batches_tracked) == 4


@pytest.mark.parametrize('set_to_none', [True, False])
def test_replayed_piece_accumulates_gradients_like_eager(hip, monkeypatch, set_to_none):
    """Two backward passes into the same .grad, then zero_grad(set_to_none=...) and two more: the static gradient
    buffers of the replay never stand in for an accumulated gradient."""
    from u2mkd_amd import graphs
    monkeypatch.setattr(graphs, '_ENABLED', True)
    eager, replay, piece = _pair()
    for rnd in range(2):
        for k in range(2):
            (xe, ye), (xr, yr) = _inputs(10 * rnd + k), _inputs(10 * rnd + k)
            _loss(eager(xe, ye)).backward()
            _loss(piece(xr, yr)).backward()
        for (n, pe), pr in zip(eager.named_parameters(), replay.parameters()):
            _same(pe.grad, pr.grad, '%s round %d' % (n, rnd))
        eager.zero_grad(set_to_none=set_to_none)
        replay.zero_grad(set_to_none=set_to_none)
    assert all(r is not None for r in piece._records.values())


def test_piece_runs_eagerly_when_it_does_not_qualify(hip, monkeypatch):
    from u2mkd_amd import graphs
    monkeypatch.setattr(graphs, '_ENABLED', True)
    eager, replay, piece = _pair()
    x, y = _inputs(0)
    with torch.no_grad():
        piece(x, y)
    replay.eval()
    piece(x, y)
    replay.train()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        piece(x, y)
    assert not piece._records
    monkeypatch.setattr(graphs, '_ENABLED', False)
    piece(x, y)
    assert not piece._records


def test_kd_step_with_captured_camera_side_equals_eager_step(hip, monkeypatch):
    from u2mkd_amd import graphs, train as T
    from u2mkd_amd.synth import synth_kd_batch
    from test_gpu_configs import _runner
    d = T.kd_batch_to_device(synth_kd_batch(2500, 1, seed=5, image_hw=(64, 112)))
    losses = {}
    for mode in (False, True):
        monkeypatch.setattr(graphs, '_ENABLED', mode)
        run = _runner(1.0, 1.0)
        losses[mode] = [float(run(d)) for _ in range(3)]
        if mode:
            pieces = run.model.model_s._pieces
            assert sorted(pieces) == sorted(['head', 'stage1', 'stage2', 'stage3', 'l2c0', 'l2c1', 'l2c2', 'l2c3', 'decoder_low'])
            for name, p in pieces.items():
                assert p._records and all(r is not None for r in p._records.values()), name
            for n, p in run.model.model_s.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        del run
        torch.cuda.empty_cache()
    # the first step is exactly comparable (same weights); later ones carry two lr-0.24 updates whose atomics-order
    # noise the tiny scene amplifies
    assert np.isclose(losses[True][0], losses[False][0], rtol=2e-3), losses
    assert np.allclose(losses[True], losses[False], rtol=5e-2), losses


def test_two_forwards_before_one_backward_keep_both_sets_of_activations(hip, monkeypatch):
    """a second forward of a captured piece before the first one's backward must not overwrite the first
    one's saved activations / outputs (two student passes summed into one loss).  The second call runs eagerly; a
    forward whose graph is dropped without a backward does not block later replays."""
    from u2mkd_amd import graphs
    monkeypatch.setattr(graphs, '_ENABLED', True)
    eager, replay, piece = _pair()
    (x0, y0), (x1, y1) = _inputs(0), _inputs(1)
    _loss(piece(x0, y0)).backward()                      # capture + one ordinary step
    replay.zero_grad()
    (xe0, ye0), (xe1, ye1) = _inputs(0), _inputs(1)
    (xr0, yr0), (xr1, yr1) = _inputs(0), _inputs(1)
    eager.load_state_dict(replay.state_dict())
    out_a = piece(xr0, yr0)
    kept = out_a[0].clone()
    out_b = piece(xr1, yr1)                              # before out_a's backward
    _same(out_a[0], kept, 'first output overwritten by the second forward')
    (_loss(out_a) + _loss(out_b)).backward()
    (_loss(eager(xe0, ye0)) + _loss(eager(xe1, ye1))).backward()
    _same(ye0.grad, yr0.grad, 'input gradient of the first forward')
    _same(ye1.grad, yr1.grad, 'input gradient of the second forward')
    for (n, pe), pr in zip(eager.named_parameters(), replay.parameters()):
        _same(pe.grad, pr.grad, n)
    # a forward that never gets a backward: once its graph is gone the piece replays again
    rec = next(iter(piece._records.values()))
    out_c = piece(xr0, yr0)
    assert rec.pending is not None
    del out_c
    assert rec.pending() is None
    out_d = piece(xr0, yr0)
    assert rec.pending is not None and rec.pending() is not None      # replayed (an eager call leaves `pending` alone)
    _loss(out_d).backward()
    assert rec.pending is None
