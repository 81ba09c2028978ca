"""Host-side behaviour of the pieces added around the camera branch: the fused BatchNorm2d module is nn.BatchNorm2d on
CPU tensors (same parameters, buffers and state-dict keys), bn_act serves converted SyncBatchNorm-style modules, and a
StaticPiece / PieceCache never captures (or survives a copy) off the device."""
import copy
import pickle

import torch
import torch.nn.functional as F
from torch import nn

from u2mkd_amd import camera, graphs


def test_fused_batchnorm2d_is_nn_batchnorm2d_on_the_cpu():
    torch.manual_seed(0)
    mine, ref = camera.BatchNorm2d(8, momentum=0.05), nn.BatchNorm2d(8, momentum=0.05)
    assert list(mine.state_dict()) == list(ref.state_dict())
    ref.load_state_dict(mine.state_dict())
    x, r = torch.randn(3, 8, 5, 7), torch.randn(3, 8, 5, 7)
    want = F.relu(ref(x) + r)
    got = mine(x, relu=True, residual=r)
    assert torch.equal(got, want)
    for (n, a), b in zip(mine.named_buffers(), ref.buffers()):
        assert torch.equal(a, b), n
    mine.eval(); ref.eval()
    assert torch.equal(mine(x), ref(x))
    assert torch.equal(camera.bn_act(ref, x, relu=True), F.relu(ref(x)))       # any BatchNorm flavour


def test_swiftnet_state_dict_keys_do_not_change_with_the_fused_modules():
    m = camera.SwiftNetRes18()
    keys = list(m.state_dict())
    assert 'bn1.running_mean' in keys and 'layer2.0.downsample.1.num_batches_tracked' in keys
    assert 'upsample.0.bottleneck.norm.weight' in keys and 'spp.spp.spp_bn.norm.bias' in keys
    y = m(torch.randn(1, 3, 32, 64), im_size=(32, 64))
    assert y.shape == (1, 128, 32, 64)


def test_static_piece_runs_eagerly_off_the_device_and_caches_do_not_travel():
    lin = nn.Linear(4, 4)
    piece = graphs.StaticPiece('lin', lambda x: lin(x), [lin])
    x = torch.randn(2, 4, requires_grad=True)
    piece(x).sum().backward()
    assert not piece._records and lin.weight.grad is not None
    cache = graphs.PieceCache(lin=piece)
    assert len(copy.deepcopy(cache)) == 0 and len(pickle.loads(pickle.dumps(cache))) == 0
    holder = nn.Module()
    holder.__dict__['_pieces'] = cache
    assert len(copy.deepcopy(holder).__dict__['_pieces']) == 0
