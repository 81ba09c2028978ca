"""Row f2 / a19: LR schedule pinned against the reference's function, optimizer settings, and the trainer
checkpoint (save / resume / the three weight sources of `_before_train`) with the reference's keys.  CPU only."""
import json
import os

import pytest
import torch

from u2mkd_amd import train as T

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_cosine_schedule_equals_reference_values():
    """tests/golden/cosine_schedule.json = core/schedulers.py:10-35 evaluated in the build container
    (make_scheduler_golden.py): warm-up of 1000 // world steps only for world > 1, batch scaled by the world."""
    rows = json.load(open(os.path.join(G, 'cosine_schedule.json')))
    assert len(rows) >= 150
    for r in rows:
        got = T.cosine_schedule_with_warmup(r['k'], r['num_epochs'], r['batch_size'], r['dataset_size'], r['world'])
        assert got == pytest.approx(r['value'], rel=0, abs=1e-15), r


def test_optimizer_is_the_reference_sgd():
    """core/builder.py:663-669: SGD(lr 0.24, momentum 0.9, nesterov, weight_decay 1e-4)."""
    opt = T.make_optimizer([torch.nn.Parameter(torch.zeros(3))])
    g = opt.param_groups[0]
    assert (g['lr'], g['momentum'], g['weight_decay'], g['nesterov']) == (0.24, 0.9, 1.0e-4, True)


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.model_t = torch.nn.Linear(4, 3)
        self.model_s = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
        self.classifier = torch.nn.Linear(2, 2)


class _Runner:
    """The attributes state_dict / load_state_dict use, on a CPU module (LidarStep / KDStep need the HIP device)."""

    def __init__(self, wrap=False):
        self.model = _Tiny()
        self.net = torch.nn.Sequential()
        self.net = self.model if not wrap else _Wrapped(self.model)
        self.amp = T._Amp(False)
        self.opt = T.make_optimizer(self.net.parameters())
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda k: T.cosine_schedule_with_warmup(k, 2, 1, 10, 1))


class _Wrapped(torch.nn.Module):          # stands in for DistributedDataParallel: keys get a `module.` prefix
    def __init__(self, m):
        super().__init__()
        self.module = m


def _train_a_bit(r, steps=3):
    for _ in range(steps):
        loss = sum((p ** 2).sum() for p in r.net.parameters()) + r.net.state_dict()[next(iter(r.net.state_dict()))].sum()
        r.opt.zero_grad()
        loss.backward()
        r.opt.step()
        r.sched.step()


def test_optimizer_variants_split_the_sphereformer_blocks_like_the_reference():
    """core/builder.py:670-716: `sgd_spformer` / `adamw_spformer` give the parameters whose name contains
    "transformer_block" their own group at a reduced learning rate; `adam` / `adamw` take everything at once."""
    net = torch.nn.ModuleDict({'stem': torch.nn.Linear(4, 4), 'transformer_blocks': torch.nn.ModuleList([torch.nn.Linear(4, 4)])})
    net['stem'].bias.requires_grad_(False)
    opt = T.make_optimizer(net, name='sgd_spformer')
    assert isinstance(opt, torch.optim.SGD) and len(opt.param_groups) == 2
    rest, blocks = opt.param_groups
    assert [id(p) for p in rest['params']] == [id(net['stem'].weight)]                     # frozen bias left out
    assert [id(p) for p in blocks['params']] == [id(p) for p in net['transformer_blocks'].parameters()]
    assert rest['lr'] == 0.24 and abs(blocks['lr'] - 0.024) < 1e-12 and all(g['nesterov'] and g['momentum'] == 0.9 for g in opt.param_groups)
    opt = T.make_optimizer(net, name='adamw_spformer', lr=1e-3, weight_decay=0.05, transformer_lr_scale=0.5)
    assert isinstance(opt, torch.optim.AdamW) and [g['lr'] for g in opt.param_groups] == [1e-3, 5e-4]
    assert isinstance(T.make_optimizer(net.parameters(), name='adam', lr=1e-3), torch.optim.Adam)
    assert isinstance(T.make_optimizer(net, name='adamw', lr=1e-3), torch.optim.AdamW)
    import pytest
    with pytest.raises(NotImplementedError):
        T.make_optimizer(net.parameters(), name='lion')
    with pytest.raises(TypeError):
        T.make_optimizer(net.parameters(), name='sgd_spformer')


def test_checkpoint_roundtrip_and_ddp_prefix(tmp_path):
    a = _Runner(wrap=True)
    _train_a_bit(a)
    ck = T.state_dict(a)
    assert sorted(ck) == ['model', 'optimizer', 'scaler', 'scheduler']          # core/nusc_trainers.py:423-429
    assert all(k.startswith('module.') for k in ck['model'])
    path = os.path.join(tmp_path, 'step-3.pt')
    torch.save(ck, path)
    # resume into an unwrapped runner: `module.` is stripped like core/nusc_trainers.py:180
    b = _Runner(wrap=False)
    T.load_state_dict(b, torch.load(path, weights_only=False))
    for (ka, va), (kb, vb) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert b.sched.last_epoch == 3 and b.opt.param_groups[0]['lr'] == a.opt.param_groups[0]['lr']
    assert b.opt.state_dict()['state'].keys() == a.opt.state_dict()['state'].keys()
    # the three weight sources of _before_train, in the reference's order of precedence
    m = _Tiny()
    assert T.load_weights(m, weight_path=path) == 'weight_path'
    assert torch.equal(m.model_s[0].weight, a.model.model_s[0].weight)
    # pretrain_weight: keys are taken AS SAVED (the reference does not strip `module.` here,
    # core/nusc_trainers.py:186-190), classifier heads are skipped, non-strict
    plain = os.path.join(tmp_path, 'plain.pt')
    torch.save({'model': a.model.state_dict()}, plain)
    m = _Tiny()
    w0 = m.classifier.weight.clone()
    assert T.load_weights(m, weight_path=os.path.join(tmp_path, 'nope.pt'), pretrain_weight=plain) == 'pretrain_weight'
    assert torch.equal(m.classifier.weight, w0) and torch.equal(m.model_t.weight, a.model.model_t.weight)
    teacher = os.path.join(tmp_path, 'teacher.pt')
    torch.save({'model': {'module.' + k: v for k, v in a.model.model_t.state_dict().items()}}, teacher)
    m = _Tiny()
    assert T.load_weights(m, teacher_pretrain_weight=teacher) == 'teacher_pretrain_weight'
    assert torch.equal(m.model_t.bias, a.model.model_t.bias)
    assert T.load_weights(_Tiny()) is None
