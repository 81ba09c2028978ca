"""Weight gradients that are joined at the end of the backward (u2mkd_amd/deferred.py): the gradients a caller reads after
``backward()`` equal those of the inline formulation, also when a parameter already holds a gradient (accumulation) and when a
hook reads ``.grad`` during the backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from u2mkd_amd import _lib, deferred
    _lib.load()
    was = deferred.enabled()
    deferred.enable()              # (the package's trainers switch it on; the bare operators leave it off)
    yield _lib
    deferred.enable(was)


def test_off_by_default_for_bare_operators():
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from u2mkd_amd import deferred; print(deferred.enabled())' % root],
                         capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == 'False', out.stderr[-500:]


def _spvcnn_grads(overlap, steps=2):
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    from u2mkd_amd.synth import synth_batch
    from u2mkd_amd.torchsparse.nn import functional as F
    old = F._OVERLAP_WGRAD
    F._OVERLAP_WGRAD = overlap
    try:
        torch.manual_seed(3)
        model = lidar.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
        crit = MixLovaszCrossEntropy(ignore_index=0)
        b = synth_batch(6000, 1, seed=5)
        feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
        out = []
        for _ in range(steps):          # the second pass ACCUMULATES into existing gradients (the inline path of _wgrad_side)
            loss = crit(model({'lidar': ts.SparseTensor(feats, coords)})['x_vox'], labels)
            loss.backward()
            out.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        return out
    finally:
        F._OVERLAP_WGRAD = old


def test_deferred_weight_gradients_equal_the_inline_ones(hip):
    a = _spvcnn_grads(True)
    b = _spvcnn_grads(False)
    assert a[0].keys() == b[0].keys() and len(a[0]) > 50
    for step in range(2):
        for n in a[step]:
            assert torch.equal(a[step][n], b[step][n]), (step, n)


def test_a_hook_that_joins_reads_finished_gradients(hip):
    """distributed.BucketedGradientAverage's protocol: a post-accumulate hook calls deferred.join() before it reads .grad."""
    from u2mkd_amd import deferred, lidar, torchsparse as ts
    from u2mkd_amd.synth import synth_batch
    torch.manual_seed(4)
    model = lidar.SPVCNN(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
    b = synth_batch(6000, 1, seed=6)
    feats, coords = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords'))
    seen = {}

    def hook(p):
        deferred.join()
        seen[id(p)] = p.grad.clone()
    handles = [p.register_post_accumulate_grad_hook(hook) for p in model.parameters()]
    model({'lidar': ts.SparseTensor(feats, coords)})['x_vox'].square().mean().backward()
    torch.cuda.synchronize()
    for h in handles:
        h.remove()
    n = 0
    for p in model.parameters():
        if p.grad is not None:
            assert torch.equal(seen[id(p)], p.grad)
            n += 1
    assert n > 50


def test_camera_conv2d_equals_nn_conv2d(hip):
    from u2mkd_amd.camera import Conv2d
    torch.manual_seed(0)
    for cin, cout, k, stride in ((64, 64, 3, 1), (64, 128, 3, 2), (128, 128, 1, 1), (3, 64, 7, 1)):
        conv = Conv2d(cin, cout, k, stride, k // 2, bias=False).cuda()
        ref = torch.nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).cuda()
        ref.weight.data.copy_(conv.weight.data)
        x = torch.randn(2, cin, 45, 80, device='cuda')
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        g = torch.randn_like(ref(xb))
        ya = conv(xa)
        ya.backward(g)
        ref(xb).backward(g)
        torch.cuda.synchronize()
        assert torch.allclose(ya, ref(xb), rtol=1e-5, atol=1e-5)
        assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(conv.weight.grad, ref.weight.grad, rtol=1e-4, atol=2e-3 * float(ref.weight.grad.abs().max()))
        # a second backward accumulates: the inline path
        conv(xa).backward(g)
        ref(xb).backward(g)
        assert torch.allclose(conv.weight.grad, ref.weight.grad, rtol=1e-4, atol=2e-3 * float(ref.weight.grad.abs().max()))


def test_a_weight_used_twice_in_one_pass_accumulates_finished_gradients(hip):
    """a second contribution to a leaf in one backward pass (a shared / tied weight) is added to the first by
    autograd as soon as the second function returns -- the first must be complete on that stream and the second computed in
    line.  A SubM convolution applied twice and a camera Conv2d applied twice, against the formulation without side streams."""
    from u2mkd_amd import torchsparse as ts
    from u2mkd_amd.camera import Conv2d
    from u2mkd_amd.synth import synth_batch
    from u2mkd_amd.torchsparse import nn as spnn
    from u2mkd_amd.torchsparse.nn import functional as F
    b = synth_batch(20000, 1, seed=9)
    coords = torch.from_numpy(b['coords']).cuda()
    torch.manual_seed(1)
    feats = torch.randn(coords.shape[0], 64, device='cuda')
    conv = spnn.Conv3d(64, 64, 3).cuda()
    cam = Conv2d(32, 32, 3, 1, 1, bias=False).cuda()
    img = torch.randn(2, 32, 90, 160, device='cuda')

    def run(overlap):
        old = F._OVERLAP_WGRAD
        F._OVERLAP_WGRAD = overlap
        try:
            for m in (conv, cam):
                m.zero_grad(set_to_none=True)
            x = ts.SparseTensor(feats, coords)
            y = conv(conv(x))                     # the same kernel twice
            z = cam(torch.relu(cam(img)))         # the same filter twice
            (y.F.square().mean() + z.square().mean()).backward()
            torch.cuda.synchronize()
            return conv.kernel.grad.clone(), cam.weight.grad.clone()
        finally:
            F._OVERLAP_WGRAD = old
    a = [run(True) for _ in range(3)]
    from u2mkd_amd import deferred
    with deferred.scope(False):
        ref = run(False)
    for got in a:
        assert torch.equal(got[0], ref[0])
        assert torch.allclose(got[1], ref[1], rtol=1e-4, atol=2e-3 * float(ref[1].abs().max()))     # (MIOpen: not bit-reproducible)


def test_a_backward_pass_that_raises_does_not_disarm_the_next_one(hip):
    """the engine's final callbacks do not run when a backward raises; the next pass books its own join (per graph
    task) and first waits for what the failed pass left on the side streams."""
    from u2mkd_amd import deferred, torchsparse as ts
    from u2mkd_amd.synth import synth_batch
    from u2mkd_amd.torchsparse import nn as spnn
    b = synth_batch(20000, 1, seed=10)
    coords = torch.from_numpy(b['coords']).cuda()
    torch.manual_seed(2)
    feats = torch.randn(coords.shape[0], 64, device='cuda', requires_grad=True)
    c1, c2 = spnn.Conv3d(64, 64, 3).cuda(), spnn.Conv3d(64, 64, 3).cuda()

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise ValueError('boom')

    def loss(fail):
        h = c1(ts.SparseTensor(feats, coords))
        if fail:
            h.F = Boom.apply(h.F)                 # c2's weight gradient is issued (deferred) before this node raises
        return c2(h).F.square().mean()
    with pytest.raises(ValueError, match='boom'):
        loss(True).backward()
    assert deferred.pending()                     # the failed pass left booked work behind
    for m in (c1, c2):
        m.zero_grad(set_to_none=True)
    loss(False).backward()
    assert not deferred.pending()                 # this pass's own end-of-backward join ran
    got = (c1.kernel.grad.clone(), c2.kernel.grad.clone())
    torch.cuda.synchronize()
    for m in (c1, c2):
        m.zero_grad(set_to_none=True)
    with deferred.scope(False):
        loss(False).backward()
    torch.cuda.synchronize()
    assert torch.equal(got[0], c1.kernel.grad) and torch.equal(got[1], c2.kernel.grad)


def test_linear_bias_gradient_rides_the_side_stream_with_the_weight_gradient(hip):
    """nn.Linear's bias gradient (a column sum of dY, or the exact zero of a bias that feeds a train-mode BatchNorm) is issued
    behind the weight gradient on the side stream and joined with it at the end of the backward: same values as torch's, also on
    a second pass that accumulates into existing .grad (computed in line then) and for a bias used by two layers."""
    from u2mkd_amd import deferred
    from u2mkd_amd.torchsparse.nn import functional as F
    torch.manual_seed(4)
    n, cin, cout = 20000, 64, 96
    x = torch.randn(n, cin, device='cuda')
    w = torch.nn.Parameter(torch.randn(cout, cin, device='cuda') * 0.1)
    b = torch.nn.Parameter(torch.randn(cout, device='cuda'))
    g = torch.randn(n, cout, device='cuda')
    launched = []
    real_sum = torch.sum

    def spy(*a, **k):
        launched.append(torch.cuda.current_stream().cuda_stream)
        return real_sum(*a, **k)
    main = torch.cuda.current_stream().cuda_stream
    torch.sum = spy
    try:
        with deferred.scope():
            y = F.linear(x, w, b)
            y.backward(g)
    finally:
        torch.sum = real_sum
    torch.cuda.synchronize()
    assert launched and all(s != main for s in launched), 'the column sum ran on the backward\'s own stream'
    want_b = g.double().sum(0)
    assert float((b.grad.double() - want_b).abs().max()) <= 1e-4 * float(want_b.abs().max())
    want_w = g.double().t() @ x.double()
    assert float((w.grad.double() - want_w).abs().max()) <= 1e-4 * float(want_w.abs().max())
    # second pass: .grad exists -> autograd accumulates when the function returns -> both gradients in line, sums doubled
    with deferred.scope():
        F.linear(x, w, b).backward(g)
    torch.cuda.synchronize()
    assert float((b.grad.double() - 2 * want_b).abs().max()) <= 2e-4 * float(want_b.abs().max())
    # the exact zero of a bias in front of a train-mode BatchNorm
    b2 = torch.nn.Parameter(torch.randn(cout, device='cuda'))
    w.grad = None
    with deferred.scope():
        F.linear(x, w, b2, bias_feeds_batchnorm=True).backward(g)
    torch.cuda.synchronize()
    assert b2.grad is not None and float(b2.grad.abs().max()) == 0.0


def _spformer_grads(defer_tables, steps=2):
    from u2mkd_amd import lidar, torchsparse as ts
    from u2mkd_amd.losses import MixLovaszCrossEntropy
    from u2mkd_amd.synth import synth_batch
    import u2mkd_amd.sptr.functional as SF
    old = SF._DEFER_TABLES
    SF._DEFER_TABLES = defer_tables
    launched = []
    real = SF.L.call

    def spy(name, *a):
        if name == 'u2mkd_sptr_table_reduce':
            launched.append(a[-1] != torch.cuda.current_stream().cuda_stream)
        return real(name, *a)
    SF.L.call = spy
    try:
        torch.manual_seed(3)
        model = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=0.5, drop_path_rate=0.0)).cuda().train()
        crit = MixLovaszCrossEntropy(ignore_index=0)
        b = synth_batch(6000, 1, seed=5)
        feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
        out = []
        for _ in range(steps):          # the second pass accumulates into existing gradients: the sums run in line again
            loss = crit(model({'lidar': ts.SparseTensor(feats, coords)})['x_vox'], labels)
            loss.backward()
            torch.cuda.synchronize()
            out.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        return out, launched
    finally:
        SF.L.call = real
        SF._DEFER_TABLES = old


def test_sptr_table_gradients_summed_on_the_side_stream_equal_the_inline_ones(hip):
    """The relative-position tables are leaf parameters: the slab sum that finishes their gradients is queued on the
    weight-gradient side stream (u2mkd_sptr_table_reduce) and joined at the end of the backward -- the same bits as the sum
    inside u2mkd_sptr_attention_backward_strided, in the first pass (deferred) and in the second (in line: .grad exists)."""
    a, launched = _spformer_grads(True)
    b, none = _spformer_grads(False)
    assert launched and all(launched), 'no table sum left the backward\'s stream'
    assert not none
    tables = [n for n in a[0] if 'table' in n]
    assert len(tables) >= 6, sorted(a[0])[:20]
    for step in range(2):
        assert a[step].keys() == b[step].keys()
        for n in a[step]:
            assert torch.equal(a[step][n], b[step][n]), (step, n)
