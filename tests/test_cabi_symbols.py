"""The C-ABI library loads and exports every symbol include/u2mkd_hip.h
declares (no compute calls: runs without a GPU)."""
import os
import re

from u2mkd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'u2mkd_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(u2mkd_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_typed():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/u2mkd_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in u2mkd_amd/_lib.py'
    for n in _lib.SIGNATURES:
        assert n in names, f'{n} bound in _lib.py but not declared in the header'


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.u2mkd_version() >= 100
    assert lib.u2mkd_hash_table_bytes(1000) == 2048 * 12
    assert lib.u2mkd_hash_table_bytes(80000) == 262144 * 12
    assert lib.u2mkd_conv_wgrad_pairs_workspace_bytes(80000, 64, 64, 27) == (1024 + 27) * 64 * 64 * 4
    assert lib.u2mkd_wgrad_plan_ints(27) == 58


def test_no_cpu_fallback():
    import pytest
    import torch
    from u2mkd_amd.torchsparse.nn import functional as F
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.sphash(torch.zeros(4, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.spvoxelize(torch.zeros(4, 4), torch.zeros(4, dtype=torch.int32), torch.ones(2, dtype=torch.int32))


def test_host_extension_loads_and_exposes_its_operators():
    """lib/_u2mkd_host.so (csrc_host/host_ops.cpp: the C++ host side of the hottest operators, built by build()) loads without a
    GPU, links against the C-ABI library and exposes what torchsparse/nn/functional.py routes to it; a missing library raises."""
    import pytest
    from u2mkd_amd import _host
    mod = _host.load()
    assert callable(mod.batch_norm_rows)
    assert 'batch_norm_rows(x' in mod.batch_norm_rows.__doc__
    real, _host.PATH, _host._mod = _host.PATH, _host.PATH + '.missing', None
    try:
        with pytest.raises(RuntimeError, match='is missing'):
            _host.load()
    finally:
        _host.PATH, _host._mod = real, mod


def test_fused_sgd_on_cpu_parameters_is_torchs_sgd():
    """optim.FusedSGD away from the HIP device: torch's own step (same class hierarchy, hooks, state layout); the fast
    zero_grad sets every gradient to None, zero_grad(set_to_none=False) keeps torch's semantics."""
    import torch
    from u2mkd_amd.optim import FusedSGD
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    kw = dict(lr=0.24, momentum=0.9, weight_decay=1e-4, nesterov=True)
    fa, fb = FusedSGD(a, **kw), torch.optim.SGD(b, **kw)
    assert isinstance(fa, torch.optim.SGD)
    for k in range(3):
        for p, q in zip(a, b):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        fa.step(); fb.step()
        for p, q in zip(a, b):
            assert torch.equal(p, q)
    assert fa.state_dict()['param_groups'] == fb.state_dict()['param_groups'] and not fa._fused_groups
    fa.zero_grad()
    assert all(p.grad is None for p in a)
    a[0].grad = torch.ones_like(a[0])
    fa.zero_grad(set_to_none=False)
    assert a[0].grad is not None and float(a[0].grad.abs().max()) == 0.0


def test_counts_mailbox_falls_back_to_a_copy_for_cpu_tensors():
    import torch
    from u2mkd_amd.torchsparse.nn import functional as spf
    h = spf.post_counts([torch.tensor(5), torch.tensor([9], dtype=torch.int32)])
    assert h.seq is None and spf.wait_counts(h) == [5, 9] and spf.wait_counts(h) == [5, 9]
    assert spf.read_counts([]) == []
