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
