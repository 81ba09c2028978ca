"""The drop-in boundary, host side: every torchsparse v1.4.0 / sptr name the reference imports or calls
(grep of /root/reference/core and train*.py at fixture time, listed below with one call site each)
resolves after ``install_as_torchsparse()`` -- no GPU and no HIP library needed to import."""
import importlib

import pytest

# name -> one reference call site
SURFACE = {
    'torchsparse': {
        'SparseTensor': 'core/models/semantickitti/spvcnn.py:4', 'PointTensor': 'core/models/semantickitti/spvcnn.py:5',
        'cat': 'core/models/semantickitti/spvcnn.py:142 (torchsparse.cat([y1, x3]))'},
    'torchsparse.nn': {
        'Conv3d': 'core/models/build_blocks.py:25', 'BatchNorm': 'core/models/build_blocks.py:30',
        'ReLU': 'core/models/build_blocks.py:31'},
    'torchsparse.nn.functional': {
        'sphash': 'core/models/utils.py:19', 'sphashquery': 'core/models/utils.py:21', 'spcount': 'core/models/utils.py:22',
        'spvoxelize': 'core/models/utils.py:24', 'spdevoxelize': 'core/models/utils.py:99',
        'calc_ti_weights': 'core/models/utils.py:94'},
    'torchsparse.nn.utils': {'get_kernel_offsets': 'core/models/utils.py:5', 'fapply': 'core/models/utils.py:143'},
    'torchsparse.utils': {'make_ntuple': 'core/models/utils.py:7'},
    'torchsparse.utils.quantize': {'sparse_quantize': 'core/datasets/semantic_nusc.py'},
    'torchsparse.utils.collate': {'sparse_collate_fn': 'core/datasets/semantic_nusc.py',
                                  'sparse_collate': 'core/datasets/lc_semantic_nusc_tsd_full.py'},
    'torchsparse.point_tensor': {'PointTensor': 'core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py:3'},
}


def test_torchsparse_names_resolve():
    import u2mkd_amd
    u2mkd_amd.install_as_torchsparse()
    for mod, names in SURFACE.items():
        m = importlib.import_module(mod)
        for name, site in names.items():
            assert hasattr(m, name), f'{mod}.{name} missing (reference uses it at {site})'
    import torchsparse
    assert torchsparse.__version__.startswith('1.4')


def test_conv3d_module_signature_and_parameter_layout():
    """spnn.Conv3d(inc, outc, kernel_size, stride, dilation, bias, transposed) with a `kernel`
    parameter [K, Cin, Cout] ([Cin, Cout] for K = 1): reference checkpoints load unchanged."""
    import u2mkd_amd
    u2mkd_amd.install_as_torchsparse()
    import torchsparse.nn as spnn
    c = spnn.Conv3d(16, 32, kernel_size=3, stride=1, dilation=1)
    assert tuple(c.kernel.shape) == (27, 16, 32) and c.bias is None
    c = spnn.Conv3d(16, 32, kernel_size=2, stride=2, transposed=True)
    assert tuple(c.kernel.shape) == (8, 16, 32)
    c = spnn.Conv3d(16, 32, kernel_size=1)
    assert tuple(c.kernel.shape) == (16, 32)
    bn = spnn.BatchNorm(32)
    assert set(dict(bn.named_parameters())) == {'weight', 'bias'}


def test_sptr_names_resolve():
    from u2mkd_amd import sptr
    for name in ('get_indices_params', 'sparse_self_attention', 'to_3d_numpy', 'SparseTrTensor'):
        assert hasattr(sptr, name), name      # core/models/sphereformer/spherical_transformer.py:7


def test_ops_refuse_cpu_tensors():
    """Product operators have no CPU path: calling them without the HIP device raises."""
    import torch
    import u2mkd_amd
    u2mkd_amd.install_as_torchsparse()
    import torchsparse.nn.functional as F
    with pytest.raises(RuntimeError):
        F.sphash(torch.zeros(4, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        F.linear(torch.zeros(4, 8), torch.zeros(8, 8))
