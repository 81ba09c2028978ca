"""u2mkd_amd: MI355X-native (gfx950) implementation of the U2MKD training hot path.

``u2mkd_amd.torchsparse`` is a drop-in for the torchsparse v1.4.0 Python API
the reference's ``core/models`` code calls; every operator
runs as a hand-written HIP kernel behind the C ABI in ``include/u2mkd_hip.h``.
There is no CPU / PyTorch fallback: operators raise on CPU tensors.
"""
__version__ = '0.1.0'

# (Importing the package changes nothing in the process environment.  The launchers -- bench.py, run_training.py -- call
# ``distributed.configure_runtime()`` before their first GPU call: 8 HIP hardware queues for a rank of a multi-rank job, the
# runtime's 4 for a single-rank process.)


def install_as_torchsparse():
    """Make ``import torchsparse`` resolve to this drop-in (what a maintainer
    of the reference does instead of installing torchsparse v1.4.0)."""
    import importlib
    import sys
    base = 'u2mkd_amd.torchsparse'
    sys.modules['torchsparse'] = importlib.import_module(base)
    for sub in ('tensor', 'point_tensor', 'operators', 'nn', 'nn.functional', 'nn.utils', 'nn.modules',
                'utils', 'utils.quantize', 'utils.collate'):
        sys.modules['torchsparse.' + sub] = importlib.import_module(base + '.' + sub)


def install_as_sptr():
    """Make ``from third_party.SparseTransformer.sptr import to_3d_numpy, SparseTrTensor,
    sparse_self_attention, get_indices_params`` (core/models/sphereformer/spherical_transformer.py:7)
    resolve to :mod:`u2mkd_amd.sptr` -- what a maintainer of the reference does instead of building
    the ``sptr_cuda`` extension (third_party/SparseTransformer/setup.py) and installing torch_scatter /
    torch_geometric / torch_cluster.  Parent packages that are not importable are registered as empty
    namespace modules; an existing ``third_party`` package is left in place and only its
    ``SparseTransformer.sptr`` entry is overridden."""
    import importlib
    import sys
    import types
    drop_in = importlib.import_module('u2mkd_amd.sptr')
    parent = None
    for name in ('third_party', 'third_party.SparseTransformer'):
        mod = sys.modules.get(name)
        if mod is None:
            try:
                mod = importlib.import_module(name)
            except Exception:
                mod = types.ModuleType(name)
                mod.__path__ = []
                sys.modules[name] = mod
        if parent is not None:
            setattr(parent, name.rsplit('.', 1)[1], mod)
        parent = mod
    sys.modules['third_party.SparseTransformer.sptr'] = drop_in
    parent.sptr = drop_in
    for sub in ('functional',):
        sys.modules['third_party.SparseTransformer.sptr.' + sub] = importlib.import_module('u2mkd_amd.sptr.' + sub)
    return drop_in


def install_as_sptr_cuda():
    """Make ``import sptr_cuda`` (third_party/SparseTransformer/sptr/functional.py:5) resolve to
    :mod:`u2mkd_amd.sptr.sptr_cuda`: the ten functions of the reference's CUDA extension (src/sptr/pointops_api.cpp:9-20)
    over the C-ABI entries ``u2mkd_sptr_*`` -- for a maintainer who keeps sptr's own Python layer instead of switching to
    :func:`install_as_sptr`'s fused attention."""
    import importlib
    import sys
    mod = importlib.import_module('u2mkd_amd.sptr.sptr_cuda')
    sys.modules['sptr_cuda'] = mod
    return mod
