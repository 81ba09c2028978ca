"""u2mkd_amd: MI355X-native (gfx950) implementation of the U2MKD training hot path.

``u2mkd_amd.torchsparse`` is a drop-in for the torchsparse v1.4.0 Python API
the reference's ``core/models`` code calls; every operator
runs as a hand-written HIP kernel behind the C ABI in ``include/u2mkd_hip.h``.
There is no CPU / PyTorch fallback: operators raise on CPU tensors.
"""
__version__ = '0.1.0'


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
