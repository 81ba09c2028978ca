"""Loader of ``lib/_u2mkd_host.so`` (csrc_host/host_ops.cpp): the hottest operators' host side -- allocation, the C-ABI call,
the autograd node -- in C++ instead of Python Functions, because the training step is bound by the host.  Same kernels, same C
ABI, same results; ``U2MKD_HOST_OPS=0`` keeps every operator on its Python Function (the formulation the tests compare with).
Like ``_lib``: a missing library raises (build it with ``python -m u2mkd_amd.build``)."""
import importlib.util
import os

import torch  # noqa: F401  (libtorch has to be loaded before the extension)

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, 'lib', '_u2mkd_host.so')
ENABLED = os.environ.get('U2MKD_HOST_OPS', '1') != '0'
_mod = None


def load():
    global _mod
    if _mod is None:
        _lib.load()                       # (libu2mkd_hip.so first: the extension links against it)
        if not os.path.exists(PATH):
            raise RuntimeError(f'{PATH} is missing: build it with `python -m u2mkd_amd.build` (or run with U2MKD_HOST_OPS=0)')
        spec = importlib.util.spec_from_file_location('_u2mkd_host', PATH)
        _mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mod)
    return _mod


def ops():
    """The extension module, or None when the Python Functions are asked for (U2MKD_HOST_OPS=0)."""
    return load() if ENABLED else None
