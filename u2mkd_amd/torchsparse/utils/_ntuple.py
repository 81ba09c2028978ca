import numpy as np


def make_ntuple(x, ndim=3):
    """torchsparse.utils.make_ntuple."""
    if isinstance(x, (int, np.integer)):
        return tuple(int(x) for _ in range(ndim))
    x = tuple(int(v) for v in x)
    assert len(x) == ndim, x
    return x
