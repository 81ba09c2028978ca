import numpy as np

_SEEN = {}      # (value, ndim) -> tuple: the same few strides / kernel sizes are asked for ~600 times per training step


def _cacheable(x):
    # plain ints and tuples of plain ints only: 1, 1.0 and True hash equal, and a cached int answer must not hide the
    # TypeError / truncation a float argument gets on the long way
    return type(x) is int or (type(x) is tuple and all(type(v) is int for v in x))


def make_ntuple(x, ndim=3):
    """torchsparse.utils.make_ntuple."""
    cache = _cacheable(x)
    if cache:
        hit = _SEEN.get((x, ndim))
        if hit is not None:
            return hit
    if isinstance(x, (int, np.integer)):
        out = tuple(int(x) for _ in range(ndim))
    else:
        out = tuple(int(v) for v in x)
        assert len(out) == ndim, x
    if cache and len(_SEEN) < 4096:
        _SEEN[(x, ndim)] = out
    return out
