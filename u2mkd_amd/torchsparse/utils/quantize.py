"""torchsparse.utils.quantize.sparse_quantize -- dataset-side numpy
(core/datasets/lc_semantic_nusc_tsd_full.py:214,421): floor to the voxel grid,
ravel-hash, keep the first point of every voxel."""
import numpy as np

__all__ = ['sparse_quantize', 'ravel_hash']


def ravel_hash(x: np.ndarray) -> np.ndarray:
    assert x.ndim == 2, x.shape
    x = x - np.min(x, axis=0)
    x = x.astype(np.uint64, copy=False)
    xmax = np.max(x, axis=0).astype(np.uint64) + 1
    h = np.zeros(x.shape[0], dtype=np.uint64)
    for k in range(x.shape[1] - 1):
        h += x[:, k]
        h *= xmax[k + 1]
    h += x[:, -1]
    return h


def sparse_quantize(coords, voxel_size=1, *, return_index=False, return_inverse=False):
    if isinstance(voxel_size, (float, int)):
        voxel_size = tuple(voxel_size for _ in range(3))
    voxel_size = np.array(voxel_size)
    coords = np.floor(coords / voxel_size).astype(np.int32)
    _, indices, inverse_indices = np.unique(ravel_hash(coords), return_index=True, return_inverse=True)
    coords = coords[indices]
    outputs = [coords]
    if return_index:
        outputs += [indices]
    if return_inverse:
        outputs += [inverse_indices]
    return outputs[0] if len(outputs) == 1 else outputs
