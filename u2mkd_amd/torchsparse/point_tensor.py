"""torchsparse.point_tensor (v1.4.0 module path of PointTensor)."""
from .tensor import PointTensor

__all__ = ['PointTensor']
