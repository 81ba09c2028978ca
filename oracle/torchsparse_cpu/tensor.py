"""SparseTensor / PointTensor of torchsparse v1.4.0."""
from ..ts_ref import make_ntuple

__all__ = ['SparseTensor', 'PointTensor']


class SparseTensor:
    def __init__(self, feats, coords, stride=1):
        self.feats = feats
        self.coords = coords
        self.stride = make_ntuple(stride, ndim=3)
        self.cmaps = {}
        self.kmaps = {}

    @property
    def F(self):
        return self.feats

    @F.setter
    def F(self, feats):
        self.feats = feats

    @property
    def C(self):
        return self.coords

    @C.setter
    def C(self, coords):
        self.coords = coords

    @property
    def s(self):
        return self.stride

    @s.setter
    def s(self, stride):
        self.stride = stride

    def cpu(self):
        self.coords = self.coords.cpu()
        self.feats = self.feats.cpu()
        return self

    def cuda(self):
        return self  # the oracle is CPU-only

    def detach(self):
        self.coords = self.coords.detach()
        self.feats = self.feats.detach()
        return self

    def to(self, device, non_blocking=True):
        return self

    def __add__(self, other):
        output = SparseTensor(coords=self.coords, feats=self.feats + other.feats, stride=self.stride)
        output.cmaps = self.cmaps
        output.kmaps = self.kmaps
        return output


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = idx_query if idx_query is not None else {}
        self.weights = weights if weights is not None else {}
        self.additional_features = {}
        self.additional_features['idx_query'] = {}
        self.additional_features['counts'] = {}

    def cuda(self):
        return self

    def detach(self):
        self.F = self.F.detach()
        self.C = self.C.detach()
        return self

    def to(self, device, non_blocking=True):
        return self

    def __add__(self, other):
        tensor = PointTensor(self.F + other.F, self.C, self.idx_query, self.weights)
        tensor.additional_features = self.additional_features
        return tensor
